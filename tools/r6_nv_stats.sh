export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_nv
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nv -o nv -- python $R/bench.py --workload netvlad --steps 30 --warmup 6 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/nv_line.json 2>/dev/null
cp $(find $O/nv -name "*kernel_stats.csv" | head -1) $O/nv_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/nv_kernel_stats.csv')))
steps=41.0
for r in rows[:30]:
    print("%-64s %5.1f/step %7.1f us %.3f ms/step" % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:64], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/steps))
PY
find $O -name "*.csv" -size +4M -delete
