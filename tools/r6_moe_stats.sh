export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_moe
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/moe -o moe -- python $R/bench.py --workload moe --steps 200 --warmup 10 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/moe_line.json 2>/dev/null
cp $(find $O/moe -name "*kernel_stats.csv" | head -1) $O/moe_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/moe_kernel_stats.csv')))
steps=230.0
for r in rows[:18]:
    print("%-64s %5.1f/step %7.1f us %.3f ms/step" % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:64], int(r['Calls'])/steps, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6/steps))
PY
find $O -name "*.csv" -size +4M -delete
