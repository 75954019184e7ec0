cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cnn_prof -o cnn -- python $R/tools/model_bench.py ${1:-cnn_chain} > $R/gpurun_out/cnn_prof.txt 2>&1
f=$(find $R/gpurun_out/cnn_prof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms / 7 steps: %.2f" % (tot/1e6))
for r in rows[:32]:
    print("%-86s %5s %8.1f us %6.2f%%" % (r['Name'].replace('(anonymous namespace)::','')[:86], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
find $R/gpurun_out/cnn_prof -name "*.csv" -size +3M -delete
