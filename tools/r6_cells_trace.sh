# round 6: gru_pool / ln_lstm plugin steps under kernel trace: per-kernel stats, launch durations and gaps inside the time loops
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_cells
rm -rf $O; mkdir -p $O
cd /tmp
for m in gru_pool ln_lstm; do
  YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -o $m -- python $R/tools/model_bench.py $m < /dev/null > $O/$m.txt 2>&1
  grep "ms/step" $O/$m.txt
  cp $(find $O/$m -name "*kernel_stats.csv" | head -1) $O/${m}_kernel_stats.csv
  python - $(find $O/$m -name "*kernel_trace.csv" | head -1) <<'PY' > $O/${m}_gaps.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[len(ev) // 2:]                      # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i in range(1, len(ev)):
    n = ev[i][2].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    dur[n].append((ev[i][1] - ev[i][0]) / 1e3)
    gap[n].append((ev[i][0] - ev[i - 1][1]) / 1e3)
print("%-52s %7s %9s %9s" % ("kernel", "calls", "avg us", "gap before us"))
for n in sorted(dur, key=lambda n: -sum(dur[n]))[:14]:
    print("%-52s %7d %9.2f %9.2f" % (n, len(dur[n]), sum(dur[n]) / len(dur[n]), sum(gap[n]) / len(gap[n])))
PY
  cat $O/${m}_gaps.txt
done
find $O -name "*.csv" -size +6M -delete
