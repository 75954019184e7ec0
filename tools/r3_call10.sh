#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for cfg in config5_bf16_b1024 moe; do
  rm -rf /tmp/st_$cfg
  YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$cfg -o t -- python $R/tools/model_bench.py $cfg > $O/r3_c10_$cfg.log 2>&1
  s=$(find /tmp/st_$cfg -name "*kernel_stats.csv" | head -1)
  python - "$s" > $O/r3_c10_${cfg}_stats.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
    print("%-70s calls %6s  avg_us %9.1f  total_ms %9.2f  %5s%%" % (n[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
  grep "B=" $O/r3_c10_$cfg.log | cut -c1-200
done
