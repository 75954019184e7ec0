#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout=600 -p no:cacheprovider -k "image_t or persistent_kernels_vs_oracle" 2>&1 | tail -5
cd /tmp
for mode in native python; do
  rm -rf /tmp/tr_$mode
  if [ $mode = python ]; then export YT8M_LSTM_STACK_NATIVE=0; else unset YT8M_LSTM_STACK_NATIVE; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$mode -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline > $O/r3_tr_$mode.json 2> $O/r3_tr_$mode.err
  f=$(find /tmp/tr_$mode -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_step.py $f 40 2 > $O/r3_trace_$mode.txt 2>&1
  s=$(find /tmp/tr_$mode -name "*kernel_stats.csv" | head -1)
  cut -c1-200 $s | head -30 > $O/r3_stats_$mode.csv
  python $R/tools/bench_brief.py $mode < $O/r3_tr_$mode.json
done
