#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
export YT8M_DP_RESERVED_CUS=0
for mode in plain red; do
  rm -rf /tmp/tr_$mode
  X=""; if [ $mode = red ]; then X="--force-reducer"; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$mode -o t -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-gap --no-extra --no-roofline $X > $O/r3_c4_$mode.json 2> $O/r3_c4_$mode.err
  f=$(find /tmp/tr_$mode -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_step.py $f 30 2 u8_frames_tm_kernel > $O/r3_c4_trace_$mode.txt 2>&1
  python $R/tools/bench_brief.py $mode < $O/r3_c4_$mode.json | head -1
done
