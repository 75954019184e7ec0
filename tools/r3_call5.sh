#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/r3_c5_$name.json 2> $O/r3_c5_$name.err; python tools/bench_brief.py $name < $O/r3_c5_$name.json 2>&1 | head -1; grep -i "error\|Traceback" $O/r3_c5_$name.err | head -3; }
run base YT8M_X=1
run dx YT8M_STACK_DX_STREAM=1
run sw2 YT8M_STACK_SW2=1
run dx_sw2 YT8M_STACK_DX_STREAM=1 YT8M_STACK_SW2=1
run parts543 YT8M_STACK_BWD_PARTS=5,4,3
run parts345 YT8M_STACK_BWD_PARTS=3,4,5
run parts_dx_433 YT8M_STACK_DX_STREAM=1 YT8M_STACK_BWD_PARTS=4,3,3
run dx_b4 YT8M_STACK_DX_STREAM=1 YT8M_LSTM_PERSIST_BWD_CHUNKS=4
run dx_b6 YT8M_STACK_DX_STREAM=1 YT8M_LSTM_PERSIST_BWD_CHUNKS=6
run base2 YT8M_X=1
