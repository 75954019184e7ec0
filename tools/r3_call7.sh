#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/r3_c7_$name.json 2> $O/r3_c7_$name.err; python tools/bench_brief.py $name < $O/r3_c7_$name.json 2>&1 | head -1; grep -i "error\|Traceback" $O/r3_c7_$name.err | head -3; }
for p in 6,4,2 7,4,2 6,3,2 8,5,3 6,4,2,1 7,5,3,1 5,3,1 3,2,1 8,4,2 10,6,3 6,4,3 4,3,2,1; do run p_$p YT8M_STACK_BWD_PARTS=$p; done
run p642_sw2 YT8M_STACK_SW2=1 YT8M_STACK_BWD_PARTS=6,4,2
run p642_again YT8M_STACK_BWD_PARTS=6,4,2
run dflt YT8M_X=1
