#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout=600 -p no:cacheprovider -k "native or c_abi or persistent_kernels or full_length or layer0" 2>&1 | tail -4
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/r3_c6_$name.json 2> $O/r3_c6_$name.err; python tools/bench_brief.py $name < $O/r3_c6_$name.json 2>&1 | head -1; grep -i "error\|Traceback" $O/r3_c6_$name.err | head -3; }
run dflt543 YT8M_X=1
run sw2 YT8M_STACK_SW2=1
run uniform3 YT8M_STACK_BWD_PARTS=1,1,1
run p642 YT8M_STACK_BWD_PARTS=6,4,2
run p532 YT8M_STACK_BWD_PARTS=5,3,2
run p5432 YT8M_STACK_BWD_PARTS=5,4,3,2
run p654 YT8M_STACK_BWD_PARTS=6,5,4
run p543_sw2 YT8M_STACK_SW2=1 YT8M_STACK_BWD_PARTS=5,4,3
run dflt543b YT8M_X=1
