#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_x3.py tests/test_gpu_models.py -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -15
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline"
run() { name=$1; shift; env "$@" timeout 300 $B $EXTRA > $O/r3_c3_$name.json 2> $O/r3_c3_$name.err; python tools/bench_brief.py $name < $O/r3_c3_$name.json 2>&1 | head -1; tail -2 $O/r3_c3_$name.err | cut -c1-200; }
EXTRA=""; run plain YT8M_X=1
EXTRA="--force-reducer"; run reducer_allreduce YT8M_X=1
EXTRA="--force-reducer"; run reducer_rsag YT8M_DP_ALGO=rs_ag
EXTRA="--force-reducer"; run reducer_noreserve YT8M_DP_RESERVED_CUS=0
EXTRA=""; run reserve32_plain YT8M_PERSIST_RESERVED_CUS=32
