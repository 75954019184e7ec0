#!/bin/bash
# round-3 GPU call 1: full GPU test suite, then the headline step with the native stack on / off and a few partition knobs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/r3_tests1.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3_tests1.log
tail -25 gpurun_out/r3_tests1.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r3_b_$name.json 2> gpurun_out/r3_b_$name.err; python tools/bench_brief.py $name < gpurun_out/r3_b_$name.json 2>&1 | head -3; tail -3 gpurun_out/r3_b_$name.err; }
run native YT8M_X=1
run python YT8M_LSTM_STACK_NATIVE=0
run native_b4 YT8M_LSTM_PERSIST_BWD_CHUNKS=4
run native_b2 YT8M_LSTM_PERSIST_BWD_CHUNKS=2
run native_again YT8M_X=1
