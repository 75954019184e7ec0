#!/bin/bash
# round 4 working call: recurrence-written operand images -- parity, then A/B on the headline step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c6
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -x > $O/pytest_r4.log 2>&1; echo "rc=$?" >> $O/pytest_r4.log
grep -v "^$" $O/pytest_r4.log | tail -15 | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -8 | cut -c1-300
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
{
run A=warm
run A=default
run YT8M_STACK_FUSED_IMAGES=0
run A=default
run YT8M_STACK_FUSED_IMAGES=0
run YT8M_STACK_BWD_PARTS=3,2,1
run YT8M_STACK_BWD_PARTS=2,2,2,1,1
run YT8M_STACK_BWD_PARTS=1,1,1,1,1,1
run YT8M_STACK_BWD_PARTS=2,1,1,1,1
} > $O/ab.txt 2>&1
cat $O/ab.txt
for rot in 1; do echo "== stand-alone bwd kernel"; timeout 120 python tools/persist_check.py bwd 2>&1 | grep "us/step" | tail -2; done
