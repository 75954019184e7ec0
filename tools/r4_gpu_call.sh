#!/bin/bash
# round 4 working call: early head optimizer pass -- parity, A/B, partition around it
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c7
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -8 | cut -c1-300
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
{
run A=warm
run A=default
run YT8M_EARLY_ADAM=0
run A=default
run YT8M_EARLY_ADAM=0
run YT8M_STACK_BWD_PARTS=4,4,2,3
run YT8M_STACK_BWD_PARTS=2,2,1,2
run YT8M_STACK_BWD_PARTS=3,3,2,2
run YT8M_STACK_BWD_PARTS=3,3,1,1
run YT8M_STACK_SW2=1
run YT8M_STACK_DX_STREAM=1
run YT8M_STACK_SUB0_LAST=2
} > $O/ab.txt 2>&1
cat $O/ab.txt
