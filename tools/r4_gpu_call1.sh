#!/bin/bash
# round 4, first GPU call: parity of the in-kernel split-K combine / sub-parts / fused dz split, then their A/B on the headline step
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c1
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-90s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@"; }
{
run A=warm
run A=default
run YT8M_X3_FIXUP_KERNEL=1
run YT8M_STACK_SUB0_LAST=1
run YT8M_X3_FIXUP_KERNEL=1 YT8M_STACK_SUB0_LAST=1
run YT8M_STACK_SUB0_LAST=2
run YT8M_STACK_SUB0_LAST=5
run YT8M_STACK_SUB0=3,2,1
run YT8M_STACK_SUB0=3,2,2
run YT8M_STACK_FUSE_DZ_SPLIT=1
run YT8M_STACK_BWD_PARTS=2,2,1,1
run YT8M_STACK_BWD_PARTS=1,1,1 YT8M_STACK_SUB0=2,1,1
run A=default
} > $O/ab.txt 2>&1
cat $O/ab.txt
# parity of knob variants that are not the default (bitwise-deterministic paths: run the LSTM tests under each)
for kv in YT8M_X3_FIXUP_KERNEL=1 YT8M_STACK_FUSE_DZ_SPLIT=1 YT8M_STACK_SUB0=3,2,2; do
  env $kv timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_x3.py -m gpu -x -q 2>&1 | tail -2 | sed "s/^/$kv: /" >> $O/knob_parity.txt
done
cat $O/knob_parity.txt
