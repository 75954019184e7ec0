#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c3
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -12 | cut -c1-300
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
{
run A=warm
run A=default
run YT8M_X3_FIXUP_KERNEL=1
run YT8M_STACK_COLPARTS=0
run YT8M_X3_FIXUP_KERNEL=1 YT8M_STACK_COLPARTS=0
run A=default
run YT8M_X3_FIXUP_KERNEL=1
EXTRA="--workload moe --steps 200"
run A=default
run YT8M_X3_FIXUP_KERNEL=1
EXTRA="--workload netvlad --steps 30"
run A=default
run YT8M_X3_FIXUP_KERNEL=1
EXTRA=
} > $O/ab.txt 2>&1
cat $O/ab.txt
for kv in YT8M_X3_FIXUP_KERNEL=1 YT8M_STACK_COLPARTS=0; do
  env $kv timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_x3.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | sed "s/^/$kv: /" >> $O/knob_parity.txt
done
cat $O/knob_parity.txt
timeout 500 python tools/reader_bench.py > $O/reader_bench.txt 2> $O/reader_bench.err
cat $O/reader_bench.txt; tail -3 $O/reader_bench.err
