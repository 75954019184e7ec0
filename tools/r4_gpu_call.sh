#!/bin/bash
# round 4 working call: full -m gpu suite, then A/B of the rotated backward epilogue (stand-alone kernel and headline step)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c4
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -12 | cut -c1-300
{
for rot in 1 0; do for b in 128 512; do
  echo "== YT8M_BWD_ROT=$rot B=$b"; YT8M_BWD_ROT=$rot PCHECK_B=$b timeout 120 python tools/persist_check.py bwd 2>&1 | grep "us/step" | tail -3
done; done
} > $O/persist_bwd.txt 2>&1
cat $O/persist_bwd.txt
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
{
run A=warm
run A=default
run YT8M_BWD_ROT=0
run A=default
run YT8M_BWD_ROT=0
run YT8M_STACK_BWD_PARTS=2,2,1,1
run YT8M_STACK_BWD_PARTS=1,1,1
EXTRA="--batch 512 --steps 8 --warmup 2"
run A=default
run YT8M_BWD_ROT=0
EXTRA=
} > $O/ab.txt 2>&1
cat $O/ab.txt
YT8M_BWD_ROT=0 timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | sed "s/^/YT8M_BWD_ROT=0: /"
timeout 500 python tools/reader_bench.py > $O/reader_bench.txt 2> $O/reader_bench.err
cat $O/reader_bench.txt; tail -3 $O/reader_bench.err
