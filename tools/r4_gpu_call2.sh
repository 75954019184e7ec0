#!/bin/bash
# round 4, second GPU call: the whole -m gpu suite (new reference pins, golden replays, full-size checksums), combine A/B, bf16
# recurrence knob, reader bench, one driver-style bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c2
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^$" $O/pytest.log | tail -25 | cut -c1-300
run() { env "$@" timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gap --no-extra --no-roofline $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s %.2f ms' % (' '.join(sys.argv[1:]), d['ms_per_step']))" "$@" $EXTRA; }
{
run A=warm
run A=default
run YT8M_X3_FIXUP_KERNEL=1
run A=default
run YT8M_X3_FIXUP_KERNEL=1
EXTRA="--dtype bf16"
run YT8M_REC_BF16=1
run YT8M_REC_BF16=0
EXTRA="--workload moe --steps 200"
run A=default
run YT8M_X3_FIXUP_KERNEL=1
EXTRA=
} > $O/ab.txt 2>&1
cat $O/ab.txt
timeout 400 python tools/reader_bench.py --videos 4096 --threads 4,8,16,32 > $O/reader_bench.txt 2> $O/reader_bench.err
cat $O/reader_bench.txt; tail -3 $O/reader_bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
tail -5 $O/bench.err
python tools/bench_brief.py full < $O/bench_line.json
