# round 6: state check -- full GPU suite, then the headline step under kernel trace (timeline + stats of the timed region)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_state
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/pytest.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gap --no-extra < /dev/null > $O/bench_line_profiled.json 2> $O/bench_prof.err
f=$(find $O/bench -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $f 40 10 u8_frames_tm_kernel > $O/step_timeline.txt 2>&1
cp $(find $O/bench -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*.csv" -size +6M -delete
cat $O/pytest.txt
