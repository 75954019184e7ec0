# round 6: configs[4] (bf16, B = 1024) under kernel trace: per-kernel stats + one step's timeline
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_c5
rm -rf $O; mkdir -p $O
cd /tmp
YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o c5 -- python $R/tools/model_bench.py config5_bf16_b1024 < /dev/null > $O/config5_bf16.txt 2>&1
f=$(find $O/c5 -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_step.py $f 15 2 adam_tile_kernel\<true > $O/step_timeline.txt 2>&1
cp $(find $O/c5 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*.csv" -size +6M -delete
tail -3 $O/config5_bf16.txt
