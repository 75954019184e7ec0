cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "lnlstm or layernorm or ln_lstm or LayerNorm" 2>&1 | grep -E "passed|failed|assert " | head -5
export TMPDIR=/tmp; cd /tmp
YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ln2 -o ln -- python $GRAFT_REPO_ROOT/tools/model_bench.py ln_lstm > $GRAFT_REPO_ROOT/gpurun_out/ln2.txt 2>&1
grep "ms/step" $GRAFT_REPO_ROOT/gpurun_out/ln2.txt; head -6 $(find $GRAFT_REPO_ROOT/gpurun_out/ln2 -name "*kernel_stats.csv" | head -1) | cut -c1-60,150-230
