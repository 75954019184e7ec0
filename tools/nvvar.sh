export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "netvlad or config5" 2>&1 | tail -6
cd /tmp; timeout 100 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nvb5 -- python $GRAFT_REPO_ROOT/tools/netvlad_bench.py 128 1024 < /dev/null 2>&1 | grep "nsplit"; f=$(find $GRAFT_REPO_ROOT/gpurun_out/nvb5 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/prof_summary.py "$f" | grep "vlad_rows\|vlad_cols"
