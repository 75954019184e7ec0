# A/B of the interleaved three-plane GEMM (YT8M_X3_PIPE) -- unit shapes, counters, headline step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q > gpurun_out/ab/pytest.txt 2>&1; grep -E "passed|failed" gpurun_out/ab/pytest.txt | tail -2
for p in 0 1; do YT8M_X3_PIPE=$p python tools/x3_time.py 2>/dev/null | sed "s/^/x3pipe=$p /"; done
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for p in 0 1; do YT8M_X3_PIPE=$p rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/ab/pmc$p -o q --output-format csv -- python $R/tools/x3_time.py > /dev/null 2>&1; python $R/tools/pmc_mfma_busy.py $R/gpurun_out/ab/pmc$p/q_counter_collection.csv | grep gemm_x3; done
