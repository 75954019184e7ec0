cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_golden_models.py -x -q -m gpu -k "gru or lnlstm or layernorm or ln_lstm" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
for m in gru_pool ln_lstm; do timeout 300 python tools/model_bench.py $m 2>&1 | grep "ms/step"; done
YT8M_GRU_PERSIST_BWD=1 timeout 300 python tools/model_bench.py gru_pool 2>&1 | grep "ms/step"
YT8M_GRU_PERSIST=0 timeout 300 python tools/model_bench.py gru_pool 2>&1 | grep "ms/step"
