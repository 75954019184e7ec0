cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_kernels.py -x -q -m gpu -k "gru" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
echo "== U=16 whole chip"; python tools/gru_persist_check.py 300 2>&1 | tail -2
echo "== U=16 128 CUs";   YT8M_GRU_BWD_CUS=128 python tools/gru_persist_check.py 300 2>&1 | tail -1
echo "== U=8";            YT8M_GRU_BWD_U=8 python tools/gru_persist_check.py 300 2>&1 | tail -1
