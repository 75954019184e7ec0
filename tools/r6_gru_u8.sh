cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "gru_and_layernorm" 2>&1 | tail -15
timeout 900 python -m pytest tests -x -q -m gpu -k "gru or layernorm or ln_lstm or LayerNorm or lnlstm" 2>&1 | grep -E "passed|failed" | tail -3
for m in gru_pool ln_lstm; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-80; done
