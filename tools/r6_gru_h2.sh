cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "gru or layernorm or ln_lstm or LayerNorm or lnlstm" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for m in gru_pool ln_lstm; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-80; done
