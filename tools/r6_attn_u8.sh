cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "attention_lstm_plugins" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_models.py tests/test_golden_models.py tests/test_gpu_round3.py -x -q -m gpu -k "attention or golden or u8" 2>&1 | tail -3
for m in lstm_attn lstm_posattn; do YT8M_NO_PROF=1 python tools/model_bench.py $m 2>&1 | grep "ms/step" | cut -c1-80; done
