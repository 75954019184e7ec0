cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "bf16_logits or gru" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
timeout 900 python -m pytest tests/test_gpu_fullsize_golden.py tests/test_gpu_round2.py tests/test_gpu_models.py -x -q -m gpu -k "bf16 or c4 or config5 or chain" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
for z in 1 0; do echo "== Z16_LOGITS=$z"; YT8M_Z16_LOGITS=$z YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 2>&1 | grep "ms/step"; done
for z in 1 0; do echo "== Z16_LOGITS=$z"; YT8M_Z16_LOGITS=$z YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 2>&1 | grep "ms/step"; done
