cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "bf16_logits" 2>&1 | grep -E "passed|failed|Error|assert |timed out|error" | head -10
for z in 1 0; do echo "== Z16=$z"; YT8M_Z16_LOGITS=$z timeout 900 python -m pytest tests/test_gpu_fullsize_golden.py -x -q -m gpu -k "c4" 2>&1 | grep -E "passed|failed|AssertionError: \(" | head -5; done
