#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
YT8M_LSTM_STACK_NATIVE=0 timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_x3.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "lstm or Lstm or stack or persist" 2>&1 | grep -E "^FAILED|^E  " | cut -c1-220 | head -40
