#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python -m pytest tests/test_gpu_round2.py tests/test_gpu_models.py -m gpu -q --timeout=300 -p no:cacheprovider -k "frame_pool or attention or config5 or dbof or Dbof" 2>&1 | tail -3
for i in 1 2; do YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 2>&1 | grep "B=" | cut -c1-100; done
timeout 300 python tools/model_bench.py lstm_attn dbof 2>&1 | grep "B=" | cut -c1-100
