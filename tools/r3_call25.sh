#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -q --timeout=300 -p no:cacheprovider -k "mix or bf16 or images" 2>&1 | tail -3
timeout 100 python tools/mixb_bench.py 2>&1 | tail -1
YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 2>&1 | grep "B=" | cut -c1-100
