#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "== 8 waves"; timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | cut -c1-150
echo "== 4 waves"; YT8M_B1_W4=1 timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | cut -c1-150
YT8M_B1_W4=1 timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q --timeout=300 -p no:cacheprovider -k "b1 or images or bf16" 2>&1 | tail -3
