#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python -m pytest tests/test_gpu_round3.py -m gpu -q --timeout=300 -p no:cacheprovider -k "b1 or images" 2>&1 | tail -2
echo "== default (pd 3)"; timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | cut -c1-150
for v in b1_pd2 b1_pd4; do echo "== $v"; YT8M_LIB=tools/variants/lib_$v.so timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | cut -c1-150; done
