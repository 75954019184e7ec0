#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "== random"; timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | grep head | cut -c1-110
echo "== zeros"; B1_ZEROS=1 timeout 200 python tools/b1_bench.py 2>&1 | grep "image b1" | grep head | cut -c1-110
