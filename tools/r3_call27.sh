#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3_gpu_suite.log 2>&1; grep -E "passed|failed" gpurun_out/r3_gpu_suite.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep "smoke ok" | tail -2
