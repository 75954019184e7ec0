#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3_gpu_suite.log 2>&1; tail -3 gpurun_out/r3_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 500 python tools/model_bench.py > gpurun_out/r3_model_bench.txt 2>&1; grep -c "B=" gpurun_out/r3_model_bench.txt
