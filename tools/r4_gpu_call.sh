#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c11
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -x > $O/pytest.log 2>&1
grep -v "^$" $O/pytest.log | tail -40 | cut -c1-250
