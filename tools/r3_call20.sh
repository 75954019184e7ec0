#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round3.py tests/test_gpu_models.py -m gpu -q --timeout=300 -p no:cacheprovider -k "u8 or composite or config5" 2>&1 | grep -E "passed|failed" | tail -1
rm -rf $O/prof_c5; mkdir -p $O/prof_c5
cd /tmp && YT8M_NO_PROF=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 --output-format csv -- python $R/tools/model_bench.py config5_bf16_b1024 2>&1 | grep "B=" | cut -c1-100
find $O/prof_c5 -name "*kernel_trace.csv" -delete
