#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q --timeout=600 -p no:cacheprovider -k "b1 or bf16 or composite or config5 or images" 2>&1 | tail -5
for v in 1 0; do YT8M_BF16_IMAGES=$v YT8M_NO_PROF=1 timeout 300 python tools/model_bench.py config5_bf16_b1024 lstm_bf16 2>&1 | grep "B=" | cut -c1-100; done
