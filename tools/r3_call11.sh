#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
echo "== stagger"; timeout 300 python tools/gemm_bf16_big.py 2>&1 | grep -v amdgpu.ids
echo "== lockstep"; YT8M_LIB=$R/tools/variants/lib_nostagger.so timeout 300 python tools/gemm_bf16_big.py 2>&1 | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -p no:cacheprovider -k "bf16" 2>&1 | tail -3
