#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); export TMPDIR=/tmp
echo "== k64"; timeout 300 python tools/gemm_bf16_big.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== k32 ring"; YT8M_BF16_K64=0 timeout 300 python tools/gemm_bf16_big.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --timeout=600 -p no:cacheprovider -k "bf16" 2>&1 | tail -3
