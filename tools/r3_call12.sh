#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); export TMPDIR=/tmp
for v in bf16_NO_DMA bf16_NO_LDS bf16_NO_MFMA bf16_nodma_nolds; do echo "== $v"; YT8M_LIB=$R/tools/variants/lib_$v.so timeout 300 python tools/gemm_bf16_big.py 2>&1 | grep "head" | cut -c1-110; done
