"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K tA tB reps"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.ops as ops  # noqa: E402

M, N, K, tA, tB, reps = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
A = torch.randn((K, M) if tA else (M, K), device=dev)
B = torch.randn((N, K) if tB else (K, N), device=dev)
C = torch.empty((M, N), device=dev)
for _ in range(reps):
    ops.gemm(A, B, out=C, transA=bool(tA), transB=bool(tB))
torch.cuda.synchronize()
print("done")
