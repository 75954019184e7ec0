"""Timing aid (variant builds with -DYT8M_GEMM_TIMING): per-tile cycles of the GEMM main loop / epilogue."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops
dev = torch.device("cuda:0")
for name, M, N, K, tA in [("fwd gates", 1024, 14148, 1152, 0), ("dW gates", 1152, 14148, 1024, 1), ("square", 4096, 4096, 4096, 0)]:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    for _ in range(3):
        ops.gemm(A, B, out=C, transA=bool(tA))
    torch.cuda.synchronize()
    t = C[::128, ::128]
    loop, epi = t, C[::128, 1::128]
    full = (loop > 1000)                      # tiles handled whole (split-K parts write nothing here)
    print("%-10s tiles %4d  loop cycles mean %.0f  epilogue mean %.0f max %.0f  (epilogue share %.1f%%)"
          % (name, int(full.sum()), loop[full].mean(), epi[full].mean(), epi[full].max(), 100 * epi[full].mean() / (loop[full].mean() + epi[full].mean())))
