"""Timing aid (-DYT8M_GEMM_TIMING variant): per-phase cycles of one wave of one tile of the grouped GEMM's K loop."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops, yt8m_amd._lib as L
dev = torch.device("cuda:0")
raw = ctypes.CDLL(L.LIB_PATH)
for name, M, N, K, tA in [("fwd gates", 1024, 14148, 1152, 0), ("dW gates", 1152, 14148, 1024, 1), ("square", 4096, 4096, 4096, 0)]:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    for _ in range(3):
        ops.gemm(A, B, out=C, transA=bool(tA))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    raw.yt8m_debug_gemm_phase(out)
    n = max(out[5], 1)
    tot = sum(out[i] for i in range(5))
    print("%-10s K-steps %4d  per K-step: dma-issue %5.0f  lds-reads %5.0f  mfma-block %5.0f  vmcnt-wait %5.0f  barrier %5.0f  | total %5.0f"
          % ((name, n) + tuple(out[i] / n for i in range(5)) + (tot / n,)))
