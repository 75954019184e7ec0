"""Sustained-load check: TFLOP/s of one fp32 GEMM shape over successive windows (power / clock behaviour of the box)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

dev = torch.device("cuda:0")
M, N, K = 8192, 23580, 1152
fresh = len(sys.argv) > 1 and sys.argv[1] == "fresh"      # new output buffer every call (as the model does)
A = torch.randn(M, K, device=dev)
B = torch.randn(K, N, device=dev)
out = torch.empty(M, N, device=dev)
for w in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 40
    for _ in range(n):
        if fresh:
            ops.gemm(A, B)
        else:
            ops.gemm(A, B, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("window %2d  %7.3f ms  %6.1f TFLOP/s" % (w, ms, 2.0 * M * N * K / ms / 1e9))
