"""bf16 NT GEMM throughput on the MoE-head shapes (fwd: x . Wt^T, bwd: xT . dZt^T)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")
for name, M, N, K in [("fwd gates+experts (grouped)", 1024, 23580, 1152), ("dW (grouped)", 1152, 23580, 1024),
                      ("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192)]:
    A = (torch.rand((M, K), device=dev) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), device=dev) * 2 - 1).to(torch.bfloat16)
    C = torch.empty((M, N), device=dev)
    for _ in range(2):
        ops.gemm_bf16_nt_grouped([dict(A=A, B=B, out=C)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_bf16_nt_grouped([dict(A=A, B=B, out=C)])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-28s M=%5d N=%5d K=%5d  %8.3f ms  %8.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
