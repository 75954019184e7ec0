"""Micro-benchmark of yt8m_gemm_f32 on the GEMM shapes of the BASELINE configs (per-shape TFLOP/s via events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [  # name, M, N, K, transA, transB
    ("cfg2 fwd gates   x.Wg", 1024, 14148, 1152, 0, 0),
    ("cfg2 fwd experts x.We", 1024, 9432, 1152, 0, 0),
    ("cfg2 dW gates  xT.dZg", 1152, 14148, 1024, 1, 0),
    ("cfg2 dW experts xT.dZe", 1152, 9432, 1024, 1, 0),
    ("cfg2 dx       dZ.WgT", 1024, 1152, 14148, 0, 1),
    ("lstm head fwd B=128", 128, 14148, 4096, 0, 0),
    ("lstm head dW  B=128", 4096, 14148, 128, 1, 0),
    ("lstm inproj F*B=38400", 38400, 4096, 1152, 0, 0),
    ("lstm recur B=128", 128, 4096, 1024, 0, 0),
    ("square 4096", 4096, 4096, 4096, 0, 0),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, M, N, K, tA, tB in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device=dev)
    B = torch.randn((N, K) if tB else (K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    for _ in range(2):
        ops.gemm(A, B, out=C, transA=bool(tA), transB=bool(tB))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(A, B, out=C, transA=bool(tA), transB=bool(tB))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-26s M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
