"""Throughput of yt8m_gemm_f32 vs number of workgroup tiles (occupancy / tail experiment)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")
K = 2048
for tiles_m in (8, 16, 24, 32, 40, 48, 56, 64, 96, 128):
    M, N = tiles_m * 128, 16 * 128
    A = torch.rand((M, K), device=dev) * 2 - 1
    B = torch.rand((K, N), device=dev) * 2 - 1
    C = torch.empty((M, N), device=dev)
    for _ in range(2):
        ops.gemm(A, B, out=C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(A, B, out=C)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("tiles=%5d (%.2f per CU)  %8.3f ms  %7.1f TFLOP/s" % (tiles_m * 16, tiles_m * 16 / 256.0, ms, 2.0 * M * N * K / ms / 1e9))
