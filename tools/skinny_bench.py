"""GB/s of the skinny fully-connected kernels on the attention-logit shapes: python tools/skinny_bench.py (GPU box)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (M, K, N) in [(307200, 1152, 8), (38400, 1152, 8), (38400, 1024, 8), (307200, 2304, 8), (307200, 1152, 16)]:
    x = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev)
    dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    dW = torch.empty(K, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    xb = 4.0 * M * K
    f = timeit(lambda: ops.skinny_fwd(x, W, None, y))
    w = timeit(lambda: ops.skinny_dw(x, dy, dW))
    d = timeit(lambda: ops.skinny_dx(dy, W, dx=dx))
    g = timeit(lambda: ops.gemm(x, W, out=y))
    print("M=%6d K=%4d N=%2d | fwd %.3f ms %.2f TB/s | dW %.3f ms %.2f TB/s | dx %.3f ms %.2f TB/s | MFMA-tile fwd %.3f ms"
          % (M, K, N, f, xb / f / 1e9, w, xb / w / 1e9, d, xb / d / 1e9, g))
    del x, dx
