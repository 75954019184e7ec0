"""bf16 GEMM on one-plane operand images (gemm_b1_kernel) against the row-major large-tile kernel on the cfg[4] head shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
import yt8m_amd.ops as ops
from yt8m_amd.ops import _p, _stream

dev = torch.device("cuda:0")
lib = L.lib()


def img(x, trans=False):
    R, C = x.shape
    n = lib.yt8m_x3_image_bytes(C if trans else R, R if trans else C) // 3
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_bf16_image(_p(x), R, C, C, 1.0, None if trans else _p(out), _p(out) if trans else None, _stream()))
    return out


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


w = torch.randn(4096, 4096, device=dev)
for _ in range(0 if os.environ.get("B1_ONLY") else 40):
    ops.gemm(w, w)
ws = ops._workspace(dev)
SHAPES = [("head fwd", [(8192, 14148, 2304), (8192, 9432, 2304)]), ("head dW", [(2304, 14148, 8192), (2304, 9432, 8192)]),
          ("head dx g", [(8192, 2304, 14148)]), ("head dx e", [(8192, 2304, 9432)]), ("cfg1 fwd", [(1024, 14148, 1152), (1024, 9432, 1152)]),
          ("ragged", [(1000, 777, 1000), (300, 5000, 72)]), ("sq4k", [(4096, 4096, 4096)]), ("sq8k", [(8192, 8192, 8192)])]
if os.environ.get("B1_ONLY"):                              # restrict to one shape set (counter passes)
    SHAPES = [x for x in SHAPES if x[0] == os.environ["B1_ONLY"]]
for label, probs in SHAPES:
    items, pr, keep, fl = [], [], [], 0.0
    for (M, N, K) in probs:
        A32, B32 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        if os.environ.get("B1_ZEROS"):                      # operand entropy probe: zero-filled operands draw less power (higher clock)
            A32.zero_(); B32.zero_()
        bias = torch.randn(N, device=dev)
        A = ops._bf16_empty(M, K, dev); A.copy_(A32)
        B = ops._bf16_empty(N, K, dev); B.copy_(B32)
        items.append(dict(A=A, B=B, bias=bias))
        ia, ib = img(A32), img(B32)
        out = torch.empty((M, N), device=dev)
        pr.append(L.GemmProblem(M, N, K, ia.data_ptr(), 0, ib.data_ptr(), 0, out.data_ptr(), N, bias.data_ptr(), 0.0))
        keep.append((ia, ib, out, A32, B32))
        fl += 2.0 * M * N * K
    arr = (L.GemmProblem * len(pr))(*pr)
    t0 = timeit(lambda: ops.gemm_bf16_nt_grouped(items))
    t1 = timeit(lambda: L.check(lib.yt8m_gemm_b1_nt_grouped(len(pr), arr, _p(ws), ws.numel() * 4, _stream())))
    ref = ops.gemm_bf16_nt_grouped(items)
    err = max(float((k[2] - r).abs().max()) / max(1.0, float(r.abs().max())) for k, r in zip(keep, ref))
    x = keep[0][3]
    t2 = timeit(lambda: img(x))
    print("%-10s row-major %7.3f ms %7.1f TF/s | image b1 %7.3f ms %7.1f TF/s | max rel diff %.2e | image pass of A[0] %.3f ms"
          % (label, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, err, t2), flush=True)
