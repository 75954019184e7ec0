"""L2 channel camping probe: the b1 product with the operand images' K-block stride padded by 0..4 blocks (1 KiB each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
import yt8m_amd.ops as ops
from yt8m_amd.ops import _p, _stream
dev = torch.device("cuda:0")
lib = L.lib()
ws = ops._workspace(dev)
w = torch.randn(4096, 4096, device=dev)
for _ in range(30):
    ops.gemm(w, w)


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


for (M, N, K) in [(8192, 14148, 2304), (2304, 14148, 8192), (8192, 2304, 14148), (8192, 9432, 2304)]:
    A32, B32 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    ref = None
    line = "M=%5d N=%5d K=%5d KB=%4d:" % (M, N, K, (K + 15) // 16)
    for pad in (0, 1, 2, 3, 4, 5, 8, 9):
        Ap = torch.zeros(M, K + 16 * pad, device=dev); Ap[:, :K] = A32
        Bp = torch.zeros(N, K + 16 * pad, device=dev); Bp[:, :K] = B32
        ia, ib = ops.bf16_image(Ap), ops.bf16_image(Bp)
        kb = (K + 15) // 16 + pad
        out = torch.empty((M, N), device=dev)
        Kp = K if K % 16 == 0 else K + 16 * pad          # a K range of a larger image must be a multiple of 16
        pr = (L.GemmProblem * 1)(L.GemmProblem(M, N, Kp, ia.buf.data_ptr(), kb if pad else 0, ib.buf.data_ptr(), kb if pad else 0,
                                               out.data_ptr(), N, None, 0.0))
        t = timeit(lambda: L.check(lib.yt8m_gemm_b1_nt_grouped(1, pr, _p(ws), ws.numel() * 4, _stream())))
        if ref is None:
            ref = out.clone()
        assert torch.equal(ref, out), "padded image changed the result"
        line += "  +%d %.0f" % (pad, 2e-9 * M * N * K / t)
    print(line + "  TF/s", flush=True)
