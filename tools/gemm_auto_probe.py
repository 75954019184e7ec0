"""Times ops.gemm (library dispatch) against the fp32 kernel on given (M, N, K, transA, transB) shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.ops as ops

SHAPES = [(1024, 1024, 73728, 0, 0), (1024, 1152, 23580, 0, 1), (1024, 2048, 14148, 0, 1), (512, 1024, 73728, 0, 0),
          (1024, 23580, 1152, 0, 0), (73728, 1024, 1024, 1, 0)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for M, N, K, ta, tb in SHAPES:
    A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g)
    res = {}
    for x3 in (True, False):
        ops.X3 = x3
        out = ops.gemm(A, B, transA=bool(ta), transB=bool(tb))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.gemm(A, B, out=out, transA=bool(ta), transB=bool(tb))
        torch.cuda.synchronize()
        res[x3] = ((time.perf_counter() - t0) / 10 * 1e3, out.clone())
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    e = [float((res[k][1].double() - ref).abs().max() / ref.abs().max()) for k in (True, False)]
    print("M=%6d N=%6d K=%6d ta=%d tb=%d  auto %.3f ms (%.0f TF, err %.1e)  fp32 %.3f ms (%.0f TF, err %.1e)" % (
        M, N, K, ta, tb, res[True][0], 2e-9 * M * N * K / res[True][0], e[0], res[False][0], 2e-9 * M * N * K / res[False][0], e[1]))
