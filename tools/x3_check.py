"""fp32 GEMM from six bf16 products (csrc/gemm_x3.hip) vs the fp32-MFMA kernel vs an fp64 product: error and rate.
usage: python tools/x3_check.py [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd.ops as ops  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def rel(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


def check(M, N, K, kind="randn"):
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    A = torch.randn((M, K), device=dev, generator=g)
    B = torch.randn((N, K), device=dev, generator=g)
    if kind == "wide":                       # 12 decades of dynamic range inside one reduction
        A = A * torch.exp(torch.randn((M, K), device=dev, generator=g) * 4)
        B = B * torch.exp(torch.randn((N, K), device=dev, generator=g) * 4)
    bias = torch.randn((N,), device=dev, generator=g)
    ref = A.double() @ B.double().t() + bias.double()
    c32 = ops.gemm_simple(A, B, transB=True, bias=bias)
    ia, _ = ops.x3_split(A)
    ib, _ = ops.x3_split(B)
    cx = ops.gemm_x3_grouped([dict(A=ia, B=ib, bias=bias)])[0]
    _, iat = ops.x3_split(A.t().contiguous(), plain=False, trans=True)       # the transposing pass must give the same image
    same = bool(torch.equal(iat.buf, ia.buf))
    e32, ex = rel(c32, ref), rel(cx, ref)
    # accumulate form
    c0 = torch.randn((M, N), device=dev, generator=g)
    cx2 = ops.gemm_x3_grouped([dict(A=ia, B=ib, out=c0.clone(), beta=1.0)])[0]
    eacc = rel(cx2, ref - bias.double() + c0.double())
    print("M=%6d N=%6d K=%6d %-5s  fp32-mfma err %.2e   x3 err %.2e   x3 accumulate err %.2e   transposed split identical: %s  %s"
          % (M, N, K, kind, e32, ex, eacc, same, "OK" if ex < max(4 * e32, 2e-7) and same and eacc < max(4 * e32, 2e-7) else "MISMATCH"),
          flush=True)


def timing(M, N, K, reps=10):
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K), device=dev)
    ia, _ = ops.x3_split(A)
    ib, _ = ops.x3_split(B)
    out = torch.empty((M, N), device=dev)
    res = {}
    def f32():
        ops.X3 = False
        ops.gemm(A, B, transB=True, out=out)
        ops.X3 = True

    ha, _ = ops.h2_split(A, scale=2.0 ** 10)
    hb, _ = ops.h2_split(B, dynamic=True)
    for name, fn in (("fp32-mfma", f32),
                     ("x3", lambda: ops.gemm_x3_grouped([dict(A=ia, B=ib, out=out)])),
                     ("h2", lambda: ops.gemm_h2_grouped([dict(A=ha, B=hb, out=out)])),
                     ("h2 split A dual", lambda: ops.h2_split(A, plain=True, trans=True, scale=2.0 ** 10)),
                     ("h2 split A dual dyn", lambda: ops.h2_split(A, plain=True, trans=True, dynamic=True)),
                     ("split A", lambda: ops.x3_split(A)),
                     ("split A dual", lambda: ops.x3_split(A, plain=True, trans=True))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * N * K
    print("M=%6d N=%6d K=%6d  fp32-mfma %.3f ms (%.0f TF)   x3 %.3f ms (%.0f TF fp32-equivalent, %.0f TF of bf16 MFMA)   split A %.3f ms  "
          "(dual %.3f ms, %.2f TB/s)   h2 %.3f ms (%.0f TF fp32-equivalent, %.0f TF of f16 MFMA; dual split %.3f ms, with device scale %.3f)"
          % (M, N, K, res["fp32-mfma"], fl / res["fp32-mfma"] / 1e9, res["x3"], fl / res["x3"] / 1e9,
             6 * fl / res["x3"] / 1e9, res["split A"], res["split A dual"], M * K * 16.0 / res["split A dual"] / 1e9,
             res["h2"], fl / res["h2"] / 1e9, 3 * fl / res["h2"] / 1e9, res["h2 split A dual"], res["h2 split A dual dyn"]), flush=True)


if __name__ == "__main__":
    check(300, 77, 50)
    check(256, 256, 16)
    check(1000, 515, 1153)
    check(2048, 4096, 1024)
    check(513, 4716, 2304, "wide")
    check(2176, 4096, 19200)
    if len(sys.argv) > 1:
        timing(19200, 4096, 1024)
        timing(19200, 1024, 4096)
        timing(19200, 4096, 1152)
        timing(2176, 4096, 19200)
        timing(1024, 4096, 19200)
        timing(8192, 8192, 8192)
        timing(1024, 23580, 1152)
