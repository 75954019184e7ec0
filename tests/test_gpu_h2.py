"""-m gpu, round 5: fp32 products as THREE f16 MFMA products of two-plane half images (csrc/gemm_x3.hip gemm_h2q_kernel,
csrc/x3_image.h split_h2): images bit for bit against oracle/x3_ref.image_h2, the device's three-product sum against the oracle's, and
the error against an fp64 product next to the exact-fp32 MFMA kernel's and the six-product bf16 split's."""
import numpy as np
import pytest
import torch

import yt8m_amd.ops as ops

pytestmark = pytest.mark.gpu


def _pow2_scale(x, target=13):
    m = float(x.abs().max())
    return 2.0 ** (target - int(np.ceil(np.log2(m)))) if m > 0 else 1.0


@pytest.mark.parametrize("R,C,scale", [(70, 50, 1.0), (64, 64, 4096.0), (300, 130, 2.0 ** -3), (33, 1153, 512.0)])
def test_h2_images_equal_the_oracle_bit_for_bit(dev, R, C, scale):
    from oracle import x3_ref
    rs = np.random.RandomState(R * 5 + C)
    x = (rs.randn(R, C) * np.exp(rs.randn(R, C) * 2)).astype(np.float32)
    x[0, 0], x[R - 1, C - 1] = 0.0, -1.0
    x[1, 1] = 1e9                                                          # beyond the half range at any of these scales: clamps
    ip, it = ops.h2_split(torch.from_numpy(x).to(dev), plain=True, trans=True, scale=scale)
    want_p = x3_ref.image_h2(x, scale)
    want_t = x3_ref.image_h2(np.ascontiguousarray(x.T), scale)
    got_p = ip.buf.cpu().numpy().view(np.uint16).reshape(want_p.shape)
    got_t = it.buf.cpu().numpy().view(np.uint16).reshape(want_t.shape)
    assert (ip.rows, ip.K, it.rows, it.K) == (R, C, C, R)
    assert np.array_equal(got_p, want_p) and np.array_equal(got_t, want_t)


@pytest.mark.parametrize("M,N,K", [(300, 76, 50), (256, 256, 16), (1000, 516, 1153), (2176, 4096, 4800), (128, 23580, 1152)])
def test_gemm_h2_error_is_fp32_grade(dev, M, N, K):
    """Three f16 products of the two-half split against an fp64 product, next to the exact-fp32 MFMA kernel and the six-product bf16
    split on the same operands: all within a few 1e-7 of max |C| growing with sqrt(K); bias and the accumulate form included; static
    scales on A, the device-measured scale on B (as a weight gradient's dz operand takes it)."""
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    A = torch.randn((M, K), device=dev, generator=g)
    B = torch.randn((N, K), device=dev, generator=g) * 3e-4               # gradient-sized
    bias = torch.randn((N,), device=dev, generator=g) * 1e-3
    ref = A.double() @ B.double().t() + bias.double()
    c32 = ops.gemm_simple(A, B, transB=True, bias=bias)
    ia, _ = ops.x3_split(A)
    ib, _ = ops.x3_split(B)
    cx = ops.gemm_x3_grouped([dict(A=ia, B=ib, bias=bias)])[0]
    ha, _ = ops.h2_split(A, scale=_pow2_scale(A))
    hb, _ = ops.h2_split(B, dynamic=True)
    ch = ops.gemm_h2_grouped([dict(A=ha, B=hb, bias=bias)])[0]
    rel = lambda c, r: float((c.double() - r).abs().max() / r.abs().max())
    e32, ex, eh = rel(c32, ref), rel(cx, ref), rel(ch, ref)
    assert eh < max(2.0 * e32, 6e-7), (e32, ex, eh)
    _, hat = ops.h2_split(A.t().contiguous(), plain=False, trans=True, scale=_pow2_scale(A))
    assert torch.equal(hat.buf, ha.buf)                                    # the transposing pass writes the same image
    c0 = torch.randn((M, N), device=dev, generator=g) * 1e-2
    ch2 = ops.gemm_h2_grouped([dict(A=ha, B=hb, out=c0.clone(), beta=1.0)])[0]
    assert rel(ch2, ref - bias.double() + c0.double()) < max(2.0 * e32, 6e-7)
    # the device word holds max |B| (as float bits): the scale both the split and the product derive from it
    assert float(hb.dinv.view(torch.float32)[0]) == float(B.abs().max())
    # deterministic
    assert torch.equal(ch, ops.gemm_h2_grouped([dict(A=ha, B=hb, bias=bias)])[0])


def test_gemm_h2_against_the_three_product_oracle(dev):
    from oracle import x3_ref
    rs = np.random.RandomState(4)
    M, N, K = 96, 80, 528
    A = (rs.randn(M, K) * np.exp(rs.randn(M, K) * 0.5)).astype(np.float32)
    B = (rs.randn(N, K) * np.exp(rs.randn(N, K) * 0.5)).astype(np.float32)
    sa, sb = 2.0 ** 9, 2.0 ** 8
    ha, _ = ops.h2_split(torch.from_numpy(A).to(dev), scale=sa)
    hb, _ = ops.h2_split(torch.from_numpy(B).to(dev), scale=sb)
    got = ops.gemm_h2_grouped([dict(A=ha, B=hb)])[0].cpu().numpy().astype(np.float64)
    want = x3_ref.three_products(A, B, sa, sb)
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.all(np.abs(got - want) <= 2.0 ** -19 * bound)               # fp32 accumulation inside the MFMAs only
    exact = A.astype(np.float64) @ B.astype(np.float64).T
    assert np.all(np.abs(want - exact) <= 2.0 ** -21 * bound)


def test_gemm_h2_grouped_equals_single_and_split_k(dev):
    """Two problems in one launch give the bits of two launches; a long-K product that the launch splits along K (fix-up pass)
    keeps the scale factors (alpha and the device word are applied after the parts are summed)."""
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((1024, 1152), device=dev, generator=g)
    d1 = torch.randn((1024, 9432), device=dev, generator=g) * 1e-5
    d2 = torch.randn((1024, 14148), device=dev, generator=g) * 1e-5
    _, xt = ops.h2_split(x, plain=False, trans=True, scale=2.0 ** 10)      # [1152 rows, K = 1024]
    _, t1 = ops.h2_split(d1, plain=False, trans=True, dynamic=True)
    _, t2 = ops.h2_split(d2, plain=False, trans=True, dynamic=True)
    a, b = ops.gemm_h2_grouped([dict(A=xt, B=t1), dict(A=xt, B=t2)])
    a1 = ops.gemm_h2_grouped([dict(A=xt, B=t1)])[0]
    b1 = ops.gemm_h2_grouped([dict(A=xt, B=t2)])[0]
    assert torch.equal(a, a1) and torch.equal(b, b1)
    ref = x.double().t() @ d1.double()
    e32 = float((ops.gemm_simple(x, d1, transA=True).double() - ref).abs().max() / ref.abs().max())
    eh = float((a.double() - ref).abs().max() / ref.abs().max())
    assert eh < max(2.0 * e32, 6e-7), (eh, e32)
    # K = 19200 on 64 tiles: split along K
    A = torch.randn((1024, 19200), device=dev, generator=g)
    B = torch.randn((4096, 19200), device=dev, generator=g) * 1e-6
    ha, _ = ops.h2_split(A, scale=2.0 ** 10)
    hb, _ = ops.h2_split(B, dynamic=True)
    c = ops.gemm_h2_grouped([dict(A=ha, B=hb)])[0]
    ref = A.double() @ B.double().t()
    e32 = float((ops.gemm_simple(A, B, transB=True).double() - ref).abs().max() / ref.abs().max())
    eh = float((c.double() - ref).abs().max() / ref.abs().max())
    assert eh < max(2.0 * e32, 6e-7), (eh, e32)
