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


@pytest.mark.parametrize("t0", [0, 2])
def test_u8_projection_and_weight_gradient_on_two_f16_products(dev, t0):
    """The uint8 layer-0 products in their round-5 form (yt8m_gemm_h1x2_nt_ex): (q - 128) as a ONE-plane half image (exact) against
    the two-plane half image of (4/255) W^T under a device-measured scale, affine epilogue = fp64 x . W + b to fp32 rounding; and the
    weight gradient dW_x = x^T dz from the transposed half image of the frames against (r (.) dz)^T written by yt8m_h2_split_ex in
    the pass that also writes dz^T and both per-tile column sums."""
    import ctypes
    import yt8m_amd._lib as L
    from yt8m_amd.ops import _p, _stream
    from oracle import np_ref
    import yt8m_amd.seq_ops as seq_ops
    rs = np.random.RandomState(6)
    B, F, D, N = 32, 8, 1152, 512
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 0
    W = (rs.randn(D, N) * 0.05).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    qd, nfd, Wd, bd = (torch.from_numpy(a).to(dev) for a in (q, nf, W, bias))
    lib = L.lib()
    img = torch.empty(((F * B + 31) // 32) * (D // 16) * 1024, dtype=torch.uint8, device=dev)
    r = torch.empty((F * B,), dtype=torch.float32, device=dev)
    L.check(lib.yt8m_u8_frames_image_f16(_p(qd), _p(nfd), B, F, D, 1e-12, _p(img), None, _p(r), _stream()))
    alpha = 4.0 / 255.0
    _, w2 = ops.h2_split(Wd, plain=False, trans=True, scale=alpha, dynamic=True)
    cs = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.colsum(Wd, cs)
    T = F - t0
    rows = T * B - 5
    z = torch.full((rows, N), float("nan"), device=dev)
    ws = ops._workspace(dev)
    L.check(lib.yt8m_gemm_h1x2_nt_ex(rows, N, D, _p(img[(t0 * B // 32) * (D // 16) * 1024:]), 0, _p(w2.buf), 0, _p(z), N, _p(bd), 1.0,
                                     _p(w2.dinv), _p(r[t0 * B:]), _p(cs), seq_ops.U8_BETA, 0.0, _p(ws), ws.numel() * 4, _stream()))
    x64 = np_ref.dequant_l2norm_folded(q, nf).transpose(1, 0, 2).reshape(F * B, D)
    zr = x64[t0 * B:t0 * B + rows] @ W.astype(np.float64) + bias
    assert np.abs(z.cpu().numpy() - zr).max() < 2e-6 * max(1.0, np.abs(zr).max())
    # weight gradient of the time range [t0, F): dW_x = x^T dz
    M = T * B
    dz = (rs.randn(F * B, N) * 1e-4 * np.exp(rs.randn(F * B, 1))).astype(np.float32)
    dzd = torch.from_numpy(dz).to(dev)
    qT = torch.empty(((D + 31) // 32) * ((F * B + 15) // 16) * 1024, dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_u8_frames_image_t_f16(_p(qd), _p(nfd), B, F, D, _p(qT), _stream()))
    part = dzd[t0 * B:]
    word = ops.h2_absmax(part)
    nb = lambda rows_, K: lib.yt8m_x3_image_bytes(rows_, K) // 3 * 2
    dzT = torch.empty(nb(N, M), dtype=torch.uint8, device=dev)
    dzTs = torch.empty(nb(N, M), dtype=torch.uint8, device=dev)
    ntile = (M + 63) // 64
    cp = torch.empty((ntile, N), device=dev)
    cps = torch.empty((ntile, N), device=dev)
    L.check(lib.yt8m_h2_split_ex(_p(part), M, N, N, 1.0, _p(word), _p(r[t0 * B:]), None, _p(dzT), _p(dzTs), _p(cp), _p(cps), _stream()))
    dW = torch.full((D, N), 0.25, device=dev)
    qTk = ctypes.c_void_p(qT.data_ptr() + (t0 * B // 16) * 1024)
    csr = cps.sum(0).contiguous()
    L.check(lib.yt8m_gemm_h1x2_nt_ex(D, N, M, qTk, F * B // 16, _p(dzTs), 0, _p(dW), N, None, alpha, _p(word), None, _p(csr),
                                     seq_ops.U8_BETA / alpha, 1.0, _p(ws), ws.numel() * 4, _stream()))
    ref = 0.25 + x64[t0 * B:].T @ dz[t0 * B:].astype(np.float64)
    err = np.abs(dW.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 2e-6 * np.abs(ref - 0.25).max() + 1e-7, err
    # the per-tile column sums are those of the UNscaled source, plain and r-weighted
    assert float((cp.sum(0).double() - torch.from_numpy(dz[t0 * B:].astype(np.float64).sum(0)).to(dev)).abs().max()) < 1e-5 * float(np.abs(dz).max()) * M
    rw = r[t0 * B:].double().cpu().numpy()
    assert np.abs(cps.sum(0).double().cpu().numpy() - (dz[t0 * B:].astype(np.float64) * rw[:, None]).sum(0)).max() < 1e-5 * float(np.abs(dz).max()) * M
    # dz^T from the same pass serves the h-part gradient: h^T dz on three products
    h = (rs.rand(F * B, 256) * 2 - 1).astype(np.float32)
    _, hT = ops.h2_split(torch.from_numpy(h).to(dev)[t0 * B:], plain=False, trans=True, scale=8192.0)
    dzT_img = ops.H2Image(dzT, N, M, 1.0, word)
    dWh = ops.gemm_h2_grouped([dict(A=hT, B=dzT_img)])[0]
    refh = h[t0 * B:].astype(np.float64).T @ dz[t0 * B:].astype(np.float64)
    assert np.abs(dWh.cpu().numpy() - refh).max() <= 2e-6 * np.abs(refh).max() + 1e-9


def test_h2_rows_keep_their_own_precision(dev):
    """dx = dz . W^T with one power-of-two scale PER ROW of dz (yt8m_h2_rowscales / _split_rows / yt8m_gemm_h2_nt_ex): rows whose
    magnitudes differ by 24 decades -- vanishing time steps next to live ones, an all-zero row -- are each exact to 1e-6 of THEIR OWN
    scale; one scale for the whole operand would flush the small rows to zero (checked too: that is why dx takes this form)."""
    import yt8m_amd._lib as L
    from yt8m_amd.ops import _p, _stream
    lib = L.lib()
    g = torch.Generator(device=dev).manual_seed(12)
    M, N, K = 640, 1024, 4096
    A = torch.randn((M, K), device=dev, generator=g)
    mag = torch.pow(10.0, -torch.linspace(0, 24, M, device=dev))           # row r scaled by 10^-(24 r / M)
    A = A * mag[:, None]
    A[5] = 0.0
    W = torch.randn((N, K), device=dev, generator=g) * 0.03
    S = torch.empty(M, device=dev)
    inv = torch.empty(M, device=dev)
    L.check(lib.yt8m_h2_rowscales(_p(A), M, K, K, _p(S), _p(inv), _stream()))
    img = torch.empty(lib.yt8m_x3_image_bytes(M, K) // 3 * 2, dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_h2_split_rows(_p(A), M, K, K, _p(S), _p(img), _stream()))
    hw, _ = ops.h2_split(W, dynamic=True)
    C = torch.full((M, N), float("nan"), device=dev)
    ws = ops._workspace(dev)
    L.check(lib.yt8m_gemm_h2_nt_ex(M, N, K, _p(img), 0, _p(hw.buf), 0, _p(C), N, None, 1.0, None, _p(hw.dinv), _p(inv), 0.0, _p(ws),
                                   ws.numel() * 4, _stream()))
    ref = A.double() @ W.double().t()
    rowscale = ref.abs().max(dim=1).values
    err = (C.double() - ref).abs().max(dim=1).values
    live = rowscale > 0
    assert float((err[live] / rowscale[live]).max()) < 2e-6
    assert float(C[5].abs().max()) == 0.0 and float(S[5]) == 1.0
    rm = A.abs().max(dim=1).values
    assert bool(((rm[live] * S[live] >= 2.0 ** 13) & (rm[live] * S[live] < 2.0 ** 14)).all()) and torch.equal(inv, 1.0 / S)
    # one scale for the whole operand: the rows 2^-26 below the largest are gone
    ha, _ = ops.h2_split(A, dynamic=True)
    C1 = ops.gemm_h2_grouped([dict(A=ha, B=hw)])[0]
    assert float(C1[M - 1].abs().max()) == 0.0 and float(ref[M - 1].abs().max()) > 0.0


def test_h2_split_keeps_a_nan_a_nan(dev):
    """Finite and infinite values clamp into the half range (an operand that outgrew its scale degrades, it does not turn into inf);
    a NaN stays a NaN in both planes -- a diverged step must show as NaN downstream, as it would in fp32 -- exactly as the oracle's
    numpy arithmetic has it."""
    from oracle import x3_ref
    rs = np.random.RandomState(5)
    x = rs.randn(64, 48).astype(np.float32)
    x[3, 7], x[40, 0], x[10, 10] = np.nan, np.inf, -np.inf
    ip, _ = ops.h2_split(torch.from_numpy(x).to(dev), plain=True, trans=False, scale=4.0)
    want = x3_ref.image_h2(x, 4.0)
    got = ip.buf.cpu().numpy().view(np.uint16).reshape(want.shape)
    wf, gf = want.view(np.float16), got.view(np.float16)
    nan = np.isnan(wf)
    assert nan.sum() == 2 and np.array_equal(np.isnan(gf), nan)            # hi and lo of the one NaN element
    assert np.array_equal(got[~nan], want[~nan])
    assert np.isfinite(gf[~nan]).all()                                     # the infinities were clamped


def test_the_generic_transA_product_keeps_every_element_fp32_grade_and_the_dw_role_is_declared(dev):
    """the three-f16-product form (one scale per MATRIX: elements far below the matrix maximum lose bits) is taken only
    when the caller declares the weight-gradient role (ops.gemm(..., role="dw") = YT8M_GEMM_ROLE_DW in transA).  A generic
    transA product whose left operand has a column 2^-30 below the rest must keep that column's outputs to fp32 grade ON THEIR
    OWN SCALE; the declared product is held to the h2 contract (error relative to the whole matrix' scale)."""
    g = torch.Generator(device=dev).manual_seed(77)
    K, M, N = 9600, 1024, 4096                 # a weight gradient of the headline's recurrent stack: the library takes the f16 form
    A = torch.randn((K, M), device=dev, generator=g)
    A[:, 5] *= 2.0 ** -30
    Bm = torch.randn((K, N), device=dev, generator=g) * 1e-2
    ref = A.double().t() @ Bm.double()
    generic = ops.gemm(A, Bm, transA=True)
    row = lambda c: float((c[5].double() - ref[5]).abs().max() / ref[5].abs().max())
    assert row(generic) < 5e-6, row(generic)
    declared = ops.gemm(A, Bm, transA=True, role="dw")
    assert float((declared.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert row(declared) > 1e-4                # the role's contract, made visible: that column sits below one scale word's resolution
    with pytest.raises(ValueError):
        ops.gemm(A, Bm, transA=True, role="weights")
