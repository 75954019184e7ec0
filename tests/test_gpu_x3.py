"""-m gpu: the fp32-on-bf16-pipe GEMM (csrc/gemm_x3.hip), the uint8 input projection on it, the per-step exchange images of
the persistent recurrence, and the CU-masked stream / placement probe."""
import ctypes

import numpy as np
import pytest
import torch

import yt8m_amd._lib as L
import yt8m_amd.ops as ops
from yt8m_amd.ops import _p, _stream

pytestmark = pytest.mark.gpu


def _rel(c, ref):
    return float((c.double() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("M,N,K,kind", [
    (300, 77, 50, "randn"),          # partial tiles in both dimensions, K not a multiple of 16
    (256, 256, 16, "randn"),         # one tile, one K block
    (1000, 515, 1153, "randn"),
    (513, 4716, 2304, "wide"),       # 12 decades of dynamic range inside one reduction
    (2176, 4096, 4800, "randn"),     # the weight-gradient shape of the headline step (K shortened): split along K
])
def test_gemm_x3_error_is_fp32_grade(dev, M, N, K, kind):
    """Six bf16 products of the three-plane split against an fp64 product: the error is that of the fp32-MFMA kernel (both a few
    1e-7 of max|C|, growing with sqrt(K)), bias and the accumulate form included; the transposing split writes the same image."""
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    A = torch.randn((M, K), device=dev, generator=g)
    B = torch.randn((N, K), device=dev, generator=g)
    if kind == "wide":
        A = A * torch.exp(torch.randn((M, K), device=dev, generator=g) * 4)
        B = B * torch.exp(torch.randn((N, K), device=dev, generator=g) * 4)
    bias = torch.randn((N,), device=dev, generator=g)
    ref = A.double() @ B.double().t() + bias.double()
    c32 = ops.gemm_simple(A, B, transB=True, bias=bias)
    ia, _ = ops.x3_split(A)
    ib, _ = ops.x3_split(B)
    cx = ops.gemm_x3_grouped([dict(A=ia, B=ib, bias=bias)])[0]
    e32, ex = _rel(c32, ref), _rel(cx, ref)
    assert ex < max(4 * e32, 3e-7), (ex, e32)
    c0 = torch.randn((M, N), device=dev, generator=g)
    cacc = ops.gemm_x3_grouped([dict(A=ia, B=ib, out=c0.clone(), beta=1.0)])[0]
    assert _rel(cacc, ref - bias.double() + c0.double()) < max(4 * e32, 3e-7)
    _, iat = ops.x3_split(A.t().contiguous(), plain=False, trans=True)
    assert torch.equal(iat.buf, ia.buf)


def test_x3_split_is_exact(dev):
    """a1 + a2 + a3 reproduces every fp32 value exactly (also tiny, huge and negative ones); K padding is zero."""
    g = torch.Generator(device=dev).manual_seed(5)
    R, C = 70, 37
    x = torch.randn((R, C), device=dev, generator=g) * torch.exp(torch.randn((R, C), device=dev, generator=g) * 10)
    x[0, 0], x[1, 1], x[2, 2] = 0.0, -1.0, 3.0e38
    img, _ = ops.x3_split(x)
    KB = (C + 15) // 16
    blocks = img.buf.view(torch.bfloat16).view((R + 31) // 32, KB, 3, 32, 2, 8).float().cpu().numpy().astype(np.float64)
    xs = x.cpu().numpy().astype(np.float64)
    for r in range(R):
        sw = (r % 32 >> 3) & 1
        for kb in range(KB):
            row = np.concatenate([blocks[r // 32, kb, :, r % 32, sw, :], blocks[r // 32, kb, :, r % 32, sw ^ 1, :]], axis=1).sum(0)
            want = np.zeros(16)
            n = min(16, C - kb * 16)
            want[:n] = xs[r, kb * 16:kb * 16 + n]
            assert np.array_equal(row, want), (r, kb)


@pytest.mark.parametrize("R,C,scale", [(70, 37, 1.0), (64, 64, 1.0), (129, 1152, 4.0 / 255.0), (33, 16, 1.0)])
def test_x3_images_equal_the_oracle_bit_for_bit(dev, R, C, scale):
    """Plain and transposed operand images of yt8m_x3_split against oracle/x3_ref.image (numpy integer rounding): same bytes,
    padding rows / columns included, with and without the fp32 pre-scale."""
    from oracle import x3_ref
    rs = np.random.RandomState(R * 7 + C)
    x = (rs.randn(R, C) * np.exp(rs.randn(R, C) * 6)).astype(np.float32)
    x[0, 0], x[R - 1, C - 1] = 0.0, -1.0
    ip, it = ops.x3_split(torch.from_numpy(x).to(dev), plain=True, trans=True, scale=scale)
    want_p = x3_ref.image(x, scale=scale)
    want_t = x3_ref.image(np.ascontiguousarray((x * np.float32(scale)).astype(np.float32).T) if scale != 1.0 else np.ascontiguousarray(x.T))
    got_p = ip.buf.cpu().numpy().view(np.uint16).reshape(want_p.shape)
    got_t = it.buf.cpu().numpy().view(np.uint16).reshape(want_t.shape)
    assert (ip.rows, ip.K, it.rows, it.K) == (R, C, C, R)
    assert np.array_equal(got_p, want_p)
    assert np.array_equal(got_t, want_t)


def test_gemm_x3_against_the_six_product_oracle(dev):
    """The device's six-product sum against the oracle's (fp64 accumulation of the same six partial products): they differ only
    by the fp32 accumulation inside the MFMAs (held to 2^-19 sum |a| |b| here, K = 528) and the oracle itself is within 2^-24
    sum |a| |b| of the exact product (tests/test_oracle_x3.py)."""
    from oracle import x3_ref
    rs = np.random.RandomState(4)
    M, N, K = 96, 80, 528
    A = (rs.randn(M, K) * np.exp(rs.randn(M, K) * 2)).astype(np.float32)
    B = (rs.randn(N, K) * np.exp(rs.randn(N, K) * 2)).astype(np.float32)
    ia, _ = ops.x3_split(torch.from_numpy(A).to(dev))
    ib, _ = ops.x3_split(torch.from_numpy(B).to(dev))
    got = ops.gemm_x3_grouped([dict(A=ia, B=ib)])[0].cpu().numpy().astype(np.float64)
    want = x3_ref.six_products(A, B)
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.all(np.abs(got - want) <= 2.0 ** -19 * bound)       # a plain bf16 product would be off by ~2^-9 of it


def test_gemm_dispatch_grouped_equals_single(dev):
    """ops.gemm_grouped decides per problem and the x3 launch splits K per problem: a product launched alone or inside a group
    gives the same bits (the data-parallel path launches the weight-gradient products one by one, the plain step grouped)."""
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((1024, 1152), device=dev, generator=g)
    d1 = torch.randn((1024, 9432), device=dev, generator=g)
    d2 = torch.randn((1024, 14148), device=dev, generator=g)
    assert ops._x3_wins([(1152, 9432, 1024)], True, False)
    a = ops.gemm_grouped([dict(A=x, B=d1), dict(A=x, B=d2)], transA=True)
    b1 = ops.gemm(x, d1, transA=True)
    b2 = ops.gemm(x, d2, transA=True)
    assert torch.equal(a[0], b1) and torch.equal(a[1], b2)
    ref = x.double().t() @ d1.double()
    assert _rel(b1, ref) < 3e-6


@pytest.mark.parametrize("t0", [0, 3])
def test_u8_projection_on_the_x3_kernel(dev, t0):
    """(q - 128) as a one-plane image x three-plane image of (4/255) W^T with the affine epilogue = fp64 x.W + b to fp32
    rounding at D = 1152, for a time chunk that starts inside the image (t0 * B a multiple of 32) and ends on a partial row
    group; x_tm / r from the image pass are bit-identical to the dequantise kernel."""
    from oracle import np_ref
    import yt8m_amd.seq_ops as seq_ops
    rs = np.random.RandomState(6)
    B, F, D, N = 32, 7, 1152, 512
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 0
    W = (rs.randn(D, N) * 0.05).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    qd, nfd, Wd, bd = (torch.from_numpy(a).to(dev) for a in (q, nf, W, bias))
    lib = L.lib()
    img = torch.empty(((F * B + 31) // 32) * (D // 16) * 1024, dtype=torch.uint8, device=dev)
    r = torch.empty((F * B,), dtype=torch.float32, device=dev)
    xtm = torch.empty((F, B, D), dtype=torch.float32, device=dev)
    L.check(lib.yt8m_u8_frames_image(_p(qd), _p(nfd), B, F, D, 1e-12, _p(img), _p(xtm), _p(r), _stream()))
    assert torch.equal(xtm, ops.dequant_l2norm(qd, nfd).transpose(0, 1).contiguous())
    w3 = ops.x3_split(Wd, plain=False, trans=True, scale=4.0 / 255.0)[1]
    cs = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.colsum(Wd, cs)
    T = F - t0
    rows = T * B - 5                                     # stop inside the last row group
    z = torch.full((rows, N), float("nan"), device=dev)
    ws = ops._workspace(dev)
    L.check(lib.yt8m_gemm_x1x3_nt(rows, N, D, _p(img[(t0 * B // 32) * (D // 16) * 1024:]), _p(w3.buf), _p(z), N, _p(bd), _p(r[t0 * B:]),
                                  _p(cs), seq_ops.U8_BETA, _p(ws), ws.numel() * 4, _stream()))
    x64 = np_ref.dequant_l2norm_folded(q, nf).transpose(1, 0, 2).reshape(F * B, D)[t0 * B:t0 * B + rows]
    zr = x64 @ W.astype(np.float64) + bias
    assert np.abs(z.cpu().numpy() - zr).max() < 2e-6 * max(1.0, np.abs(zr).max())


def test_persistent_recurrence_step_images_vs_two_images(dev):
    """One exchange image per step against two alternating images.  Backward: the same arithmetic in the same order (plain
    XCD-L2-shared fetch instead of sc0 sc1 loads), so results are bit-identical.  Forward: with step images the recurrent product
    runs as six bf16 products of three-plane splits (lstm_persist_fwd_x3_kernel) -- fp32-grade, not bit-identical: both forms
    are compared with an fp64 recurrence."""
    lib = L.lib()
    B, F, H = 128, 12, 1024
    if not lib.yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available on this device")
    g = torch.Generator(device=dev).manual_seed(9)
    z0 = torch.randn((F, B, 4 * H), device=dev, generator=g) * 0.3
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.06
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.01
    # fp64 recurrence (BasicLSTMCell, forget_bias 1)
    c = torch.zeros((B, H), dtype=torch.float64, device=dev)
    h = torch.zeros((B, H), dtype=torch.float64, device=dev)
    ref = []
    for t in range(F):
        zz = z0[t].double() + h @ Wh.double()
        i, j, f, o = zz.split(H, dim=1)
        c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
        ref.append(h)
    ref = torch.stack(ref)
    res, fwd0 = [], None
    for nbytes in (lib.yt8m_lstm_persist_workspace_bytes(B, H), lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)):
        pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        z = z0.clone()
        cs = torch.zeros((F + 1, B, H), device=dev)
        hs = torch.zeros((F + 1, B, H), device=dev)
        out = torch.empty((F, B, H), device=dev)
        L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(pws), nbytes, _stream()))
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        assert float((out.double() - ref).abs().max()) < 5e-6
        if fwd0 is None:
            fwd0 = (z.clone(), cs.clone())
        else:
            assert float((out - res[0][0]).abs().max()) < 5e-6 and float((cs - fwd0[1]).abs().max()) < 1e-5
        zb, csb = fwd0                                    # the backward launches of both modes get the SAME saved activations
        dz = torch.empty((F, B, 4 * H), device=dev)
        work = torch.zeros((4, B, H), device=dev)
        L.check(lib.yt8m_lstm_persist_bwd(_p(zb), _p(Wh), 4 * H, _p(csb), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H, _p(pws), nbytes,
                                          _stream()))
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        res.append((out.clone(), dz.clone(), work.clone()))
    assert lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F) > lib.yt8m_lstm_persist_workspace_bytes(B, H)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert float(res[0][1].abs().max()) > 0


def test_cu_masked_stream_confines_workgroups(dev):
    """A stream created with the low 128 mask bits set runs on 16 CUs of each of the 8 XCDs (mask bit i -> XCD i % 8); the
    placement probe sees every CU from an unmasked stream."""
    lib = L.lib()
    props = torch.cuda.get_device_properties(dev)
    if props.multi_processor_count != 256:
        pytest.skip("mask layout measured on the 256-CU part")

    def cus(stream_ptr, sync):
        out = torch.full((4096, 2), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        L.check(lib.yt8m_probe_placement(_p(out), 4096, 2000, stream_ptr))
        sync()
        o = out.cpu().numpy()
        return {(int(x), (int(h) >> 8) & 0xFF) for x, h in o}

    allc = cus(_stream(), torch.cuda.synchronize)
    assert len(allc) == 256
    words = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 4 + [0] * 4))
    h = ctypes.c_void_p()
    L.check(lib.yt8m_stream_create_cu_mask(words, 8, ctypes.byref(h)))
    try:
        s = torch.cuda.ExternalStream(h.value, device=dev)
        half = cus(ctypes.c_void_p(h.value), s.synchronize)
        assert len(half) == 128 and half < allc
        per = {}
        for x, c in half:
            per[x] = per.get(x, 0) + 1
        assert sorted(per) == list(range(8)) and set(per.values()) == {16}
    finally:
        L.check(lib.yt8m_stream_destroy(ctypes.c_void_p(h.value)))


def test_forward_wavefront_and_cu_budget_give_the_same_results(dev, monkeypatch, honour_lstm_chunks):
    """The opt-in forward wavefront (half-chip forward recurrences of neighbouring layers side by side, finer time chunks, the
    projections on streams of their own) and an explicit CU budget change WHERE the recurrences run, not what they compute: same
    outputs and gradients as the default placement to fp32 rounding; the placement counters see every launch."""
    import ctypes
    import yt8m_amd.seq_ops as seq_ops
    from test_gpu_round2 import _stack_run
    lib = L.lib()
    B, F, D, H = 128, 24, 96, 1024
    if not lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, H):
        pytest.skip("bf16-pipe forward recurrence not available for this shape / device")
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(3), dtype=torch.int32)
    nf[0], nf[1] = F, 0
    nl, nw, off = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    L.check(lib.yt8m_lstm_persist_placement_stats(None, None, None, 1))
    ref, gref, _, _ = _stack_run(dev, B, F, D, H, 2, 2, nf, True)
    L.check(lib.yt8m_lstm_persist_placement_stats(ctypes.byref(nl), ctypes.byref(nw), ctypes.byref(off), 1))
    assert nl.value == 8 and nw.value == 4 * 256 + 4 * 128 and 0 <= off.value <= nw.value   # 2 layers x 2 chunks, forward + backward
    monkeypatch.setattr(seq_ops, "FWD_WAVEFRONT", True)
    monkeypatch.setattr(seq_ops, "FWD_WAVEFRONT_CHUNKS", 3)
    a, ga, _, _ = _stack_run(dev, B, F, D, H, 2, 2, nf, True)
    L.check(lib.yt8m_lstm_persist_placement_stats(ctypes.byref(nl), ctypes.byref(nw), None, 1))
    assert nl.value == 6 + 4 and nw.value == 10 * 128          # three half-chip forward chunks per layer, two backward ones
    for u, v in zip(a + ga, ref + gref):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 5e-6
    monkeypatch.setattr(seq_ops, "FWD_WAVEFRONT", False)
    try:
        L.check(lib.yt8m_lstm_persist_set_cus(128, 128))
        b, gb, _, _ = _stack_run(dev, B, F, D, H, 2, 2, nf, True)
    finally:
        L.check(lib.yt8m_lstm_persist_set_cus(-1, -1))
    for u, v in zip(b + gb, ref + gref):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 5e-6


def test_persistent_partition_defaults(dev, monkeypatch, honour_lstm_chunks):
    """The partition the product picks for the persistent kernels (ONE forward launch per layer, the library's own backward parts,
    whatever `chunks` the caller passes) changes how many launches run, not the results: same outputs and gradients as the caller's
    2 + 2 partition to fp32 rounding; the placement counters see 2 forward + 2 x (parts) backward launches."""
    import ctypes
    import yt8m_amd.seq_ops as seq_ops
    from test_gpu_round2 import _stack_run
    lib = L.lib()
    B, F, D, H = 128, 24, 96, 1024
    if not lib.yt8m_lstm_persist_supported(B, H):
        pytest.skip("persistent recurrence not available for this shape / device")
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(5), dtype=torch.int32)
    nf[0], nf[1] = F, 0
    nl = ctypes.c_int64(0)
    ref, gref, _, _ = _stack_run(dev, B, F, D, H, 2, 2, nf, True)          # honour_lstm_chunks: the caller's chunks = 2
    # plain assignments: honour_lstm_chunks restores the saved values at teardown (monkeypatch would be undone AFTER that fixture and
    # put its zeros back, switching the native stack off for every later test of the process)
    seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS = 1, 3
    L.check(lib.yt8m_lstm_persist_placement_stats(None, None, None, 1))
    a, ga, _, _ = _stack_run(dev, B, F, D, H, 2, 2, nf, True)
    L.check(lib.yt8m_lstm_persist_placement_stats(ctypes.byref(nl), None, None, 1))
    # one forward launch per layer; backward: the library's own partition for F >= 12 -- four parts 2 : 2 : 1 : 1 since round 4
    # (three parts 3 : 2 : 1 before) -- for each of the two layers
    nb = ctypes.c_int(0)
    desc = seq_ops._stack_desc(B, F, D, H, 2, False, 1.0, True)
    L.check(lib.yt8m_lstm_stack_partition(ctypes.byref(desc), None, ctypes.byref(nb)))
    assert nb.value == 4 and nl.value == 2 + 2 * nb.value
    for u, v in zip(a + ga, ref + gref):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 5e-6


def test_bf16_pipe_forward_recurrence_over_the_full_sequence(dev):
    """F = 300 steps (the BASELINE length) of the forward recurrence with the recurrent product as six bf16 products, against an
    fp64 recurrence on the same inputs: the error does not grow along the sequence (every h_t and the final c within 2e-5),
    ragged lengths included (copy-through of the state, zero output after the last frame)."""
    lib = L.lib()
    B, F, H = 64, 300, 1024
    if not lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, H):
        pytest.skip("bf16-pipe forward recurrence not available for this shape / device")
    g = torch.Generator(device=dev).manual_seed(21)
    z0 = torch.randn((F, B, 4 * H), device=dev, generator=g) * 0.5
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.08
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    nf[0], nf[1], nf[2] = F, 0, 1
    c = torch.zeros((B, H), dtype=torch.float64, device=dev)
    h = torch.zeros((B, H), dtype=torch.float64, device=dev)
    Wd = Wh.double()
    outs = []
    for t in range(F):
        zz = z0[t].double() + h @ Wd
        i, j, f, o = zz.split(H, dim=1)
        c1 = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        h1 = torch.tanh(c1) * torch.sigmoid(o)
        live = (t < nf).unsqueeze(1)
        c = torch.where(live, c1, c)
        h = torch.where(live, h1, h)
        outs.append(torch.where(live, h1, torch.zeros_like(h1)))
    ref = torch.stack(outs)
    nbytes = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)
    pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    z = z0.clone()
    cs = torch.zeros((F + 1, B, H), device=dev)
    hs = torch.zeros((F + 1, B, H), device=dev)
    out = torch.empty((F, B, H), device=dev)
    L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nf), 0, F, B, H, 1.0, _p(pws), nbytes, _stream()))
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
    err_t = (out.double() - ref).abs().amax(dim=(1, 2))
    assert float(err_t.max()) < 2e-5, float(err_t.max())
    assert float(err_t[-50:].max()) < 4 * float(err_t[:50].max()) + 1e-6         # no growth along the sequence
    assert float((cs[F].double() - c).abs().max()) < 2e-5 and float((hs[F].double() - h).abs().max()) < 2e-5
