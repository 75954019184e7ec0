"""-m gpu: parity of every C-ABI kernel (called through the ctypes boundary) against the oracle on seeded
inputs, including the edge cases the domain has: ragged / empty num_frames, non-multiple-of-tile shapes,
unaligned leading dimensions, saturated probabilities, ties."""
import numpy as np
import pytest
import torch

from oracle import np_ref
import yt8m_amd.ops as ops
import yt8m_amd.seq_ops as seq_ops
import yt8m_amd._lib as L

pytestmark = pytest.mark.gpu


def D(a, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dtype)


def H(t):
    return t.detach().cpu().numpy().astype(np.float64)


GEMM_SHAPES = [(1, 1, 1), (5, 7, 3), (128, 128, 16), (130, 257, 33), (64, 300, 1152), (257, 129, 100), (3, 4716 * 3, 40)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("tA,tB", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_gemm_all_layouts(dev, M, N, K, tA, tB):
    rs = np.random.RandomState(M * 7 + N * 3 + K + tA * 2 + tB)
    A = rs.randn(*((K, M) if tA else (M, K))).astype(np.float32)
    B = rs.randn(*((N, K) if tB else (K, N))).astype(np.float32)
    bias = rs.randn(N).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (B.T if tB else B).astype(np.float64)
    scale = np.abs(A).max() * np.abs(B).max() * K
    C = ops.gemm(D(A, dev), D(B, dev), transA=bool(tA), transB=bool(tB))
    assert np.abs(H(C) - ref).max() <= 2e-6 * scale
    Cs = ops.gemm_simple(D(A, dev), D(B, dev), transA=bool(tA), transB=bool(tB))       # plain launch, same contract
    assert np.abs(H(Cs) - ref).max() <= 2e-6 * scale
    C2 = ops.gemm(D(A, dev), D(B, dev), transA=bool(tA), transB=bool(tB), bias=D(bias, dev))
    assert np.abs(H(C2) - (ref + bias)).max() <= 2e-6 * scale
    ops.gemm(D(A, dev), D(B, dev), out=C2, transA=bool(tA), transB=bool(tB), beta=1.0)     # accumulate
    assert np.abs(H(C2) - (2 * ref + bias)).max() <= 4e-6 * scale


def test_gemm_strided_views_and_transpose_detection(dev):
    """Unaligned leading dimensions (scalar-load path) + an A = I check with an ASYMMETRIC B (row/col swap detector)."""
    rs = np.random.RandomState(0)
    big = rs.randn(70, 91).astype(np.float32)
    A = D(big, dev)[3:40, 5:70]                # ld = 91 (odd), offset pointer => unaligned
    Bm = D(rs.randn(65, 37).astype(np.float32), dev)
    out = torch.zeros((37, 50), device=dev)[:, 2:39]
    ops.gemm(A, Bm, out=out)
    assert np.abs(H(out) - big[3:40, 5:70].astype(np.float64) @ H(Bm)).max() < 1e-4
    n = 96
    Bas = np.arange(n * n, dtype=np.float32).reshape(n, n) / 100.0
    C = ops.gemm(torch.eye(n, device=dev), D(Bas, dev))
    assert np.array_equal(H(C), Bas.astype(np.float64))
    # fp32 MFMA is an exact fmaf chain: identity times anything is bit exact, and K-order is sequential
    C = ops.gemm(D(Bas, dev), torch.eye(n, device=dev), transA=True)
    assert np.array_equal(H(C), Bas.T.astype(np.float64))


def test_gemm_grouped_persistent_and_splitk(dev):
    """Persistent grouped launch: several problems in one tile space; covers whole rounds (T > 768 tiles), the
    split-K remainder + deterministic fix-up, the T < slots split, bias / beta per problem and ragged edges."""
    rs = np.random.RandomState(11)
    x = rs.randn(1024, 256).astype(np.float32)
    Wg = rs.randn(256, 14148).astype(np.float32)        # 8 x 111 tiles
    We = rs.randn(256, 9432).astype(np.float32)         # 8 x 74 tiles  -> 1480 tiles = 1 round + 712 remainder (S = 1)
    be = rs.randn(9432).astype(np.float32)
    Zg, Ze = ops.gemm_grouped([dict(A=D(x, dev), B=D(Wg, dev)), dict(A=D(x, dev), B=D(We, dev), bias=D(be, dev))])
    x64 = x.astype(np.float64)
    assert np.abs(H(Zg) - x64 @ Wg).max() < 2e-3 and np.abs(H(Ze) - (x64 @ We + be)).max() < 2e-3
    # dW shapes: 9 x 111 + 9 x 74 = 1665 tiles = 2 rounds + 129 remainder tiles split S = 5 ways (K = 1024 -> 64 steps)
    dZg = rs.randn(1024, 14148).astype(np.float32)
    dZe = rs.randn(1024, 9432).astype(np.float32)
    x2 = rs.randn(1024, 1152).astype(np.float32)
    g0 = torch.full((1152, 14148), 7.0, device=dev)
    g1 = torch.full((1152, 9432), 1.0, device=dev)
    ops.gemm_grouped([dict(A=D(x2, dev), B=D(dZg, dev), out=g0, beta=0.0), dict(A=D(x2, dev), B=D(dZe, dev), out=g1, beta=1.0)], transA=True)
    r0 = x2.astype(np.float64).T @ dZg
    r1 = x2.astype(np.float64).T @ dZe + 1.0
    assert np.abs(H(g0) - r0).max() < 5e-3 and np.abs(H(g1) - r1).max() < 5e-3
    g0b = torch.empty_like(g0)
    ops.gemm_grouped([dict(A=D(x2, dev), B=D(dZg, dev), out=g0b)], transA=True)
    ops.gemm_grouped([dict(A=D(x2, dev), B=D(dZg, dev), out=g0)], transA=True)
    assert torch.equal(g0, g0b)                        # fixed-order fix-up: bitwise reproducible
    # few tiles, long K: split-K fills the chip (LstmModel head at B = 128)
    a = rs.randn(100, 4096).astype(np.float32)
    w = rs.randn(4096, 1000).astype(np.float32)
    bb = rs.randn(1000).astype(np.float32)
    c = ops.gemm(D(a, dev), D(w, dev), bias=D(bb, dev))
    assert np.abs(H(c) - (a.astype(np.float64) @ w + bb)).max() < 5e-3
    assert np.abs(H(ops.gemm_simple(D(a, dev), D(w, dev), bias=D(bb, dev))) - H(c)).max() < 2e-3


def test_gemm_batched(dev):
    rs = np.random.RandomState(1)
    A = rs.randn(5, 30, 8).astype(np.float32)       # [b, F, A]
    X = rs.randn(5, 30, 70).astype(np.float32)      # [b, F, H]
    C = ops.gemm_batched(D(A, dev), D(X, dev), transA=True)
    assert np.abs(H(C) - np.einsum("bfa,bfh->bah", A.astype(np.float64), X.astype(np.float64))).max() < 1e-4
    dC = rs.randn(5, 8, 70).astype(np.float32)
    dA = ops.gemm_batched(D(X, dev), D(dC, dev), transB=True)
    assert np.abs(H(dA) - np.einsum("bfh,bah->bfa", X.astype(np.float64), dC.astype(np.float64))).max() < 1e-4


def test_gemm_error_paths(dev):
    a = torch.zeros(4, 4, device=dev)
    with pytest.raises(ValueError):
        ops.gemm(a, torch.zeros(5, 4, device=dev))
    with pytest.raises(ValueError):
        ops.gemm(a, a, out=torch.zeros(3, 4, device=dev))
    with pytest.raises(ValueError):
        ops.gemm(a, a, beta=1.0)
    assert ops.gemm(torch.zeros(0, 4, device=dev), a).shape == (0, 4)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 8, 16])
def test_moe_mix_fwd_bwd(dev, M):
    rs = np.random.RandomState(M)
    B, V = 7, 53
    Zg = (rs.randn(B, V * (M + 1)) * 3).astype(np.float32)
    Ze = (rs.randn(B, V * M) * 3).astype(np.float32)
    Zg[0, :M + 1] = [80.0] + [-80.0] * M        # saturated softmax / sigmoid
    Ze[0, :M] = 90.0
    Ze[1, :M] = -90.0
    g = np_ref.softmax(Zg.astype(np.float64).reshape(B, V, M + 1), axis=2)
    e = np_ref.sigmoid(Ze.astype(np.float64).reshape(B, V, M))
    p_ref = (g[:, :, :M] * e).sum(2)
    p = ops.moe_mix_fwd(D(Zg, dev), D(Ze, dev), V, M)
    assert np.abs(H(p) - p_ref).max() < 2e-6
    dp = rs.randn(B, V).astype(np.float32)
    epad = np.concatenate([e, np.zeros((B, V, 1))], axis=2)
    dG = dp[:, :, None] * g * (epad - p_ref[:, :, None])
    dE = dp[:, :, None] * g[:, :, :M] * e * (1 - e)
    zg, ze = D(Zg, dev), D(Ze, dev)
    ops.moe_mix_bwd_(zg, ze, D(dp, dev), V, M)
    assert np.abs(H(zg) - dG.reshape(B, -1)).max() < 2e-6
    assert np.abs(H(ze) - dE.reshape(B, -1)).max() < 2e-6


@pytest.mark.parametrize("kind", ["sigmoid", "relu", "relu6", "tanh", "elu"])
def test_activations(dev, kind):
    x = np.linspace(-9, 9, 1001).astype(np.float32)
    x64 = x.astype(np.float64)
    ref = {"sigmoid": 1 / (1 + np.exp(-x64)), "relu": np.maximum(x64, 0), "relu6": np.clip(x64, 0, 6), "tanh": np.tanh(x64),
           "elu": np.where(x64 > 0, x64, np.exp(np.minimum(x64, 0)) - 1)}[kind]
    y = ops.act_fwd(kind, D(x, dev))
    assert np.abs(H(y) - ref).max() < 1e-6
    dref = {"sigmoid": ref * (1 - ref), "relu": (x64 > 0) * 1.0, "relu6": ((x64 > 0) & (x64 < 6)) * 1.0,
            "tanh": 1 - ref ** 2, "elu": np.where(x64 > 0, 1.0, ref + 1)}[kind]
    dx = ops.act_bwd(kind, y, torch.full_like(y, 2.0))
    assert np.abs(H(dx) - 2 * dref).max() < 1e-5


def test_colsum(dev):
    rs = np.random.RandomState(2)
    for rows, cols in [(1, 1), (1024, 9432 // 8), (33, 130), (300, 64), (0, 5), (38400, 64), (9000, 8), (4097, 200)]:  # last 3: row-split path
        X = rs.randn(rows, cols).astype(np.float32)
        out = torch.full((cols,), 5.0, device=dev)
        ops.colsum(D(X, dev).reshape(rows, cols), out, beta=0.0)
        assert np.abs(H(out) - X.astype(np.float64).sum(0)).max() < 1e-4 * max(1.0, rows / 1024.0)
        ops.colsum(D(X, dev).reshape(rows, cols), out, beta=1.0)
        assert np.abs(H(out) - 2 * X.astype(np.float64).sum(0)).max() < 2e-4 * max(1.0, rows / 1024.0)


@pytest.mark.parametrize("label_kind", ["bool", "u8", "f32"])
def test_cross_entropy_fwd_bwd(dev, label_kind):
    rs = np.random.RandomState(3)
    B, V = 9, 4716
    p = (rs.rand(B, V) * 0.98 + 0.01).astype(np.float32)
    p[0, 0], p[0, 1], p[0, 2], p[0, 3] = 0.0, 1.0, 1.0, 0.0          # eps guards (W/losses.py:115)
    y = rs.rand(B, V) < 3.4 / V
    y[0, :4] = [True, True, False, False]
    w = rs.rand(B).astype(np.float32)
    if label_kind == "f32":
        ylab = np_ref.label_smoothing(y, 0.1)
        yd = D(ylab.astype(np.float32), dev)
    else:
        ylab = y
        yd = torch.from_numpy(y).to(dev) if label_kind == "bool" else torch.from_numpy(y.astype(np.uint8)).to(dev)
    p64 = p.astype(np.float64)
    for weights in (None, w):
        ref = np_ref.cross_entropy_loss(p64, ylab.astype(np.float64), weights)
        dref = np_ref.cross_entropy_loss_bwd(p64, ylab.astype(np.float64), weights, upstream=0.7)
        wd = None if weights is None else D(weights, dev)
        loss, dp = ops.xent_fwd(D(p, dev), yd, wd, want_dp=True, upstream=0.7)
        assert abs(float(loss) - ref) < 2e-5 * abs(ref)
        # 1/(p+eps) amplifies fp32 rounding of (p + eps): compare relative to each element
        assert np.abs(H(dp) - dref).max() <= 2e-5 * np.abs(dref).max()
        up = torch.tensor([0.35], device=dev)
        dp2 = ops.xent_bwd(D(p, dev), yd, wd, up, upstream=2.0)
        assert np.abs(H(dp2) - dref).max() <= 2e-5 * np.abs(dref).max()
    with pytest.raises(ValueError):
        ops.xent_fwd(D(p, dev), yd[:, :-1])
    with pytest.raises(ValueError):
        ops.xent_fwd(torch.zeros(0, 5, device=dev), torch.zeros(0, 5, device=dev))


def test_l2norm_fwd_bwd(dev):
    rs = np.random.RandomState(4)
    for rows, cols in [(5, 1152), (3, 7), (64, 128), (2, 73728), (3, 8196), (1, 1)]:     # >= 8192 cols: 1024-thread float4 kernel
        x = rs.randn(rows, cols).astype(np.float32)
        if rows > 1:
            x[1] = 0.0                                 # zero rows stay zero (A.9)
        dy = rs.randn(rows, cols).astype(np.float32)
        y = ops.l2norm_fwd(D(x, dev))
        assert np.abs(H(y) - np_ref.l2_normalize(x.astype(np.float64))).max() < 1e-6
        dx = ops.l2norm_bwd(D(x, dev), D(dy, dev))
        ref = np_ref.l2_normalize_bwd(x.astype(np.float64), dy.astype(np.float64))
        assert np.abs(H(dx) - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())


def test_dequant_l2norm(dev):
    rs = np.random.RandomState(5)
    B, F, Dm = 4, 12, 1152
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    q[0, 0] = 0
    q[0, 1] = 255
    nf = np.array([12, 1, 7, 0], dtype=np.int32)
    ref = np_ref.dequant_l2norm_folded(q, nf)
    x = ops.dequant_l2norm(torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev))
    assert np.abs(H(x) - ref).max() < 1e-6
    assert (H(x)[1, 1:] == 0).all() and (H(x)[3] == 0).all()          # padding rows are exactly 0 (readers.py:186)
    x2 = ops.dequant_l2norm(torch.from_numpy(q).to(dev), None)
    assert np.abs(H(x2) - np_ref.dequant_l2norm_folded(q)).max() < 1e-6
    xm = ops.dequant_mean_l2norm(torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev))
    refm = np.stack([np_ref.l2_normalize(np_ref.dequantize(q[b, :nf[b]]).mean(0)) if nf[b] > 0 else np.zeros(Dm) for b in range(B)])
    assert np.abs(H(xm) - refm).max() < 1e-6
    with pytest.raises(TypeError):
        ops.dequant_l2norm(torch.zeros(2, 3, 4, device=dev))


def test_topk_rows(dev):
    rs = np.random.RandomState(6)
    B, V = 33, 4716
    p = rs.rand(B, V).astype(np.float32)
    p[0, 10] = p[0, 20] = p[0, 5] = 2.0                                # ties -> lower index first
    vals, idx = ops.topk_rows(D(p, dev), 20)
    order = np.lexsort((np.arange(V)[None, :].repeat(B, 0), -p), axis=1)[:, :20]
    assert np.array_equal(idx.cpu().numpy(), order)                    # index selection is integer work: bit exact
    assert np.array_equal(H(vals), np.take_along_axis(p, order, 1).astype(np.float64))
    assert idx[0, :3].tolist() == [5, 10, 20]
    v2, i2 = ops.topk_rows(D(p[:, :7], dev), 20)                       # k > V clamps like eval_util.top_k_triplets
    assert v2.shape == (B, 7)
    # rows with fewer than k scores above -inf (NaN / -inf entries): the indices stay valid and distinct
    bad = np.full((3, 40), np.nan, dtype=np.float32)
    bad[1, :] = -np.inf
    bad[2, 7], bad[2, 30] = 0.5, 0.9
    v3, i3 = ops.topk_rows(D(bad, dev), 20)
    i3 = i3.cpu().numpy()
    assert i3.min() >= 0 and i3.max() < 40 and all(len(set(r.tolist())) == 20 for r in i3)
    assert i3[2, :2].tolist() == [30, 7] and i3[0, :3].tolist() == [0, 1, 2]


@pytest.mark.parametrize("B,F,Din,Hh", [(5, 9, 6, 4), (37, 6, 20, 128), (64, 4, 16, 256)])
def test_lstm_layer_fwd_bwd(dev, B, F, Din, Hh):
    """Whole-layer recurrence (ragged num_frames incl. 0 and F) vs the oracle, gradients vs torch autograd fp64.
    H = 4 takes the generic per-step GEMM path, H = 128 / 256 the fused one-launch-per-step kernels (lstm_fused.hip),
    with B not a multiple of the 32 / 16 row tiles."""
    from oracle import torch_ref
    from yt8m_amd.variables import reset_default_graph, xavier_uniform, zeros
    rs = np.random.RandomState(7)
    x = rs.randn(B, F, Din).astype(np.float32)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[:5] = [F, 1, min(4, F), 0, F]
    W = (rs.randn(Din + Hh, 4 * Hh) * (0.4 if Hh < 64 else 0.08)).astype(np.float32)
    b = (rs.randn(4 * Hh) * 0.1).astype(np.float32)
    g = reset_default_graph(device=dev)
    g.begin_step()
    Wv = g.get_variable("w", W.shape, xavier_uniform)
    bv = g.get_variable("b", b.shape, zeros)
    g.finalize()
    Wv.data.copy_(D(W, dev)); bv.data.copy_(D(b, dev))
    xt = D(x, dev).transpose(0, 1).contiguous().requires_grad_(True)
    out, c, h = seq_ops.lstm_layer(xt, Wv, bv, torch.from_numpy(nf).to(dev))
    ro, fin = np_ref.dynamic_rnn_lstm(x.astype(np.float64), nf, [(W.astype(np.float64), b.astype(np.float64))])
    assert np.abs(H(out).transpose(1, 0, 2) - ro).max() < 1e-5
    assert np.abs(H(c) - fin[0][0]).max() < 1e-5 and np.abs(H(h) - fin[0][1]).max() < 1e-5
    go, gc, gh = rs.randn(F, B, Hh).astype(np.float32), rs.randn(B, Hh).astype(np.float32), rs.randn(B, Hh).astype(np.float32)
    ((out * D(go, dev)).sum() + (c * D(gc, dev)).sum() + (h * D(gh, dev)).sum()).backward()
    tx = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    tW = torch.from_numpy(W.astype(np.float64)).requires_grad_(True)
    tb = torch.from_numpy(b.astype(np.float64)).requires_grad_(True)
    to, tc, th = torch_ref.lstm_stack(tx, torch.from_numpy(nf), [(tW, tb)])
    ((to * torch.from_numpy(go.astype(np.float64)).transpose(0, 1)).sum() + (tc[0] * torch.from_numpy(gc.astype(np.float64))).sum()
     + (th[0] * torch.from_numpy(gh.astype(np.float64))).sum()).backward()
    assert np.abs(H(Wv.grad) - tW.grad.numpy()).max() < 1e-4
    assert np.abs(H(bv.grad) - tb.grad.numpy()).max() < 1e-4
    assert np.abs(H(xt.grad).transpose(1, 0, 2) - tx.grad.numpy()).max() < 1e-4


def test_attention_and_assignment_softmax(dev):
    rs = np.random.RandomState(8)
    B, F, A = 4, 11, 3
    act = (rs.randn(B, F, A) * 2).astype(np.float32)
    nf = np.array([11, 1, 5, 11], dtype=np.int32)
    mask = (np.arange(F)[None, :] < nf[:, None]).astype(np.float64)
    sm = np_ref.softmax(act.astype(np.float64), axis=1) * mask[:, :, None]
    wref = sm / sm.sum(1, keepdims=True)
    a = D(act, dev).requires_grad_(True)
    w = seq_ops.attention_weights(a, torch.from_numpy(nf).to(dev))
    assert np.abs(H(w) - wref).max() < 1e-6
    dw = rs.randn(B, F, A).astype(np.float32)
    w.backward(D(dw, dev))
    ta = torch.from_numpy(act.astype(np.float64)).requires_grad_(True)
    tw = torch.softmax(ta, 1) * torch.from_numpy(mask)[:, :, None]
    (tw / tw.sum(1, keepdim=True)).backward(torch.from_numpy(dw.astype(np.float64)))
    assert np.abs(H(a.grad) - ta.grad.numpy()).max() < 1e-5
    # num_frames = 0 -> 0/0 = NaN in the reference (no guard): same here
    w0 = seq_ops.attention_weights(D(act[:1], dev), torch.zeros(1, dtype=torch.int32, device=dev))
    assert torch.isnan(w0).all()
    # NetVLAD assignment softmax over K with frame mask
    K = 64
    s = (rs.randn(B, F, K) * 3).astype(np.float32)
    sd = D(s, dev).requires_grad_(True)
    av = seq_ops.masked_softmax_rows(sd, torch.from_numpy(nf).to(dev))
    aref = np_ref.softmax(s.astype(np.float64), axis=2) * mask[:, :, None]
    assert np.abs(H(av) - aref).max() < 1e-6
    da = rs.randn(B, F, K).astype(np.float32)
    av.backward(D(da, dev))
    ts = torch.from_numpy(s.astype(np.float64)).requires_grad_(True)
    (torch.softmax(ts, 2) * torch.from_numpy(mask)[:, :, None]).backward(torch.from_numpy(da.astype(np.float64)))
    assert np.abs(H(sd.grad) - ts.grad.numpy()).max() < 1e-5


def _netvlad_pool_ref(q, nf, Wc, bc, dagg=None, dn=None):
    """fp64 restatement of the fused pooling (SURVEY.md Appendix B on the folded uint8 input, section 0.7)."""
    x = np_ref.dequant_l2norm_folded(q, nf)                                  # [B,F,D], rows >= nf are 0
    B, F, _ = x.shape
    mask = (np.arange(F)[None, :] < nf[:, None]).astype(np.float64)
    # the assignment sees the NORMALISED frame for every row (padding rows are masked afterwards)
    xa = np_ref.dequant_l2norm_folded(q, None)
    a = np_ref.softmax(xa @ Wc + bc, axis=2) * mask[:, :, None]
    agg = np.einsum("bfk,bfd->bkd", a, x)
    if dagg is None:
        return a, agg
    da = np.einsum("bfd,bkd->bfk", xa, dagg) + dn[:, None, :]
    ds = a * (da - (a * da).sum(2, keepdims=True))
    dWc = np.einsum("bfd,bfk->dk", xa, ds)
    return a, agg, dWc, ds.sum((0, 1))


@pytest.mark.parametrize("shape", [(5, 50, 128), (3, 300, 1152), (2, 37, 1024), (260, 9, 64)])
@pytest.mark.parametrize("nsplit", [2, 1])
def test_netvlad_fused_u8(dev, shape, nsplit):
    """Fused uint8 NetVLAD pooling (rows + cols kernels, csrc/netvlad_fused.hip) against the fp64 oracle: assignment,
    aggregation and the backward into W_c / b_c; ragged num_frames incl. 0, 1 and F; frame counts that are not multiples
    of the 32-frame step; feature widths with a partial last 384-slice (1024) and a single block (64)."""
    B, F, Dm = shape
    K = 64
    rs = np.random.RandomState(B * 1000 + F)
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    q[0, 0] = 0
    q[0, F - 1] = 255
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0] = F
    nf[1] = 1
    if B > 2:
        nf[2] = 0
    Wc = (rs.randn(Dm, K) * 3.0 / np.sqrt(Dm)).astype(np.float32)
    bc = (rs.randn(K) * 0.5).astype(np.float32)
    dagg = (rs.randn(B, K, Dm) * 1e-3 * np.exp(rs.randn(B, 1, 1))).astype(np.float32)    # per-video gradient scales differ
    dn = (rs.randn(B, K) * 1e-3).astype(np.float32)
    aref, aggref, dWref, dbref = _netvlad_pool_ref(q, nf, Wc.astype(np.float64), bc.astype(np.float64),
                                                   dagg.astype(np.float64), dn.astype(np.float64))
    qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)
    cT, nsum, agg = seq_ops.netvlad_fwd_u8(qd, nfd, D(Wc, dev), D(bc, dev), nsplit=nsplit)
    tol = 2e-6 if nsplit == 2 else 3e-3
    # cT[b,k,f] = a[b,f,k] / ||dequantised frame||, frames contiguous and zero padded to a multiple of 32
    Fp = (F + 31) // 32 * 32
    assert tuple(cT.shape) == (B, K, Fp)
    norm = np.sqrt(((q.astype(np.float64) * (4.0 / 255.0) + (4.0 / 512.0 - 2.0)) ** 2).sum(-1))       # [B,F]
    a = H(cT)[:, :, :F].transpose(0, 2, 1) * norm[:, :, None]
    assert np.abs(a - aref).max() < tol
    assert (H(cT)[:, :, F:] == 0).all()
    assert np.abs(H(nsum) - aref.sum(1)).max() < tol * F
    assert np.abs(H(agg) - aggref).max() < tol * max(1.0, np.abs(aggref).max())
    assert (a[1, 1:] == 0).all()                                              # masked frames are exactly 0
    if B > 2:
        assert (a[2] == 0).all() and (H(agg)[2] == 0).all()
    a = cT
    # backward (uses the kernel's own a, as the training step does)
    dW = torch.full((Dm, K), 7.0, device=dev)
    db = torch.full((K,), 7.0, device=dev)
    seq_ops.netvlad_bwd_u8(qd, nfd, a, D(dagg, dev), D(dn, dev), dW, 0.0, db, 0.0, nsplit=nsplit)
    gt = 2e-5 if nsplit == 2 else 5e-3
    assert np.abs(H(dW) - dWref).max() < gt * np.abs(dWref).max()
    assert np.abs(H(db) - dbref).max() < gt * max(np.abs(dbref).max(), 1e-6)
    seq_ops.netvlad_bwd_u8(qd, nfd, a, D(dagg, dev), D(dn, dev), dW, 1.0, db, 1.0, nsplit=nsplit)   # beta = 1 accumulates
    assert np.abs(H(dW) - 2 * dWref).max() < 2 * gt * np.abs(dWref).max()
    # vanishing upstream gradients (saturated model: fp32 denormal range) stay finite -- no scale overflow
    seq_ops.netvlad_bwd_u8(qd, nfd, a, D(dagg * 1e-35, dev), D(dn * 1e-35, dev), dW, 0.0, db, 0.0, nsplit=nsplit)
    assert torch.isfinite(dW).all() and torch.isfinite(db).all() and float(dW.abs().max()) < 1e-30
    # deterministic: same bits on a second run
    c2, n2, agg2 = seq_ops.netvlad_fwd_u8(qd, nfd, D(Wc, dev), D(bc, dev), nsplit=nsplit)
    assert torch.equal(cT, c2) and torch.equal(agg, agg2) and torch.equal(nsum, n2)


def test_netvlad_fused_rejects_unsupported(dev):
    q = torch.zeros((2, 4, 100), dtype=torch.uint8, device=dev)               # D % 64 != 0
    assert not seq_ops.netvlad_fused_supported(q, 64)
    assert not seq_ops.netvlad_fused_supported(torch.zeros((2, 4, 128), dtype=torch.uint8, device=dev), 32)
    with pytest.raises(ValueError):                                            # YT8M_E_SHAPE, no silent fallback in the op
        seq_ops.netvlad_fwd_u8(q, None, torch.zeros((100, 64), device=dev), torch.zeros(64, device=dev))


def test_whole_head_entry_points(dev):
    """yt8m_moe_fwd / yt8m_moe_bwd / yt8m_logistic_fwd_bwd: the complete MoeModel / LogisticModel head +
    CrossEntropyLoss in two (one) C calls, against fp64 autograd of the oracle; beta accumulation; inference form."""
    import ctypes
    from oracle import torch_ref
    lib = L.lib()
    rs = np.random.RandomState(44)
    B, Dm, V, M = 37, 70, 131, 2
    x = rs.randn(B, Dm).astype(np.float32)
    Wg = (rs.randn(Dm, V * (M + 1)) * 0.3).astype(np.float32)
    We = (rs.randn(Dm, V * M) * 0.3).astype(np.float32)
    be = (rs.randn(V * M) * 0.3).astype(np.float32)
    y = (rs.rand(B, V) < 0.05).astype(np.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    xd, Wgd, Wed, bed, yd = D(x, dev), D(Wg, dev), D(We, dev), D(be, dev), torch.from_numpy(y).to(dev)
    Zg = torch.empty((B, V * (M + 1)), device=dev)
    Ze = torch.empty((B, V * M), device=dev)
    p = torch.empty((B, V), device=dev)
    loss = torch.zeros((), device=dev)
    ws = torch.empty(lib.yt8m_moe_workspace_bytes(B, V), dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_moe_fwd(P(xd), P(Wgd), P(Wed), P(bed), P(yd), 0, B, Dm, V, M, 1e-5, P(Zg), P(Ze), P(p), P(loss), P(ws),
                             ws.numel(), st))
    t = {k: torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for k, v in dict(x=x, Wg=Wg, We=We, be=be).items()}
    pr = torch_ref.moe(t["x"], t["Wg"], t["We"], t["be"], M)
    lr = torch_ref.cross_entropy(pr, torch.from_numpy(y.astype(np.float64)))
    (0.9 * lr).backward()
    assert np.abs(H(p) - pr.detach().numpy()).max() < 1e-5
    assert abs(float(loss) - lr.item()) < 1e-5 * abs(lr.item())
    dWg, dWe, dbe = torch.full_like(Wgd, 3.0), torch.full_like(Wed, 3.0), torch.full_like(bed, 3.0)
    dx = torch.empty_like(xd)
    L.check(lib.yt8m_moe_bwd(P(xd), P(Wgd), P(Wed), P(Zg), P(Ze), P(yd), 0, B, Dm, V, M, 1e-5, 0.9, P(dWg), P(dWe), P(dbe), 1.0,
                             P(dx), P(ws), ws.numel(), st))
    for got, ref, off in ((dWg, t["Wg"].grad, 3.0), (dWe, t["We"].grad, 3.0), (dbe, t["be"].grad, 3.0), (dx, t["x"].grad, 0.0)):
        r = ref.numpy()
        assert np.abs(H(got) - off - r).max() <= 2e-4 * max(1.0, np.abs(r).max())
    # inference form: no labels, no loss
    p2 = torch.empty_like(p)
    L.check(lib.yt8m_moe_fwd(P(xd), P(Wgd), P(Wed), P(bed), None, 0, B, Dm, V, M, 1e-5, P(Zg), P(Ze), P(p2), None, P(ws),
                             ws.numel(), st))
    assert torch.equal(p, p2)
    assert lib.yt8m_moe_fwd(P(xd), P(Wgd), P(Wed), P(bed), P(yd), 0, B, Dm, V, M, 1e-5, P(Zg), P(Ze), P(p2), None, P(ws),
                            ws.numel(), st) == -1                                      # labels without loss_out
    assert lib.yt8m_moe_fwd(P(xd), P(Wgd), P(Wed), P(bed), None, 0, B, Dm, V, M, 1e-5, P(Zg), P(Ze), P(p2), None, P(ws), 16,
                            st) == -1                                                  # workspace too small
    # LogisticModel
    W = (rs.randn(Dm, V) * 0.3).astype(np.float32)
    b = (rs.randn(V) * 0.3).astype(np.float32)
    Wd, bd = D(W, dev), D(b, dev)
    Z = torch.empty((B, V), device=dev)
    dW, db = torch.empty_like(Wd), torch.empty_like(bd)
    L.check(lib.yt8m_logistic_fwd_bwd(P(xd), P(Wd), P(bd), P(yd), 0, B, Dm, V, 1e-5, P(p), P(loss), P(Z), P(dW), P(db), 0.0,
                                      P(dx), P(ws), ws.numel(), st))
    tl = {k: torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for k, v in dict(x=x, W=W, b=b).items()}
    pl = torch_ref.logistic(tl["x"], tl["W"], tl["b"])
    ll = torch_ref.cross_entropy(pl, torch.from_numpy(y.astype(np.float64)))
    ll.backward()
    assert np.abs(H(p) - pl.detach().numpy()).max() < 1e-5 and abs(float(loss) - ll.item()) < 1e-5 * abs(ll.item())
    for got, ref in ((dW, tl["W"].grad), (db, tl["b"].grad), (dx, tl["x"].grad)):
        r = ref.numpy()
        assert np.abs(H(got) - r).max() <= 2e-4 * max(1.0, np.abs(r).max())


def test_sqnorm_and_adam_multi(dev):
    """Per-tensor clip on (g*gscale + l2 w) + TF-Adam over the arena vs the oracle, incl. ragged tensor sizes that
    exercise chunk tails, and bitwise reproducibility of the reduction."""
    from yt8m_amd.variables import reset_default_graph, random_normal, zeros
    rs = np.random.RandomState(9)
    g = reset_default_graph(device=dev, seed=3)
    g.begin_step()
    shapes = {"a/weights": (37, 131), "a/biases": (131,), "b/weights": (4097, 3), "c/weights": (1, 1)}
    vs = {k: g.get_variable(k, s, random_normal(0.5), l2=1e-2 if k.endswith("weights") else 0.0) for k, s in shapes.items()}
    g.finalize()
    params = {k: H(v.data) for k, v in vs.items()}
    state = {}
    for step in range(3):
        grads = {k: rs.randn(*s) * (5.0 if step == 0 else 0.01) for k, s in shapes.items()}
        for k, v in vs.items():
            v.grad.copy_(D(grads[k] * 2.0, dev))                 # stored as the SUM over 2 ranks; gscale = 1/2
        lr = np_ref.exponential_decay(0.01, step, 8)
        t = step + 1
        ops.sqnorm_and_adam(g, lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t), gscale=0.5, clip=1.0)
        n1 = g.norms.clone()
        ops._lib.check(ops._lib.lib().yt8m_sqnorm_multi(ops._p(g.params), ops._p(g.grads), ops._p(g.chunks), g.nchunks,
                                                          ops._p(g.l2), 0.5, ops._p(g.partial), ops._p(g.norms), 0, 4, None, 0, ops._stream()))
        params, state = np_ref.train_step_update(params, grads, state, step, 0.01, 8,
                                                 {k for k in shapes if k.endswith("weights")}, l2_penalty=1e-2, clip=1.0)
        for k, v in vs.items():
            assert np.abs(H(v.data) - params[k]).max() < 2e-6, (step, k)
    a = g.norms.clone()
    ops._lib.check(ops._lib.lib().yt8m_sqnorm_multi(ops._p(g.params), ops._p(g.grads), ops._p(g.chunks), g.nchunks,
                                                      ops._p(g.l2), 0.5, ops._p(g.partial), ops._p(g.norms), 0, 4, None, 0, ops._stream()))
    assert torch.equal(a, g.norms)                               # fixed-order reduction: bitwise reproducible


def _bf16_round(a):
    """numpy restatement of round-to-nearest-even fp32 -> bf16 (returned as float32 values)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def test_linear_bf16_mode(dev, monkeypatch):
    """ops.linear with bf16=True: forward equals the fp64 product of the bf16-ROUNDED operands; dW / dx / db against fp64
    autograd of the unrounded op at bf16 resolution; small problems stay on the exact fp32 path."""
    from yt8m_amd.variables import reset_default_graph, random_normal, zeros
    monkeypatch.setattr(ops, "BF16_MIN_ROWS", 2)                             # production: 512 rows (cast amortisation)
    rs = np.random.RandomState(33)
    M, K, N = 256, 512, 1024                                                # M*N*K = 2^27: bf16 for all three GEMMs
    g = reset_default_graph(device=dev, seed=1)
    g.begin_step()
    W = g.get_variable("fc/weights", (K, N), random_normal(0.05))
    b = g.get_variable("fc/biases", (N,), zeros)
    g.finalize()
    x = rs.randn(M, K).astype(np.float32)
    dy = rs.randn(M, N).astype(np.float32)
    xd = D(x, dev).requires_grad_(True)
    y = ops.linear(xd, W, b, bf16=True)
    Wh = H(W.data)
    ref = _bf16_round(x).astype(np.float64) @ _bf16_round(Wh.astype(np.float32)).astype(np.float64)
    assert np.abs(H(y) - ref).max() < 1e-4
    assert 1e-4 < np.abs(H(y) - x.astype(np.float64) @ Wh).max() < 5e-2       # it really is the bf16 path
    y.backward(D(dy, dev))
    dW, dx, db = x.astype(np.float64).T @ dy, dy.astype(np.float64) @ Wh.T, dy.astype(np.float64).sum(0)
    assert np.abs(H(W.grad) - dW).max() < 1e-2 * np.abs(dW).max()
    assert np.abs(H(xd.grad) - dx).max() < 1e-2 * np.abs(dx).max()
    assert np.abs(H(b.grad) - db).max() < 1e-4 * np.abs(db).max()
    # small problem: exact path even when bf16 is requested
    g.begin_step()
    xs = D(x[:8], dev)
    ys = ops.linear(xs, W, b, bf16=True)
    assert np.abs(H(ys) - x[:8].astype(np.float64) @ Wh).max() < 1e-5


def test_cast_bf16_and_bf16_gemm(dev):
    """bf16 path: the cast is bit-exact against the RNE restatement (also transposed, ragged shapes); the bf16 NT GEMM
    equals the fp64 product of the ROUNDED operands to fp32-accumulation noise, across tile / split-K / K-tail cases."""
    rs = np.random.RandomState(31)
    # (5,7) / (130,257): scalar kernels; (64,64), (1024,1152), (132,260), (8,4716): vector / tiled kernels incl. ragged tiles
    for R, C in [(5, 7), (64, 64), (130, 257), (1024, 1152), (132, 260), (8, 4716), (6, 8)]:
        x = (rs.randn(R, C) * 3).astype(np.float32)
        x.flat[0] = 1.00390625         # exact tie between two bf16 values -> even
        x.flat[-1] = np.nan
        want = _bf16_round(x)
        xb = ops.cast_bf16(D(x, dev))
        assert np.array_equal(xb.float().cpu().numpy(), want, equal_nan=True)
        xt = ops.cast_bf16(D(x, dev), transpose=True)
        assert xt.shape == (C, R) and np.array_equal(xt.float().cpu().numpy(), want.T, equal_nan=True)
        pb, pt = ops.cast_bf16_both(D(x, dev))                       # both layouts from one pass
        assert torch.equal(pb.view(torch.int16), xb.view(torch.int16)) and torch.equal(pt.view(torch.int16), xt.view(torch.int16))
        wide = D(np.concatenate([x, x], axis=1), dev)[:, :C]          # a row-strided view (ld = 2C)
        assert torch.equal(ops.cast_bf16(wide).view(torch.int16), xb.view(torch.int16))
        assert torch.equal(ops.cast_bf16_both(wide)[1].view(torch.int16), xt.view(torch.int16))
    for M, N, K in [(1, 1, 2), (5, 7, 6), (128, 128, 32), (130, 257, 66), (64, 300, 1152), (1024, 2358, 1152), (100, 1000, 4096)]:
        A = rs.randn(M, K).astype(np.float32)
        B = rs.randn(N, K).astype(np.float32)
        bias = rs.randn(N).astype(np.float32)
        ref = _bf16_round(A).astype(np.float64) @ _bf16_round(B).astype(np.float64).T
        Ab, Bb = ops.cast_bf16(D(A, dev)), ops.cast_bf16(D(B, dev))
        C1, = ops.gemm_bf16_nt_grouped([dict(A=Ab, B=Bb)])
        scale = np.abs(A).max() * np.abs(B).max() * K
        assert np.abs(H(C1) - ref).max() <= 2e-6 * scale, (M, N, K)
        C2, = ops.gemm_bf16_nt_grouped([dict(A=Ab, B=Bb, bias=D(bias, dev))])
        ops.gemm_bf16_nt_grouped([dict(A=Ab, B=Bb, out=C2, beta=1.0)])
        assert np.abs(H(C2) - (2 * ref + bias)).max() <= 4e-6 * scale
    with pytest.raises(TypeError):
        ops.gemm_bf16_nt_grouped([dict(A=torch.zeros(4, 4, device=dev), B=torch.zeros(4, 4, device=dev))])
    with pytest.raises(ValueError):
        ops.gemm_bf16_nt_grouped([dict(A=torch.zeros(4, 3, device=dev, dtype=torch.bfloat16), B=torch.zeros(4, 3, device=dev, dtype=torch.bfloat16))])


@pytest.mark.parametrize("n,offset", [(1, 0), (7, 0), (1024, 0), (1027, 5), (4099, 4096), (3, 2), (100001, 123457)])
def test_dropout_bit_exact_with_philox_oracle(dev, n, offset):
    """yt8m_dropout_f32 == oracle.philox.dropout bit for bit (mask AND the x / keep_prob quotient), for aligned and unaligned
    offsets / sizes; in place; and the autograd backward replays the same mask."""
    from oracle import philox
    rs = np.random.RandomState(n)
    x = rs.randn(n).astype(np.float32)
    seed = 0x9E3779B97F4A7C15 ^ n
    for keep in (0.5, 0.7, 1.0):
        ref = philox.dropout(x, keep, seed, offset)
        xd = D(x, dev).requires_grad_(True)
        y = ops._Dropout.apply(xd, keep, seed, offset)
        assert np.array_equal(y.detach().cpu().numpy(), ref), (n, offset, keep)
        dy = rs.randn(n).astype(np.float32)
        y.backward(D(dy, dev))
        assert np.array_equal(xd.grad.cpu().numpy(), philox.dropout(dy, keep, seed, offset))
        z = D(x, dev)
        ops.dropout_(z, keep, seed, offset)
        assert np.array_equal(z.cpu().numpy(), ref)
    if n > 8:                                      # a chunk draws what the whole tensor draws at that position
        h = n // 3
        whole = ops._Dropout.apply(D(x, dev), 0.7, seed, offset).cpu().numpy()
        part = ops._Dropout.apply(D(x[h:], dev), 0.7, seed, offset + h).cpu().numpy()
        assert np.array_equal(whole[h:], part)


def test_dropout_and_noise_statistics_and_errors(dev):
    from oracle import philox
    n = 1 << 20
    x = torch.ones(n, device=dev)
    y = ops.dropout(x, 0.8, seed=42).cpu().numpy()
    assert abs((y != 0).mean() - 0.8) < 2e-3 and np.allclose(y[y != 0], np.float32(1.0) / np.float32(0.8))
    z = ops.add_noise(torch.zeros(n + 3, device=dev), 0.5, seed=43, offset=1).cpu().numpy().astype(np.float64)
    ref = philox.add_noise(np.zeros(n + 3), 0.5, 43, offset=1)
    assert np.abs(z - ref).max() < 5e-6                       # device logf / cosf vs float64 Box-Muller
    assert abs(z.mean()) < 2e-3 and abs(z.std() - 0.5) < 2e-3
    zz = z / 0.5
    assert abs((zz ** 3).mean()) < 2e-2 and abs((zz ** 4).mean() - 3.0) < 5e-2
    with pytest.raises(ValueError):
        ops.dropout(x, 0.0, seed=1)
    with pytest.raises(ValueError):
        ops.dropout(x, 1.5, seed=1)
    assert ops.dropout(torch.empty(0, device=dev), 0.5, seed=1).numel() == 0


def _mkvars(dev, arrays):
    from yt8m_amd.variables import reset_default_graph, zeros
    g = reset_default_graph(device=dev)
    g.begin_step()
    vs = [g.get_variable("v%d" % i, a.shape, zeros) for i, a in enumerate(arrays)]
    g.finalize()
    for v, a in zip(vs, arrays):
        v.data.copy_(D(a, dev))
    return vs


@pytest.mark.parametrize("B,F,Din,Hh", [(5, 7, 6, 4), (37, 5, 20, 256), (70, 3, 16, 512)])
def test_gru_layer_fwd_bwd(dev, B, F, Din, Hh):
    """GRUCell layer vs the oracle (fp64 autograd): H = 4 takes the per-step grouped-GEMM form, H = 256 / 512 the packed MFMA
    step kernels with fused epilogues (lstm_fused.hip EP_GRU_*), B not a multiple of the 32 / 16 row tiles, ragged num_frames."""
    from oracle import torch_ref
    rs = np.random.RandomState(31)
    x = rs.randn(B, F, Din).astype(np.float32)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[:5] = [F, 1, min(4, F), 0, F]
    sc = 0.4 if Hh < 64 else 0.06
    arrs = [(rs.randn(Din + Hh, 2 * Hh) * sc).astype(np.float32), (rs.randn(2 * Hh) * 0.1 + 1).astype(np.float32),
            (rs.randn(Din + Hh, Hh) * sc).astype(np.float32), (rs.randn(Hh) * 0.1).astype(np.float32)]
    Wg, bg, Wc, bc = _mkvars(dev, arrs)
    xt = D(x, dev).transpose(0, 1).contiguous().requires_grad_(True)
    out, h = seq_ops.gru_layer(xt, Wg, bg, Wc, bc, torch.from_numpy(nf).to(dev))
    go, gh = rs.randn(F, B, Hh).astype(np.float32), rs.randn(B, Hh).astype(np.float32)
    ((out * D(go, dev)).sum() + (h * D(gh, dev)).sum()).backward()
    tx = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    tp = [torch.from_numpy(a.astype(np.float64)).requires_grad_(True) for a in arrs]
    to, th = torch_ref.gru_stack(tx, torch.from_numpy(nf), [tuple(tp)])
    ((to * torch.from_numpy(go.astype(np.float64)).transpose(0, 1)).sum() + (th[0] * torch.from_numpy(gh.astype(np.float64))).sum()).backward()
    assert np.abs(H(out).transpose(1, 0, 2) - to.detach().numpy()).max() < 2e-5
    assert np.abs(H(h) - th[0].detach().numpy()).max() < 2e-5
    for v, t in zip((Wg, bg, Wc, bc), tp):
        assert np.abs(H(v.grad) - t.grad.numpy()).max() < 2e-4 * max(1.0, np.abs(t.grad.numpy()).max())
    assert np.abs(H(xt.grad).transpose(1, 0, 2) - tx.grad.numpy()).max() < 2e-4


@pytest.mark.parametrize("B,F,Din,Hh,keep", [(5, 7, 6, 12, 1.0), (37, 5, 20, 256, 1.0), (33, 4, 16, 512, 0.75), (3, 3, 8, 300, 0.5)])
def test_lnlstm_layer_fwd_bwd(dev, B, F, Din, Hh, keep):
    """LayerNormBasicLSTMCell layer vs the oracle: generic product (H = 12, 300) and the packed add-only step product
    (H = 256, 512); recurrent dropout masks from the Philox oracle; gamma / beta gradients of all five normalisations."""
    from oracle import torch_ref
    rs = np.random.RandomState(32)
    x = rs.randn(B, F, Din).astype(np.float32)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[:3] = [F, 1, 0]
    arrs = [(rs.randn(Din + Hh, 4 * Hh) * (0.4 if Hh < 64 else 0.08)).astype(np.float32)]
    arrs += [(rs.rand(Hh) + 0.5).astype(np.float32) for _ in range(5)] + [(rs.randn(Hh) * 0.2).astype(np.float32) for _ in range(5)]
    vs = _mkvars(dev, arrs)
    xt = D(x, dev).transpose(0, 1).contiguous().requires_grad_(True)
    seed = 0xC0FFEE1234
    out, c, h = seq_ops.lnlstm_layer(xt, vs[0], vs[1:6], vs[6:11], torch.from_numpy(nf).to(dev), forget_bias=1.0, keep_prob=keep, seed=seed)
    go, gc, gh = rs.randn(F, B, Hh).astype(np.float32), rs.randn(B, Hh).astype(np.float32), rs.randn(B, Hh).astype(np.float32)
    ((out * D(go, dev)).sum() + (c * D(gc, dev)).sum() + (h * D(gh, dev)).sum()).backward()
    tx = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    tp = [torch.from_numpy(a.astype(np.float64)).requires_grad_(True) for a in arrs]
    to, tc, th = torch_ref.lnlstm_stack(tx, torch.from_numpy(nf), [(tp[0], tp[1:6], tp[6:11])],
                                        dropout_spec=None if keep >= 1 else (keep, [seed]))
    ((to * torch.from_numpy(go.astype(np.float64)).transpose(0, 1)).sum() + (tc[0] * torch.from_numpy(gc.astype(np.float64))).sum()
     + (th[0] * torch.from_numpy(gh.astype(np.float64))).sum()).backward()
    assert np.abs(H(out).transpose(1, 0, 2) - to.detach().numpy()).max() < 5e-5
    assert np.abs(H(c) - tc[0].detach().numpy()).max() < 5e-5 and np.abs(H(h) - th[0].detach().numpy()).max() < 5e-5
    for k, (v, t) in enumerate(zip(vs, tp)):
        ref = t.grad.numpy()
        assert np.abs(H(v.grad) - ref).max() < 5e-4 * max(1.0, np.abs(ref).max()), k
    assert np.abs(H(xt.grad).transpose(1, 0, 2) - tx.grad.numpy()).max() < 5e-4 * max(1.0, np.abs(tx.grad.numpy()).max())


@pytest.mark.parametrize("M,K,N", [(1, 4, 1), (37, 256, 8), (5000, 2304, 8), (4099, 1000, 9), (2500, 2048, 16), (70, 516, 3)])
def test_skinny_gemm_kernels(dev, M, K, N):
    """csrc/gemm_skinny.hip (<= 16 outputs over many rows) vs fp64 numpy: forward with bias / accumulate, dW through the
    chunked deterministic reduction, dx with accumulate; W given as a row slice of a wider matrix (ld = N)."""
    rs = np.random.RandomState(M + K + N)
    x = rs.randn(M, K).astype(np.float32)
    Wfull = rs.randn(K + 8, N).astype(np.float32)
    b = rs.randn(N).astype(np.float32)
    dy = rs.randn(M, N).astype(np.float32)
    y0 = rs.randn(M, N).astype(np.float32)
    xd, Wd, bd, dyd = D(x, dev), D(Wfull, dev)[4:4 + K], D(b, dev), D(dy, dev)
    W = Wfull[4:4 + K].astype(np.float64)
    assert L.lib().yt8m_skinny_supported(M, K, N) == 1
    y = ops.skinny_fwd(xd, Wd, bd, torch.empty(M, N, device=dev))
    ref = x.astype(np.float64) @ W + b
    assert np.abs(H(y) - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    y2 = ops.skinny_fwd(xd, Wd, None, D(y0, dev), beta=1.0)
    assert np.abs(H(y2) - (x.astype(np.float64) @ W + y0)).max() < 2e-5 * max(1.0, np.abs(ref).max())
    refdw = x.astype(np.float64).T @ dy
    dW = ops.skinny_dw(xd, dyd, torch.empty(K, N, device=dev))
    assert np.abs(H(dW) - refdw).max() < 2e-5 * max(1.0, np.abs(refdw).max())
    dW1 = ops.skinny_dw(xd, dyd, dW.clone(), beta=1.0)
    assert np.abs(H(dW1) - 2 * refdw).max() < 4e-5 * max(1.0, np.abs(refdw).max())
    assert torch.equal(ops.skinny_dw(xd, dyd, torch.empty(K, N, device=dev)), dW)          # deterministic
    refdx = dy.astype(np.float64) @ W.T
    dx = ops.skinny_dx(dyd, Wd)
    assert np.abs(H(dx) - refdx).max() < 2e-5 * max(1.0, np.abs(refdx).max())
    dx1 = ops.skinny_dx(dyd, Wd, dx=dx.clone(), beta=1.0)
    assert np.abs(H(dx1) - 2 * refdx).max() < 4e-5 * max(1.0, np.abs(refdx).max())


def test_skinny_rejects_and_linear_cat(dev, monkeypatch):
    """N > 16 / K % 4 != 0 are shape errors; ops.linear_cat (FC over a logical concatenation, with a per-group tiled part) ==
    torch on the materialised concatenation, through both the streaming kernels and the MFMA path."""
    from yt8m_amd.variables import reset_default_graph, zeros
    x = torch.zeros(64, 8, device=dev)
    with pytest.raises(ValueError):
        ops.skinny_fwd(x, torch.zeros(8, 17, device=dev), None, torch.zeros(64, 17, device=dev))
    with pytest.raises(ValueError):
        ops.skinny_fwd(torch.zeros(64, 6, device=dev), torch.zeros(6, 4, device=dev), None, torch.zeros(64, 4, device=dev))
    assert L.lib().yt8m_skinny_supported(100, 8192, 8) == 0 and L.lib().yt8m_skinny_supported(100, 4096, 8) == 1
    rs = np.random.RandomState(5)
    Bv, F, K1, K2, K3, N = 6, 50, 24, 12, 16, 8
    a = rs.randn(Bv, F, K1).astype(np.float32)
    c = rs.randn(Bv, F, K2).astype(np.float32)
    m = rs.randn(Bv, K3).astype(np.float32)
    Wn = (rs.randn(K1 + K2 + K3, N) * 0.3).astype(np.float32)
    bn = rs.randn(N).astype(np.float32)
    gy = rs.randn(Bv, F, N).astype(np.float32)
    ta, tc, tm, tW, tb = [torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for v in (a, c, m, Wn, bn)]
    ty = torch.cat([ta, tc, tm[:, None, :].expand(Bv, F, K3)], 2) @ tW + tb
    (ty * torch.from_numpy(gy.astype(np.float64))).sum().backward()
    for min_rows in (1, 1 << 30):                  # streaming kernels / padded MFMA tiles
        monkeypatch.setattr(ops, "SKINNY_MIN_ROWS", min_rows)
        g = reset_default_graph(device=dev)
        g.begin_step()
        W = g.get_variable("w", Wn.shape, zeros)
        b = g.get_variable("b", bn.shape, zeros)
        g.finalize()
        W.data.copy_(D(Wn, dev)); b.data.copy_(D(bn, dev))
        da, dc, dm = [D(v, dev).requires_grad_(True) for v in (a, c, m)]
        y = ops.linear_cat([da, dc], W, b, group_parts=(dm,))
        assert y.shape == (Bv, F, N) and np.abs(H(y) - ty.detach().numpy()).max() < 2e-5
        (y * D(gy, dev)).sum().backward()
        for got, ref in ((W.grad, tW.grad), (b.grad, tb.grad), (da.grad, ta.grad), (dc.grad, tc.grad), (dm.grad, tm.grad)):
            assert np.abs(H(got) - ref.numpy()).max() < 5e-5 * max(1.0, np.abs(ref.numpy()).max())


@pytest.mark.parametrize("B,F,A,Hh", [(3, 7, 3, 8), (5, 300, 8, 1152), (2, 129, 16, 260), (4, 64, 1, 1024)])
def test_attention_pooling_streaming_kernels(dev, B, F, A, Hh, monkeypatch):
    """yt8m_attn_pool_fwd / _bwd (einsum "ijk,ijl->ikl" and its two gradients) vs fp64 numpy, and seq_ops.pool_tn through both
    the streaming kernels and the batched GEMM."""
    rs = np.random.RandomState(B * F + A)
    w = rs.randn(B, F, A).astype(np.float32)
    x = rs.randn(B, F, Hh).astype(np.float32)
    g = rs.randn(B, A, Hh).astype(np.float32)
    refC = np.einsum("bfa,bfh->bah", w.astype(np.float64), x.astype(np.float64))
    refdw = np.einsum("bfh,bah->bfa", x.astype(np.float64), g.astype(np.float64))
    refdx = np.einsum("bfa,bah->bfh", w.astype(np.float64), g.astype(np.float64))
    for min_elems in (1, 1 << 40):
        monkeypatch.setattr(seq_ops, "POOL_STREAM_MIN", min_elems)
        wd, xd = D(w, dev).requires_grad_(True), D(x, dev).requires_grad_(True)
        C = seq_ops.pool_tn(wd, xd)
        assert np.abs(H(C) - refC).max() < 2e-5 * max(1.0, np.abs(refC).max())
        (C * D(g, dev)).sum().backward()
        assert np.abs(H(wd.grad) - refdw).max() < 2e-5 * max(1.0, np.abs(refdw).max())
        assert np.abs(H(xd.grad) - refdx).max() < 2e-5 * max(1.0, np.abs(refdx).max())
    assert L.lib().yt8m_attn_pool_supported(4, 10, 17, 64) == 0 and L.lib().yt8m_attn_pool_supported(4, 10, 8, 66) == 0


def test_bf16_large_tile_gemm(dev, monkeypatch):
    """gemm_bf16.hip (256 x 256 tiles, 4-stage LDS-DMA ring) forced on for small problems: ragged edges in M and N, K tails
    (K % 32 != 0), padded row pitches, bias, accumulate, two problems in one launch -- against the fp64 product of the rounded
    operands, and bit-for-bit what the small-tile kernel returns is NOT required (different summation order)."""
    monkeypatch.setenv("YT8M_BF16_BIG_MIN", "1")
    rs = np.random.RandomState(41)
    for shapes in [[(300, 520, 72)], [(1000, 777, 1000), (300, 5000, 72)], [(257, 255, 2304)], [(512, 512, 14148)], [(64, 64, 32)]]:
        items, refs = [], []
        for (M, N, K) in shapes:
            A = rs.randn(M, K).astype(np.float32)
            B = rs.randn(N, K).astype(np.float32)
            bias = rs.randn(N).astype(np.float32)
            refs.append((_bf16_round(A).astype(np.float64) @ _bf16_round(B).astype(np.float64).T, bias,
                         np.abs(A).max() * np.abs(B).max() * K))
            items.append(dict(A=ops.cast_bf16(D(A, dev)), B=ops.cast_bf16(D(B, dev)), bias=D(bias, dev)))
        outs = ops.gemm_bf16_nt_grouped(items)
        for o, (ref, bias, scale) in zip(outs, refs):
            assert np.abs(H(o) - (ref + bias)).max() <= 2e-6 * scale
        items2 = [dict(A=it["A"], B=it["B"], out=o, beta=1.0) for it, o in zip(items, outs)]
        ops.gemm_bf16_nt_grouped(items2)
        for o, (ref, bias, scale) in zip(outs, refs):
            assert np.abs(H(o) - (2 * ref + bias)).max() <= 4e-6 * scale


@pytest.mark.parametrize("B,V", [(64, 64), (100, 70), (256, 4716), (37, 129)])
def test_moe_mix_bwd_bf16_fused(dev, B, V):
    """csrc/moe_bf16.hip: dL/dZ of the MoE mixing written straight as bf16 in both layouts + per-block bias partial sums ==
    the fp32 mixing backward followed by the casts / column sum (to one bf16 ulp: the two kernels may contract differently),
    for dL/dp given and for the CrossEntropyLoss-fused form; ragged row / label tiles and padded pitches."""
    rs = np.random.RandomState(B + V)
    M = 2
    Zg = (rs.randn(B, V * 3) * 2).astype(np.float32)
    Ze = (rs.randn(B, V * 2) * 2).astype(np.float32)
    dp = rs.randn(B, V).astype(np.float32)
    y = (rs.rand(B, V) < 0.1)
    lib = L.lib()

    def fused(dp_t, lab_t, dscale, up):
        gb, gt = ops._bf16_empty(B, 3 * V, dev), ops._bf16_empty(3 * V, B, dev)
        eb, et = ops._bf16_empty(B, 2 * V, dev), ops._bf16_empty(2 * V, B, dev)
        part = torch.empty((lib.yt8m_moe_mix_bwd_bf16_partial_rows(B), 2 * V), device=dev)
        zg_in, ze_in = D(Zg, dev), D(Ze, dev)                  # named: a temporary would be freed before the launch reads it
        L.check(lib.yt8m_moe_mix_bwd_bf16(ops._p(zg_in), ops._p(ze_in), ops._p(dp_t), ops._p(lab_t), 0, B, V, M, 1e-5,
                                          dscale, ops._p(up), ops._p(gb), gb.stride(0), ops._p(gt), gt.stride(0), ops._p(eb),
                                          eb.stride(0), ops._p(et), et.stride(0), ops._p(part), ops._stream()))
        return gb.float().cpu().numpy(), gt.float().cpu().numpy(), eb.float().cpu().numpy(), et.float().cpu().numpy(), H(part)

    def close(a, ref):     # two bf16 ulps of slack (the kernels contract / order their fp32 maths differently) + an absolute
        # term for the cancellation in s * (e - p) near zero
        return np.all(np.abs(a - ref) <= np.maximum(np.abs(ref), np.abs(a)) * 2.0 ** -6 + 2e-6 * np.abs(ref).max())

    # mode 0: dL/dp given
    zg, ze = D(Zg, dev), D(Ze, dev)
    ops.moe_mix_bwd_(zg, ze, D(dp, dev), V, M)
    rg, re = _bf16_round(H(zg).astype(np.float32)), _bf16_round(H(ze).astype(np.float32))
    gb, gt, eb, et, part = fused(D(dp, dev), None, 1.0, None)
    assert close(gb, rg)
    assert close(eb, re)
    assert np.array_equal(gt, gb.T) and np.array_equal(et, eb.T)
    assert np.abs(part.sum(0) - eb.astype(np.float64).sum(0)).max() < 1e-3 * max(1.0, np.abs(eb).sum(0).max())
    # mode 1: labels, fused cross-entropy gradient with a device-side upstream scale
    zg, ze = D(Zg, dev), D(Ze, dev)
    up = torch.tensor([0.37], device=dev)
    lab = torch.from_numpy(y.view(np.uint8)).to(dev)
    L.check(lib.yt8m_moe_mix_xent_bwd(ops._p(zg), ops._p(ze), ops._p(lab), 0, ops._p(up), B, V, M, 1e-5, 1.0, ops._stream()))
    rg, re = _bf16_round(H(zg).astype(np.float32)), _bf16_round(H(ze).astype(np.float32))
    gb, gt, eb, et, part = fused(None, lab, 1.0 / B, up)
    assert close(gb, rg)
    assert close(eb, re)
    assert np.array_equal(gt, gb.T) and np.array_equal(et, eb.T)
