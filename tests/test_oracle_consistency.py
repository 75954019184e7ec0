"""The two independent restatements (oracle/np_ref.py numpy fp64, oracle/torch_ref.py torch autograd) must agree,
and the hand-derived gradients of SURVEY.md Appendix G must match autograd.  No reference golden vectors exist for
these functions (parity unpinned, see oracle/__init__.py); this guards against a restatement slip."""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref

T = lambda a: torch.from_numpy(np.asarray(a))


def _moe_params(rs, D, V, M):
    return (rs.randn(D, V * (M + 1)) * 0.3, rs.randn(D, V * M) * 0.3, rs.randn(V * M) * 0.1)


@pytest.mark.parametrize("M", [1, 2, 4])
def test_moe_forward_and_layout(M):
    rs = np.random.RandomState(M)
    B, D, V = 5, 7, 11
    x = rs.randn(B, D)
    Wg, We, be = _moe_params(rs, D, V, M)
    p = np_ref.moe_model(x, Wg, We, be, M)
    pt = torch_ref.moe(T(x), T(Wg), T(We), T(be), M).numpy()
    assert np.abs(p - pt).max() < 1e-14
    assert np.abs(p - torch_ref.moe_fast(T(x), T(Wg), T(We), T(be), M).numpy()).max() < 1e-14
    # label-major / mixture-minor bookkeeping: column l*(M+1)+m is gate m of label l (moe_model.py:40-64)
    for b in (0, B - 1):
        for l in (0, 3, V - 1):
            g = x[b] @ Wg[:, l * (M + 1):(l + 1) * (M + 1)]
            e = x[b] @ We[:, l * M:(l + 1) * M] + be[l * M:(l + 1) * M]
            g = np.exp(g - g.max()); g /= g.sum()
            ref = float(np.sum(g[:M] / (1 + np.exp(-e))))
            assert abs(p[b, l] - ref) < 1e-14
    assert (p > 0).all() and (p < 1).all()


def test_moe_backward_matches_autograd():
    rs = np.random.RandomState(0)
    B, D, V, M = 6, 9, 13, 2
    x = rs.randn(B, D)
    Wg, We, be = _moe_params(rs, D, V, M)
    dp = rs.randn(B, V)
    tx, tWg, tWe, tbe = [T(a).clone().requires_grad_(True) for a in (x, Wg, We, be)]
    torch_ref.moe(tx, tWg, tWe, tbe, M).backward(T(dp))
    g = np_ref.moe_model_bwd(x, Wg, We, be, M, dp)
    for k, t in (("dWg", tWg), ("dWe", tWe), ("dbe", tbe), ("dx", tx)):
        assert np.abs(g[k] - t.grad.numpy()).max() < 1e-12, k


def test_cross_entropy_and_grad():
    rs = np.random.RandomState(1)
    B, V = 7, 19
    p = rs.rand(B, V) * 0.98 + 0.01
    p[0, 0], p[0, 1] = 0.0, 1.0          # the eps guards exactly these (losses.py:115)
    y = rs.rand(B, V) < 0.2
    w = rs.rand(B)
    for weights in (None, w):
        l = np_ref.cross_entropy_loss(p, y, weights)
        tp = T(p).clone().requires_grad_(True)
        lt = torch_ref.cross_entropy(tp, T(y), None if weights is None else T(weights))
        assert abs(l - lt.item()) < 1e-10
        lt.backward()
        d = np_ref.cross_entropy_loss_bwd(p, y, weights)
        assert np.abs(d - tp.grad.numpy()).max() < 1e-6 * np.abs(d).max()
    # logits-form BCE is NOT what the reference computes
    assert np_ref.XENT_EPS == 1e-5


def test_label_smoothing():
    y = np.zeros((2, 10), dtype=bool)
    y[0, :2] = True
    s = np_ref.label_smoothing(y, 0.1)
    assert s[0, 0] == pytest.approx(0.9 + 0.2 * 0.1) and s[0, 5] == pytest.approx(0.02) and s[1].sum() == 0


def test_l2_normalize_and_bwd():
    rs = np.random.RandomState(2)
    x = rs.randn(4, 6)
    x[1] = 0.0
    y = np_ref.l2_normalize(x)
    assert np.allclose(np.linalg.norm(y[[0, 2, 3]], axis=1), 1.0) and (y[1] == 0).all()
    tx = T(x).clone().requires_grad_(True)
    dy = rs.randn(4, 6)
    torch_ref.l2_normalize(tx).backward(T(dy))
    assert np.abs(np_ref.l2_normalize_bwd(x, dy) - tx.grad.numpy()).max() < 1e-9


def test_dequant_fold_identity():
    """dequantise + L2-normalise expressed on raw uint8 rows with integer sums."""
    rs = np.random.RandomState(3)
    q = rs.randint(0, 256, size=(3, 7, 16)).astype(np.uint8)
    nf = np.array([7, 1, 4])
    ref = np.stack([np_ref.l2_normalize(np_ref.get_video_matrix(q[b, :nf[b]], 7)[0]) for b in range(3)])
    assert np.abs(np_ref.dequant_l2norm_folded(q, nf) - ref).max() < 1e-13
    m, n = np_ref.get_video_matrix(rs.randint(0, 256, size=(310, 4)).astype(np.uint8), 300)
    assert n == 300 and m.shape == (300, 4)
    assert torch_ref.dequantize(torch.tensor([0, 255], dtype=torch.uint8), torch.float64).tolist() == [-1.9921875, 2.0078125]


def test_labels_multihot_bookkeeping():
    a = np_ref.labels_to_multihot([5, 3, 5, 0], 8)
    assert a.tolist() == [True, False, False, True, False, True, False, False]
    assert np_ref.labels_to_multihot([], 4).sum() == 0


def _lstm_layers(rs, D, H, L):
    layers, d = [], D
    for _ in range(L):
        layers.append((rs.randn(d + H, 4 * H) * 0.3, rs.randn(4 * H) * 0.1))
        d = H
    return layers


def test_lstm_dynamic_rnn_semantics():
    rs = np.random.RandomState(4)
    B, F, D, H = 4, 7, 5, 3
    x = rs.randn(B, F, D)
    nf = np.array([7, 1, 4, 0])
    layers = _lstm_layers(rs, D, H, 2)
    out, finals = np_ref.dynamic_rnn_lstm(x, nf, layers)
    tout, tc, th = torch_ref.lstm_stack(T(x), T(nf), [(T(W), T(b)) for W, b in layers])
    assert np.abs(out - tout.numpy()).max() < 1e-13
    for l in range(2):
        assert np.abs(finals[l][0] - tc[l].numpy()).max() < 1e-13 and np.abs(finals[l][1] - th[l].numpy()).max() < 1e-13
    # copy-through: outputs are zero past num_frames; state equals the state at t = n-1; n = 0 keeps the zero state
    assert (out[1, 1:] == 0).all() and (out[2, 4:] == 0).all() and (out[3] == 0).all()
    assert (finals[1][1][3] == 0).all() and (finals[0][0][3] == 0).all()
    out1, fin1 = np_ref.dynamic_rnn_lstm(x[:, :4], np.minimum(nf, 4), layers)
    assert np.abs(fin1[1][1][2] - finals[1][1][2]).max() < 1e-15
    # head inputs: LstmModel = [c0||h0||c1||h1] (4H), LstmMemoryModel = [c0||c1] (2H)
    st = np_ref.lstm_model_state(x, nf, layers)
    assert st.shape == (B, 4 * H) and np.allclose(st[:, :H], finals[0][0]) and np.allclose(st[:, 3 * H:], finals[1][1])
    assert np.abs(st - torch_ref.lstm_model_state(T(x), T(nf), [(T(W), T(b)) for W, b in layers]).numpy()).max() < 1e-13
    assert np_ref.lstm_memory_model_state(x, nf, layers).shape == (B, 2 * H)
    # forget_bias is added before the sigmoid
    c, h = np_ref.basic_lstm_step(np.zeros((1, D)), np.ones((1, H)), np.zeros((1, H)), np.zeros((D + H, 4 * H)), np.zeros(4 * H))
    assert np.allclose(c, 1 / (1 + np.exp(-1.0)))


def test_gru_and_layer_norm_lstm_restatements_agree():
    """The numpy (loop) and torch (autograd) restatements of tf.contrib.rnn.GRUCell / LayerNormBasicLSTMCell under dynamic_rnn
    are written independently and must agree, incl. copy-through for ragged num_frames and the layer-norm epsilon placement."""
    rs = np.random.RandomState(14)
    B, F, D, H = 4, 6, 5, 3
    x = rs.randn(B, F, D)
    nf = np.array([6, 1, 4, 0])
    gl, d_in = [], D
    for _ in range(2):
        gl.append((rs.randn(d_in + H, 2 * H) * 0.5, rs.randn(2 * H) * 0.1 + 1, rs.randn(d_in + H, H) * 0.5, rs.randn(H) * 0.1))
        d_in = H
    out, hs = np_ref.dynamic_rnn_gru(x, nf, gl)
    tout, th = torch_ref.gru_stack(T(x), T(nf), [tuple(T(a) for a in lay) for lay in gl])
    assert np.abs(out - tout.numpy()).max() < 1e-13 and all(np.abs(h - t.numpy()).max() < 1e-13 for h, t in zip(hs, th))
    assert (out[1, 1:] == 0).all() and (out[3] == 0).all() and (hs[1][3] == 0).all()
    # u -> 1 keeps the state, u -> 0 with r = 1 makes it a plain tanh RNN step
    h0 = rs.randn(2, H)
    keep = np_ref.gru_step(rs.randn(2, D), h0, np.zeros((D + H, 2 * H)), np.full(2 * H, 50.0), rs.randn(D + H, H), np.zeros(H))
    assert np.abs(keep - h0).max() < 1e-12
    ll, d_in = [], D
    for _ in range(2):
        ll.append((rs.randn(d_in + H, 4 * H) * 0.5, [rs.rand(H) + 0.5 for _ in range(5)], [rs.randn(H) * 0.2 for _ in range(5)]))
        d_in = H
    out, fin = np_ref.dynamic_rnn_layer_norm_lstm(x, nf, ll)
    tout, tc, th = torch_ref.lnlstm_stack(T(x), T(nf), [(T(W), [T(g) for g in ga], [T(b) for b in be]) for W, ga, be in ll])
    assert np.abs(out - tout.numpy()).max() < 1e-12
    for l in range(2):
        assert np.abs(fin[l][0] - tc[l].numpy()).max() < 1e-12 and np.abs(fin[l][1] - th[l].numpy()).max() < 1e-12
    # the carried cell state is the NORMALISED one: unit gamma / zero beta => zero mean, unit variance over the units
    c1, _ = np_ref.layer_norm_lstm_step(rs.randn(3, D), rs.randn(3, 8), rs.randn(3, 8), rs.randn(D + 8, 32), [np.ones(8)] * 5, [np.zeros(8)] * 5)
    assert np.abs(c1.mean(1)).max() < 1e-12 and np.abs(c1.var(1) - 1).max() < 1e-9


def test_attention_pooling_and_model():
    rs = np.random.RandomState(5)
    B, F, D, H, A, V, M = 3, 6, 4, 3, 2, 5, 2
    x = rs.randn(B, F, D)
    nf = np.array([6, 2, 1])
    layers = _lstm_layers(rs, D, H, 2)
    Wa, ba = rs.randn(D + H, A), rs.randn(A)
    Wg, We, be = _moe_params(rs, H, V, M)
    out, _ = np_ref.dynamic_rnn_lstm(x, nf, layers)
    pooled, w = np_ref.attention_pool(x, out, nf, Wa, ba)
    assert np.allclose(w.sum(2), 1.0) and (w[1, :, 2:] == 0).all()
    tp = torch_ref.attention_pool(T(x), T(out), T(nf), T(Wa), T(ba)).numpy()
    assert np.abs(pooled - tp).max() < 1e-13
    p = np_ref.lstm_attention_max_pooling_model(x, nf, layers, Wa, ba, Wg, We, be, M)
    pt = torch_ref.lstm_attention_max_pooling(T(x), T(nf), [(T(W), T(b)) for W, b in layers], T(Wa), T(ba), T(Wg), T(We), T(be), M)
    assert np.abs(p - pt.numpy()).max() < 1e-13


def test_chain_model():
    rs = np.random.RandomState(6)
    B, D, V, M, L, C = 4, 6, 7, 2, 2, 3
    x = rs.randn(B, D)
    P, d = {}, D
    for i in range(L):
        s = "prediction-%d" % i
        P["gates-%s/weights" % s], P["experts-%s/weights" % s], P["experts-%s/biases" % s] = _moe_params(rs, d, V, M)
        P["relu-%d/weights" % i], P["relu-%d/biases" % i] = rs.randn(V, C), rs.randn(C)
        d += C
    P["gates--main/weights"], P["experts--main/weights"], P["experts--main/biases"] = _moe_params(rs, d, V, M)
    main, sup = np_ref.deep_combine_chain_model(x, P, L, M)
    tm, ts = torch_ref.deep_combine_chain(T(x), {k: T(v) for k, v in P.items()}, L, M)
    assert sup.shape == (B, L * V) and np.abs(main - tm.numpy()).max() < 1e-13 and np.abs(sup - ts.numpy()).max() < 1e-13
    y = rs.rand(B, V) < 0.3
    sl = np_ref.get_support_label_type(y, "label,label")
    assert sl.shape == (B, 2 * V) and (sl[:, V:] == y).all()
    l = np_ref.multitask_cross_entropy_loss(main, sup, y, sl, 0.1)
    assert l == pytest.approx(0.9 * np_ref.cross_entropy_loss(main, y) + 0.1 * np_ref.cross_entropy_loss(sup, sl))


def test_netvlad_and_dbof():
    rs = np.random.RandomState(7)
    B, F, D, K, Hf = 3, 5, 6, 4, 5
    x = np_ref.l2_normalize(rs.randn(B, F, D))
    nf = np.array([5, 2, 1])
    x = x * (np.arange(F)[None, :, None] < nf[:, None, None])
    Wc, bc, c = rs.randn(D, K), rs.randn(K), rs.randn(K, D)
    v, a = np_ref.netvlad(x, nf, Wc, bc, c)
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0) and (a[1, 2:] == 0).all()
    vt = torch_ref.netvlad(T(x), T(nf), T(Wc), T(bc), T(c)).numpy()
    assert np.abs(v - vt).max() < 1e-13
    Wh, bh, Wgt, bgt = rs.randn(K * D, Hf), rs.randn(Hf), rs.randn(Hf, Hf), rs.randn(Hf)
    h = np_ref.netvlad_hidden(x, nf, Wc, bc, c, Wh, bh, Wgt, bgt)
    ht = torch_ref.netvlad_hidden(T(x), T(nf), T(Wc), T(bc), T(c), T(Wh), T(bh), T(Wgt), T(bgt)).numpy()
    assert np.abs(h - ht).max() < 1e-13
    xs, idx = np_ref.sample_random_frames(x, nf, rs.rand(B, 3))
    assert (idx < nf[:, None]).all() and xs.shape == (B, 3, D)
    xs2, idx2 = np_ref.sample_random_sequence(x, nf, rs.rand(B), 3)
    assert (idx2 <= np.maximum(nf - 1, 0)[:, None]).all()
    hd = np_ref.dbof_model_hidden(xs, rs.randn(D, 8), rs.randn(8), rs.randn(8, 4), rs.randn(4))
    assert hd.shape == (B, 4) and hd.min() >= 0 and hd.max() <= 6


def test_optimizer_slice():
    """LR staircase, per-tensor clip on (g + l2 w), TF-Adam with eps outside the sqrt; np vs torch restatement."""
    assert np_ref.exponential_decay(0.01, 3906, 1024, 4000000, 0.95) == 0.01
    assert np_ref.exponential_decay(0.01, 3907, 1024, 4000000, 0.95) == pytest.approx(0.0095)
    assert torch_ref.exponential_decay(0.01, 2 * 3907, 1024) == pytest.approx(0.01 * 0.95 ** 2)
    g = np.array([3.0, 4.0])
    assert np.allclose(np_ref.clip_by_norm(g, 1.0), g / 5) and np.allclose(np_ref.clip_by_norm(g * 0.1, 1.0), g * 0.1)
    rs = np.random.RandomState(8)
    params = {"a/weights": rs.randn(5, 4), "a/biases": rs.randn(4)}
    tparams = {k: T(v).clone().requires_grad_(True) for k, v in params.items()}
    opt = torch_ref.TFAdam(tparams, ["a/weights"], base_lr=0.01, batch_size=8, l2=1e-2, clip=1.0)
    state = {}
    for step in range(3):
        grads = {k: rs.randn(*v.shape) * (3.0 if step == 0 else 0.05) for k, v in params.items()}
        for k in tparams:
            tparams[k].grad = T(grads[k]).clone()
        opt.step()
        params, state = np_ref.train_step_update(params, grads, state, step, 0.01, 8, {"a/weights"}, l2_penalty=1e-2, clip=1.0)
        for k in params:
            assert np.abs(params[k] - tparams[k].detach().numpy()).max() < 1e-12
    th, m, v = np_ref.adam_step(np.zeros(1), np.zeros(1), np.zeros(1), np.ones(1), 0.1, 1)
    assert th[0] == pytest.approx(-0.1 * np.sqrt(1 - 0.999) / (1 - 0.9) * 0.1 / (np.sqrt(0.001) + 1e-8))


def test_moe_train_step_cpu_runs():
    st = torch_ref.MoeTrainStepCPU(D=16, V=11, M=2, batch_size=8, dtype=torch.float64)
    x = torch.randn(8, 16, dtype=torch.float64)
    y = torch.rand(8, 11) < 0.3
    l0, _ = st.step(x, y)
    for _ in range(20):
        l1, p = st.step(x, y)
    assert l1 < l0 and p.shape == (8, 11)
