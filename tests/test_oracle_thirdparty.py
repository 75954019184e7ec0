"""Independent pins for the oracle where the reference itself cannot run (TF-1.0 is an absent, un-vendored dependency:
SURVEY.md 8c "parity unpinned"): the restatements of the TF cells are checked against PyTorch's OWN implementations of the
same published algorithms under the documented parameter mapping.  This does not pin TF's behaviour; it pins that the
restatement is the standard algorithm (LSTM cell, Adam, softmax / sigmoid mixtures, l2-normalise, clip-by-norm)."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref


def test_lstm_restatement_equals_torch_nn_lstm():
    """BasicLSTMCell (gate order i, j, f, o; forget_bias added to f; W = [x-rows ; h-rows] x 4H) under dynamic_rnn with
    sequence_length (state copy-through, zero outputs)  ==  torch.nn.LSTM (gate order i, f, g, o; separate W_ih / W_hh) on
    packed sequences, 2 layers."""
    torch.manual_seed(0)
    B, F, D, H, L = 5, 9, 7, 6, 2
    x = torch.randn(B, F, D, dtype=torch.float64)
    nf = torch.tensor([9, 1, 4, 7, 9])
    x = x * (torch.arange(F)[None, :, None] < nf[:, None, None])
    layers = []
    lstm = torch.nn.LSTM(D, H, num_layers=L, batch_first=True).double()
    for l in range(L):
        din = D if l == 0 else H
        W = torch.randn(din + H, 4 * H, dtype=torch.float64) * 0.4
        b = torch.randn(4 * H, dtype=torch.float64) * 0.2
        layers.append((W, b))
        i, j, f, o = W.chunk(4, 1)
        bi, bj, bf, bo = b.chunk(4)
        with torch.no_grad():                                   # TF (i, j, f, o) -> torch (i, f, g, o); forget_bias = 1 folded in
            Wt = torch.cat([i, f, j, o], 1)
            getattr(lstm, "weight_ih_l%d" % l).copy_(Wt[:din].t())
            getattr(lstm, "weight_hh_l%d" % l).copy_(Wt[din:].t())
            getattr(lstm, "bias_ih_l%d" % l).copy_(torch.cat([bi, bf + 1.0, bj, bo]))
            getattr(lstm, "bias_hh_l%d" % l).zero_()
    out, c, h = torch_ref.lstm_stack(x, nf, layers)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, nf, batch_first=True, enforce_sorted=False)
    po, (hn, cn) = lstm(packed)
    po, _ = torch.nn.utils.rnn.pad_packed_sequence(po, batch_first=True, total_length=F)
    assert float((out - po.detach()).abs().max()) < 1e-12                # outputs, zero past sequence_length
    for l in range(L):
        assert float((h[l] - hn[l].detach()).abs().max()) < 1e-12 and float((c[l] - cn[l].detach()).abs().max()) < 1e-12
    # the numpy restatement agrees as well
    st = np_ref.lstm_model_state(x.numpy(), nf.numpy(), [(W.numpy(), b.numpy()) for W, b in layers])
    ref = torch.cat([t for pair in zip(cn, hn) for t in pair], 1).detach().numpy()
    assert np.abs(st - ref).max() < 1e-12


def test_tf_adam_restatement_vs_torch_adam():
    """tf.train.AdamOptimizer: theta -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).  torch.optim.Adam puts epsilon
    inside the bias-corrected denominator, so the two coincide as eps -> 0: with eps = 1e-30 they must agree to rounding,
    and with the default eps = 1e-8 they must differ exactly by the documented epsilon placement."""
    torch.manual_seed(1)
    w0 = torch.randn(50, dtype=torch.float64)
    grads = [torch.randn(50, dtype=torch.float64) for _ in range(6)]
    for eps in (1e-30, 1e-8):
        wt = w0.clone().requires_grad_(True)
        opt = torch.optim.Adam([wt], lr=0.01, betas=(0.9, 0.999), eps=eps)
        wo = w0.clone().requires_grad_(True)
        ours = torch_ref.TFAdam({"w": wo}, regularised=[], base_lr=0.01, batch_size=1, clip=0, eps=eps)
        mn, vn, wn = np.zeros(50), np.zeros(50), w0.numpy().copy()
        for t, g in enumerate(grads, 1):
            wt.grad = g.clone()
            opt.step()
            wo.grad = g.clone()
            ours.step()
            wn, mn, vn = np_ref.adam_step(wn, mn, vn, g.numpy(), 0.01, t, eps=eps)
        assert np.abs(wn - wo.detach().numpy()).max() < 1e-15                      # numpy and torch restatements agree
        if eps == 1e-30:
            assert float((wo.detach() - wt.detach()).abs().max()) < 1e-12
        else:
            m = torch.zeros_like(w0); v = torch.zeros_like(w0); w = w0.clone()
            for t, g in enumerate(grads, 1):                    # the documented TF formula, written out independently
                m = 0.9 * m + 0.1 * g
                v = 0.999 * v + 0.001 * g * g
                w = w - 0.01 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (v.sqrt() + eps)
            assert float((wo.detach() - w).abs().max()) < 1e-15
            assert float((wo.detach() - wt.detach()).abs().max()) > 0                # and it is NOT torch's epsilon placement


def test_elementwise_restatements_vs_torch_functional():
    rs = np.random.RandomState(3)
    x = rs.randn(6, 11)
    assert np.abs(np_ref.l2_normalize(x) - torch.nn.functional.normalize(torch.from_numpy(x), dim=1, eps=1e-6).numpy()).max() < 1e-12
    z = rs.randn(4, 9) * 3
    assert np.abs(np_ref.softmax(z, axis=1) - torch.softmax(torch.from_numpy(z), 1).numpy()).max() < 1e-15
    assert np.abs(np_ref.sigmoid(z) - torch.sigmoid(torch.from_numpy(z)).numpy()).max() < 1e-15
    # MoE mixture = sum_m softmax(gates)[m] * sigmoid(experts)[m] over the first M of M+1 gates (moe_model.py:54-64)
    B, D, V, M = 3, 5, 4, 2
    xx, Wg, We, be = rs.randn(B, D), rs.randn(D, V * (M + 1)), rs.randn(D, V * M), rs.randn(V * M)
    g = torch.softmax(torch.from_numpy(xx @ Wg).view(B, V, M + 1), 2)[:, :, :M]
    e = torch.sigmoid(torch.from_numpy(xx @ We + be).view(B, V, M))
    assert np.abs(np_ref.moe_model(xx, Wg, We, be, M) - (g * e).sum(2).numpy()).max() < 1e-14
    # clip_by_norm (utils.py:164-174) == torch.nn.utils.clip_grad_norm_ per tensor, except torch's +1e-6 in the denominator
    gr = rs.randn(40) * 3
    t = torch.from_numpy(gr.copy()).requires_grad_(True)
    t.grad = torch.from_numpy(gr.copy())
    torch.nn.utils.clip_grad_norm_([t], 1.0)
    assert np.abs(np_ref.clip_by_norm(gr, 1.0) - t.grad.numpy()).max() < 2e-6
    # probability-space cross entropy with eps = 1e-5 -> BCE as eps -> 0
    p = rs.rand(5, 7) * 0.9 + 0.05
    y = (rs.rand(5, 7) < 0.3).astype(np.float64)
    bce = torch.nn.functional.binary_cross_entropy(torch.from_numpy(p), torch.from_numpy(y), reduction="none").sum(1).mean().item()
    assert abs(np_ref.cross_entropy_loss(p, y) - bce) < 1e-3 * bce
    assert abs(np_ref.cross_entropy_loss(p, y, eps=0.0) - bce) < 1e-12 * bce


def test_philox_known_answers_and_streams():
    """oracle/philox.py against the Random123 known-answer vectors for philox4x32-10, then the conventions built on it."""
    from oracle import philox
    from yt8m_amd.variables import random_seed

    def kat(ctr, key):
        r = philox.philox4x32_10(*[np.array([v], dtype=np.uint64) for v in ctr], key[0], key[1])
        return [int(v[0]) for v in r]

    assert kat((0, 0, 0, 0), (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert kat((0xffffffff,) * 4, (0xffffffff,) * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert kat((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    w = philox.words(64, 7)
    assert np.array_equal(w[5:41], philox.words(36, 7, offset=5))               # offset = position in the logical tensor
    blk = philox.philox4x32_10(np.array([3], dtype=np.uint64), np.array([0], dtype=np.uint64), np.array([0], dtype=np.uint64),
                               np.array([0], dtype=np.uint64), 7, 0)
    assert [int(w[12 + k]) for k in range(4)] == [int(b[0]) for b in blk]       # element e -> word e & 3 of block e >> 2
    m = philox.dropout_mask(1 << 18, 0.7, 99)
    assert abs(m.mean() - 0.7) < 4e-3 and philox.dropout_mask(1000, 1.0, 5).all()
    x = np.arange(1, 9, dtype=np.float32)
    d = philox.dropout(x, 0.5, 3)
    assert set(np.unique(d / x)) <= {0.0, 2.0}
    z = philox.normal(1 << 18, 11)
    assert abs(z.mean()) < 6e-3 and abs(z.std() - 1) < 6e-3
    seeds = {random_seed(0, r, s, c) for r in range(2) for s in range(50) for c in range(8)}
    assert len(seeds) == 800 and all(0 <= v < 1 << 64 for v in seeds)
    assert random_seed(3, 1, 2, 0) == random_seed(3, 1, 2, 0) != random_seed(4, 1, 2, 0)
