"""-m gpu: round-2 additions -- device PERR / mAP in the evaluation path, checkpoint resume equivalence on the device,
model-provided regularization_loss / update_ops, the reference-signature clip_gradient_norms."""
import numpy as np
import pytest
import torch

import yt8m_amd.checkpoint as checkpoint
import yt8m_amd.eval_util as eval_util
import yt8m_amd.frame_level_models as flm
import yt8m_amd.ops as ops
import yt8m_amd.train as train
import yt8m_amd.utils as utils
import yt8m_amd.video_level_models as vlm
from yt8m_amd.variables import reset_default_graph

pytestmark = pytest.mark.gpu


def test_perr_rows_matches_host(dev):
    """yt8m_perr_rows vs calculate_precision_at_equal_recall_rate (W/eval_util.py:74-99), incl. rows without labels, rows
    whose top classes have score <= 0, and the full V = 4716."""
    rs = np.random.RandomState(3)
    for B, V, rate in [(7, 30, 0.2), (64, 4716, 3.4 / 4716), (5, 257, 0.5)]:
        p = rs.rand(B, V).astype(np.float32)
        p[1] -= 0.7                                   # mostly non-positive scores: the (score > 0) filter matters
        y = rs.rand(B, V) < rate
        y[0] = False                                  # a video without labels
        got = ops.perr_rows(torch.from_numpy(p).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
        for r in range(B):
            exp = eval_util.calculate_precision_at_equal_recall_rate(p[r:r + 1], y[r:r + 1])
            assert abs(float(got[r]) - exp) < 1e-6, (B, V, r, got[r], exp)


def test_device_metrics_equal_host_on_all_keys(dev):
    """EvaluationMetrics.accumulate_device == accumulate on avg_hit_at_one, avg_perr, avg_loss, gap and aps."""
    rs = np.random.RandomState(11)
    B, V, k = 96, 300, 20
    host, devm = eval_util.EvaluationMetrics(V, k), eval_util.EvaluationMetrics(V, k)
    for it in range(3):
        p = rs.rand(B, V).astype(np.float32)
        y = rs.rand(B, V) < 0.02
        loss = float(rs.rand())
        host.accumulate(p, y, np.array([loss]))
        out = devm.accumulate_device(torch.from_numpy(p).to(dev), torch.from_numpy(y).to(dev), loss)
        assert set(out) == {"hit_at_one", "perr", "loss"}
    a, b = host.get(), devm.get()
    assert set(a) == set(b) == {"avg_hit_at_one", "avg_perr", "avg_loss", "aps", "gap"}
    for key in ("avg_hit_at_one", "avg_perr", "avg_loss", "gap"):
        assert abs(a[key] - b[key]) < 1e-6, (key, a[key], b[key])
    assert np.allclose(a["aps"], b["aps"], atol=1e-6)
    assert b["avg_perr"] > 0 and max(b["aps"]) > 0


def _toy_batches(dev, n, B, D, V, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    return [(torch.rand((B, D), device=dev, generator=gen) * 4 - 2, torch.rand((B, V), device=dev, generator=gen) < 0.05)
            for _ in range(n)]


def test_checkpoint_resume_is_bitwise_uninterrupted(dev, tmp_path, flags):
    """save -> restore into a FRESH graph (after one forward pass, before any step: the documented flow) -> k more steps
    == k uninterrupted steps, bit for bit: parameters, Adam slots, global_step (W/train.py:654-677,728)."""
    B, D, V = 64, 96, 257
    data = _toy_batches(dev, 8, B, D, V, 5)

    def fresh(seed):
        g = reset_default_graph(device=dev, seed=seed)
        return g, train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)

    g1, t1 = fresh(0)
    for x, y in data[:4]:
        t1.step(x, y)
    path = checkpoint.save(t1, str(tmp_path))
    for x, y in data[4:]:
        t1.step(x, y)
    g2, t2 = fresh(123)                                  # different initial weights: everything must come from the file
    t2.forward(data[0][0], data[0][1])                   # creates the variables; graph not finalized yet
    assert not g2.finalized
    checkpoint.restore(t2, path)
    assert g2.finalized and t2.global_step == 4
    for x, y in data[4:]:
        t2.step(x, y)
    assert t2.global_step == t1.global_step == 8
    assert torch.equal(g1.params, g2.params)
    assert torch.equal(g1.adam_m, g2.adam_m) and torch.equal(g1.adam_v, g2.adam_v)


def test_checkpoint_resume_frame_level_model(dev, tmp_path, flags):
    """The same for an LSTM model (TF variable names RNN/multi_rnn_cell/cell_<l>/basic_lstm_cell/...)."""
    flags.lstm_cells = "128"
    B, F, D, V = 8, 12, 64, 33
    gen = torch.Generator(device=dev).manual_seed(2)
    xs = [torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8) for _ in range(4)]
    ys = [torch.rand((B, V), device=dev, generator=gen) < 0.1 for _ in range(4)]
    nf = torch.tensor([12, 3, 7, 12, 1, 9, 12, 5], device=dev, dtype=torch.int32)

    def fresh(seed):
        g = reset_default_graph(device=dev, seed=seed)
        return g, train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)

    g1, t1 = fresh(0)
    for i in range(2):
        t1.step(xs[i], ys[i], nf)
    path = checkpoint.save(t1, str(tmp_path))
    for i in range(2, 4):
        t1.step(xs[i], ys[i], nf)
    g2, t2 = fresh(9)
    t2.forward(xs[0], ys[0], nf)
    checkpoint.restore(t2, path)
    for i in range(2, 4):
        t2.step(xs[i], ys[i], nf)
    assert sorted(g1.vars) == sorted(g2.vars) and "RNN/multi_rnn_cell/cell_1/basic_lstm_cell/weights" in g2.vars
    assert torch.equal(g1.params, g2.params) and torch.equal(g1.adam_m, g2.adam_m) and torch.equal(g1.adam_v, g2.adam_v)


def test_model_regularization_loss_and_update_ops(dev, flags):
    """result["regularization_loss"] enters the final loss with --regularization_penalty, result["update_ops"] run before
    the gradient step (W/train.py:435-456)."""
    B, D, V = 32, 48, 65
    (x, y), = _toy_batches(dev, 1, B, D, V, 1)
    ran = []

    class WithReg(vlm.LogisticModel):
        def create_model(self, model_input, vocab_size, **kw):
            out = super().create_model(model_input, vocab_size, **kw)
            out["regularization_loss"] = (out["predictions"] ** 2).sum() * 0.01
            out["update_ops"] = [lambda: ran.append(1)]
            return out

    def grads(model, penalty):
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(model, batch_size=B, graph=g, regularization_penalty=penalty, base_learning_rate=0.0)
        tg.step(x, y)
        return g.grads.clone()

    base = grads(vlm.LogisticModel(), 1)
    ran.clear()
    reg2 = grads(WithReg(), 2)
    assert ran == [1]
    reg0 = grads(WithReg(), 0)
    assert not torch.allclose(base, reg2) and torch.allclose(base, reg0, atol=1e-7)
    # linear in the penalty: g(2) - g(0) == 2 * (g(1) - g(0))
    reg1 = grads(WithReg(), 1)
    assert torch.allclose(reg2 - reg0, 2 * (reg1 - reg0), rtol=1e-3, atol=1e-8)


def test_clip_gradient_norms_reference_signature(dev):
    """utils.clip_gradient_norms([(grad, var)], max_norm) -> [(clipped, var)] == the fused per-tensor clip (W/utils.py:164-174)."""
    g1 = torch.full((10,), 3.0, device=dev)
    g2 = torch.full((4,), 0.1, device=dev)
    out = utils.clip_gradient_norms([(g1, "a"), (None, "b"), (g2, "c")], 1.0)
    assert [v for _, v in out] == ["a", "b", "c"] and out[1][0] is None
    assert abs(float(out[0][0].norm()) - 1.0) < 1e-6 and torch.equal(out[2][0], g2)


# ---- persistent LSTM recurrence (csrc/lstm_persist.hip) vs the per-step kernels and the oracle -------------------------------
def _stack_run(dev, B, F, D, H, L_, chunks, nf, persist, seed=0):
    import yt8m_amd.seq_ops as seq_ops
    from yt8m_amd.variables import xavier_uniform, zeros
    old = (seq_ops.PERSIST, seq_ops.PERSIST_BWD, seq_ops.PERSIST_CHECK)
    seq_ops.PERSIST, seq_ops.PERSIST_BWD, seq_ops.PERSIST_CHECK = persist, persist, True
    try:
        g = reset_default_graph(device=dev, seed=seed)
        g.begin_step()
        gen = torch.Generator(device=dev).manual_seed(seed)
        x = (torch.rand((F, B, D), device=dev, generator=gen) - 0.5)
        wb, d_in = [], D
        for l in range(L_):
            wb.append((g.get_variable("l%d/w" % l, (d_in + H, 4 * H), xavier_uniform), g.get_variable("l%d/b" % l, (4 * H,), zeros)))
            d_in = H
        g.finalize()
        x.requires_grad_(True)
        out, finals = seq_ops.lstm_stack(x, nf, wb, chunks=chunks)
        res = [out] + [t for p in finals for t in p]
        gen2 = torch.Generator(device=dev).manual_seed(7)
        sum((r * torch.rand(r.shape, device=dev, generator=gen2)).sum() for r in res).backward()
        torch.cuda.synchronize()
        P = [(w.data.detach().cpu().double(), b.data.detach().cpu().double()) for w, b in wb]
        return [r.detach().cpu() for r in res], [x.grad.detach().cpu(), g.grads.detach().cpu().clone()], x.detach().cpu().double(), P
    finally:
        seq_ops.PERSIST, seq_ops.PERSIST_BWD, seq_ops.PERSIST_CHECK = old


@pytest.mark.parametrize("B,F,D,H,L_,chunks,ragged", [
    (8, 9, 64, 256, 2, 1, True),          # one tile per workgroup: no prefetch ("cold" mode), H = 256
    (50, 12, 96, 512, 2, 2, True),        # rows padded to a multiple of 16, two time chunks
    (128, 16, 1152, 1024, 2, 1, False),   # BASELINE configs[3] shape (short sequence)
    (128, 24, 1152, 1024, 2, 4, True),    # ragged num_frames incl. 0 and F, four chunks
    (512, 6, 128, 1024, 1, 1, True),      # 32 tiles: several per workgroup
    (256, 40, 1152, 1024, 2, 4, True),    # the batch sweep's B = 256 (eight tiles per workgroup), F = 40, four chunks, ragged
])
def test_persistent_lstm_matches_step_kernels(dev, B, F, D, H, L_, chunks, ragged, honour_lstm_chunks):
    """Forward outputs / final states and all gradients of the persistent recurrence agree with the per-step kernels (which are
    checked against the oracle elsewhere) to fp32 rounding: the only arithmetic differences are the K summation order and the
    v_exp_f32 / v_rcp_f32 gate non-linearities (<= 1.5e-7 absolute each)."""
    import yt8m_amd._lib as L
    if not L.lib().yt8m_lstm_persist_supported(B, H):
        pytest.skip("persistent recurrence not available for this shape / device")
    nf = None
    if ragged:
        nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(3), dtype=torch.int32)
        nf[0], nf[1] = F, 0
    a, ga, _, _ = _stack_run(dev, B, F, D, H, L_, chunks, nf, True)
    b, gb, _, _ = _stack_run(dev, B, F, D, H, L_, chunks, nf, False)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) < 5e-6
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-9


def test_persistent_lstm_vs_fp64_oracle_full_width(dev, lstm_partition):
    """H = 1024 (the BASELINE width) forward AND backward against fp64 autograd of the oracle restatement on three videos'
    worth of rows per tile pattern: B = 20 (two 16-row tiles, the second padded), F = 10, ragged lengths."""
    from oracle import torch_ref
    B, F, D, H = 20, 10, 96, 1024
    nf = torch.tensor([10, 0, 3, 10, 7, 1, 9, 10, 2, 5, 10, 4, 6, 8, 10, 10, 3, 0, 10, 7], device=dev, dtype=torch.int32)
    res, grads, x64, P = _stack_run(dev, B, F, D, H, 2, 1, nf, True)
    xs = x64.transpose(0, 1).clone().requires_grad_(True)          # oracle takes [B,F,D]
    layers = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in P]
    out, c, h = torch_ref.lstm_stack(xs, nf.cpu(), layers)
    ref = [out.transpose(0, 1)] + [t for pair in zip(c, h) for t in pair]
    for u, v in zip(res, ref):
        assert float((u.double() - v).abs().max()) < 2e-5
    gen2 = torch.Generator(device=dev).manual_seed(7)
    loss = sum((r * torch.rand(r.shape, device=dev, generator=gen2).cpu().double()).sum() for r in ref)
    loss.backward()
    dx_ref = xs.grad.transpose(0, 1)
    assert float((grads[0].double() - dx_ref).abs().max()) <= 1e-4 * float(dx_ref.abs().max())
    gW = torch.cat([t.grad.reshape(-1) for pair in layers for t in pair])
    assert float((grads[1].double() - gW).abs().max()) <= 1e-4 * float(gW.abs().max())


# ---- BASELINE configs[4] in its own dtype: the bf16 composite end to end -------------------------------------
def test_config5_composite_bf16_engaged(dev, flags, monkeypatch):
    """GatedNetVLADAttentionChainModel with --compute_dtype=bfloat16 at B*A = 512 chain rows: the bf16 MFMA GEMMs, the fused
    bf16 mixing backward and the single-f16-operand (nsplit = 1) NetVLAD pooling all switch on together.  Checked against the
    oracle with the bf16 operand rounding of the MoE heads emulated in VALUE (oracle/torch_ref.py bf16_heads): predictions,
    support predictions, loss; gradients against the fp32 oracle gradients by direction and scale (bf16 products: ~1e-2)."""
    import yt8m_amd.losses as losses
    import yt8m_amd.seq_ops as seq_ops
    from oracle import np_ref, torch_ref
    rs = np.random.RandomState(23)
    B, F, Dm, K, Hf, V, A, L = 64, 32, 256, 64, 128, 600, 8, 2
    flags.compute_dtype = "bfloat16"
    flags.netvlad_cluster_size, flags.netvlad_hidden_size, flags.lstm_attentions = K, Hf, A
    flags.deep_chain_layers, flags.deep_chain_relu_cells = L, 16
    flags.support_type = ",".join(["label"] * L)
    flags.support_loss_percent = 0.1
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    y = rs.rand(B, V) < 0.02
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    x64 = np_ref.dequant_l2norm_folded(q, nf)
    calls = {"bf16_gemm": 0, "nsplit": set(), "fused_mix": 0}
    real_gemm, real_nv, real_mix = ops.gemm_bf16_nt_grouped, seq_ops.netvlad_fwd_u8, ops._moe_head_bwd_bf16_fused

    def spy_gemm(items):
        calls["bf16_gemm"] += len(items)
        return real_gemm(items)

    def spy_nv(*a, **kw):
        calls["nsplit"].add(kw.get("nsplit", a[4] if len(a) > 4 else 2))
        return real_nv(*a, **kw)

    def spy_mix(*a, **kw):
        calls["fused_mix"] += 1
        return real_mix(*a, **kw)

    monkeypatch.setattr(ops, "gemm_bf16_nt_grouped", spy_gemm)
    monkeypatch.setattr(seq_ops, "netvlad_fwd_u8", spy_nv)
    monkeypatch.setattr(ops, "_moe_head_bwd_bf16_fused", spy_mix)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.GatedNetVLADAttentionChainModel(), batch_size=B, graph=g, multitask=True,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss())
    xd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    tg.forward(xd, yd, nfd)
    g.finalize()
    P = {}
    for k, v in g.vars.items():
        P[k] = (rs.randn(*v.shape) * (0.4 if "netvlad" in k or "attention" in k else 1.0 / np.sqrt(v.shape[0]))).astype(np.float32)
        v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))
    calls["bf16_gemm"] = 0
    res = tg.forward(xd, yd, nfd)
    loss = tg.loss(res, yd)
    loss.backward()
    assert calls["bf16_gemm"] >= 3 * (L + 1) and calls["fused_mix"] >= L + 1 and 1 in calls["nsplit"], calls
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.gated_netvlad_attention_chain(T(x64), torch.from_numpy(nf), tp, L, 2, A, bf16_heads=True)
    pm, ps = res["predictions"].detach().cpu().double(), res["support_predictions"].detach().cpu().double()
    assert float((pm - main.detach()).abs().max()) < 5e-3 and float((ps - sup.detach()).abs().max()) < 5e-3
    yt = T(y)
    lr = 0.9 * torch_ref.cross_entropy(main, yt) + 0.1 * torch_ref.cross_entropy(sup, yt.repeat(1, L))
    assert abs(float(loss.detach()) - lr.item()) < 5e-3 * abs(lr.item())
    # gradients vs the fp32 oracle (no emulation of the backward roundings): direction and scale
    main32, sup32 = torch_ref.gated_netvlad_attention_chain(T(x64), torch.from_numpy(nf), tp, L, 2, A)
    (0.9 * torch_ref.cross_entropy(main32, yt) + 0.1 * torch_ref.cross_entropy(sup32, yt.repeat(1, L))).backward()
    for k, v in g.vars.items():
        if not v.trainable or tp[k].grad is None:
            continue
        a, b = v.grad.detach().cpu().double().reshape(-1), tp[k].grad.reshape(-1)
        nb = float(b.norm())
        if nb < 1e-12:
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        assert cos > 0.99 and abs(float(a.norm()) / nb - 1.0) < 0.05, (k, cos, float(a.norm()) / nb)


def test_config5_composite_bf16_full_batch_properties(dev, flags):
    """configs[4] at its per-GPU batch (B = 1024 videos x 300 frames x 1152, V = 4716, bf16): one training step runs on the bf16
    paths and keeps the size-independent properties -- probabilities in [0, 1], finite loss, predictions invariant under a
    permutation of the videos (each video is pooled independently), padded frames beyond num_frames do not matter."""
    import yt8m_amd.losses as losses
    flags.compute_dtype = "bfloat16"
    flags.deep_chain_layers, flags.deep_chain_relu_cells = 3, 128
    flags.support_type = ",".join(["label"] * 3)
    B, F, D, V = 1024, 300, 1152, 4716
    gen = torch.Generator(device=dev).manual_seed(5)
    q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    nf = torch.randint(1, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.GatedNetVLADAttentionChainModel(), batch_size=B, graph=g, multitask=True,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss())
    out = tg.step(q, y, nf)
    p = out["predictions"]
    assert torch.isfinite(out["loss"]) and float(p.min()) >= 0.0 and float(p.max()) <= 1.0
    with torch.no_grad():
        p0 = tg.predict(q[:64], nf[:64], vocab_size=V)
        perm = torch.randperm(64, device=dev, generator=gen)
        p1 = tg.predict(q[:64][perm], nf[:64][perm], vocab_size=V)
        assert float((p0[perm] - p1).abs().max()) < 2e-3          # bf16 tile boundaries move with the row order
        q2 = q[:64].clone()
        mask = torch.arange(F, device=dev)[None, :] >= nf[:64, None]
        q2[mask] = 255                                            # garbage in the padded frames
        p2 = tg.predict(q2, nf[:64], vocab_size=V)
        assert float((p0 - p2).abs().max()) < 2e-3


# ---- DbofModel pieces on the device (csrc/dbof.hip) ---------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_sample_frames_bit_exact_vs_oracle(dev, dtype):
    """SampleRandomFrames / SampleRandomSequence (W/model_utils.py:23-70): indices and gathered rows equal the oracle's on the
    same Philox uniforms, bit for bit; ragged num_frames incl. 0, 1 and F; S larger than some videos."""
    from oracle import np_ref, philox
    rs = np.random.RandomState(31)
    B, F, D, S = 9, 40, 24, 7
    nf = np.array([40, 0, 1, 3, 7, 8, 25, 40, 2], dtype=np.int32)
    x = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8) if dtype == "u8" else rs.randn(B, F, D).astype(np.float32)
    xd, nfd = torch.from_numpy(x).to(dev), torch.from_numpy(nf).to(dev)
    seed = 0x1234567890ABCDEF
    out, idx = ops.sample_frames(xd, nfd, S, 0, seed, return_index=True)
    u = philox.uniform01(B * S, seed).reshape(B, S)
    ref, ridx = np_ref.sample_random_frames(x, nf, u)
    ridx = np.clip(ridx, 0, F - 1)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(out.cpu().numpy(), np.take_along_axis(x, ridx[:, :, None].astype(np.int64), axis=1))
    assert (ridx[nf > 0] < nf[nf > 0, None]).all()
    out, idx = ops.sample_frames(xd, nfd, S, 1, seed, return_index=True)
    ref, ridx = np_ref.sample_random_sequence(x, nf, philox.uniform01(B, seed), S)
    ridx = np.clip(ridx, 0, F - 1)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(out.cpu().numpy(), np.take_along_axis(x, ridx[:, :, None].astype(np.int64), axis=1))


@pytest.mark.parametrize("shape", [(5, 6, 70), (3, 8, 72), (2, 19, 64), (4, 1, 12), (2, 30, 33)])
@pytest.mark.parametrize("method", ["max", "average"])
def test_frame_pool_with_ties(dev, method, shape):
    """FramePooling over relu6-clamped activations (ties at 0 and 6 everywhere): forward and the tie-splitting gradient vs torch;
    scalar and 16-byte column paths (C % 4), S below / above the 16 frames the backward keeps in registers, S = 1."""
    rs = np.random.RandomState(2)
    x = np.clip(rs.randn(*shape) * 4 + 3, 0, 6).astype(np.float32)
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    w = torch.from_numpy(rs.randn(shape[0], shape[2]).astype(np.float32))
    out = ops.frame_pool(xd, method)
    (out * w.to(dev)).sum().backward()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    ref = xr.amax(1) if method == "max" else xr.mean(1)
    (ref * w.double()).sum().backward()
    assert float((out.detach().cpu().double() - ref.detach()).abs().max()) < 1e-6
    assert float((xd.grad.cpu().double() - xr.grad).abs().max()) < 1e-6


@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_fwd_bwd_vs_oracle(dev, training):
    """slim.batch_norm on [N, C] rows (C not a multiple of 64, N not a multiple of 4): output, moving averages, dx, dgamma, dbeta."""
    from oracle import torch_ref
    from yt8m_amd.variables import ones, zeros
    rs = np.random.RandomState(4)
    N, C, eps, decay = 203, 150, 1e-3, 0.999
    x = (rs.randn(N, C) * rs.rand(C) * 3 + rs.randn(C)).astype(np.float32)
    g = reset_default_graph(device=dev, seed=0)
    g.begin_step()
    gamma, beta = g.get_variable("bn/gamma", (C,), ones), g.get_variable("bn/beta", (C,), zeros)
    mm, mv = g.get_variable("bn/moving_mean", (C,), zeros, trainable=False), g.get_variable("bn/moving_variance", (C,), ones, trainable=False)
    g.finalize()
    gam, bet = (rs.rand(C) + 0.5).astype(np.float32), rs.randn(C).astype(np.float32)
    mm0, mv0 = rs.randn(C).astype(np.float32), (rs.rand(C) + 0.5).astype(np.float32)
    for v, a in ((gamma, gam), (beta, bet), (mm, mm0), (mv, mv0)):
        v.data.copy_(torch.from_numpy(a).to(dev))
    xd = torch.from_numpy(x).to(dev).requires_grad_(True)
    w = rs.randn(N, C).astype(np.float32)
    y = ops.batch_norm(xd, gamma, beta, mm, mv, training, eps, decay)
    (y * torch.from_numpy(w).to(dev)).sum().backward()
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gt, bt = torch.from_numpy(gam).double().requires_grad_(True), torch.from_numpy(bet).double().requires_grad_(True)
    if training:
        yr, mu, var = torch_ref.batch_norm_train(xt, gt, bt, eps)
        assert np.abs(mm.data.cpu().numpy() - (decay * mm0 + (1 - decay) * mu.detach().numpy())).max() < 1e-6
        assert np.abs(mv.data.cpu().numpy() - (decay * mv0 + (1 - decay) * var.detach().numpy())).max() < 1e-6
    else:
        yr = gt * (xt - torch.from_numpy(mm0).double()) * torch.rsqrt(torch.from_numpy(mv0).double() + eps) + bt
        assert np.array_equal(mm.data.cpu().numpy(), mm0) and np.array_equal(mv.data.cpu().numpy(), mv0)
    (yr * torch.from_numpy(w).double()).sum().backward()
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) < 2e-5
    for got, ref in ((xd.grad, xt.grad), (gamma.grad, gt.grad), (beta.grad, bt.grad)):
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("quantized", [False, True])
def test_dbof_model_with_batch_norm_vs_oracle(dev, flags, quantized):
    """DbofModel with dbof_add_batch_norm on distinct frames (W/all_frame_models/dbof_model.py:36-124): the sampled frames are
    reproduced in the oracle from the op's Philox key, then predictions, loss and EVERY gradient (incl. all gamma / beta) are
    compared; raw uint8 input takes the sample-first path (only the sampled frames are dequantised)."""
    import yt8m_amd.variables as variables
    from oracle import np_ref, philox, torch_ref
    rs = np.random.RandomState(8)
    B, F, Dm, V, S = 12, 20, 18, 11, 6
    flags.dbof_cluster_size, flags.dbof_hidden_size, flags.iterations, flags.dbof_add_batch_norm = 32, 16, S, True
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    y = rs.rand(B, V) < 0.2
    if quantized:
        q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
        x64 = np_ref.dequant_l2norm_folded(q, nf)
        inp = q
    else:
        x64 = np_ref.l2_normalize(rs.randn(B, F, Dm)) * (np.arange(F)[None, :, None] < nf[:, None, None])
        inp = x64.astype(np.float32)
        x64 = inp.astype(np.float64)
    g = reset_default_graph(device=dev, seed=3)
    tcls = __import__("yt8m_amd.feature_transform", fromlist=["x"])
    tg = train.TrainGraph(flm.DbofModel(), batch_size=B, graph=g,
                          transformer_class=tcls.DefaultTransformer if quantized else tcls.IdenticalTransformer)
    xd, yd, nfd = torch.from_numpy(inp).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    tg.forward(xd, yd, nfd)
    g.finalize()
    P = {}
    for k, v in g.vars.items():
        if k.endswith("moving_variance") or k.endswith("gamma"):
            P[k] = (rs.rand(*v.shape) + 0.5).astype(np.float32)
        else:
            P[k] = (rs.randn(*v.shape) * 0.4).astype(np.float32)
        v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))
    res = tg.forward(xd, yd, nfd)
    seed = variables.random_seed(g.seed, g.rank, g._rng_step, 0)          # key of the first random op of this forward pass
    loss = tg.loss(res, yd)
    loss.backward()
    u = philox.uniform01(B * S, seed).reshape(B, S)
    xs, _ = np_ref.sample_random_frames(x64, nf, u)
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    tp = {k: T(v).requires_grad_(True) for k, v in P.items() if "moving" not in k}
    h = torch_ref.dbof_model_bn(T(xs), tp)
    pr = torch_ref.moe(h, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    assert float((res["predictions"].detach().cpu().double() - pr.detach()).abs().max()) < 1e-4
    lr = torch_ref.cross_entropy(pr, T(y))
    assert abs(float(loss.detach()) - lr.item()) < 1e-4 * abs(lr.item())
    lr.backward()
    for k, t in tp.items():
        if t.grad is None:
            continue
        got = g.vars[k].grad.detach().cpu().double()
        assert float((got - t.grad).abs().max()) <= 5e-4 * max(1.0, float(t.grad.abs().max())), k


# ---- uint8 operand path of the hoisted LSTM input projection (csrc/u8proj.hip) -----------------------------
@pytest.mark.parametrize("chunks,B,D", [(1, 10, 72), (3, 10, 72), (1, 32, 64), (3, 32, 64)])
def test_lstm_uint8_projection_matches_float_path_and_oracle(dev, flags, chunks, B, D, honour_lstm_chunks):
    """LstmModel on RAW uint8 frames: the layer-0 projection on exact bf16 operands ((q - 128) x 3-way split of (4/255) W) with
    the row-norm / rank-1 epilogue equals (a) the float path (DefaultTransformer first, fp32 GEMM) on the same weights and (b) the
    fp64 oracle on dequantise + l2-normalise, for predictions, loss and all gradients; ragged num_frames incl. 0 and F.
    B = 32, D = 64 takes the one-plane image + x3 kernel (every chunk starts on a 32-row group, D % 16 == 0); B = 10, D = 72 the
    K-concatenated bf16 copies."""
    from oracle import np_ref, torch_ref
    flags.lstm_cells, flags.lstm_pipeline_chunks = "128", chunks
    rs = np.random.RandomState(41)
    F, V = 9, 17
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = np.resize(np.array([9, 0, 1, 4, 9, 7, 2, 9, 5, 3], dtype=np.int32), B)
    y = rs.rand(B, V) < 0.15
    qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    outs = {}
    P = None
    for fold in (True, False):
        flags.fold_dequant = fold
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
        tg.forward(qd, yd, nfd)
        g.finalize()
        if P is None:
            P = {k: (rs.randn(*v.shape) * 0.3).astype(np.float32) for k, v in g.vars.items()}
        for k, v in g.vars.items():
            v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))
        res = tg.forward(qd, yd, nfd, fuse_loss=False)
        loss = tg.loss(res, yd)
        loss.backward()
        outs[fold] = (res["predictions"].detach().cpu().double(), float(loss.detach()), g.grads.detach().cpu().double().clone(), dict(g.vars))
    pa, la, ga, _ = outs[True]
    pb, lb, gb, vars_b = outs[False]
    assert float((pa - pb).abs().max()) < 1e-5 and abs(la - lb) < 1e-5 * abs(lb)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())
    # fp64 oracle
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    x64 = np_ref.dequant_l2norm_folded(q, nf)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    layers = [(tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    state = torch_ref.lstm_model_state(T(x64), torch.from_numpy(nf), layers)
    pr = torch_ref.moe(state, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    assert float((pa - pr.detach()).abs().max()) < 1e-5
    torch_ref.cross_entropy(pr, T(y)).backward()
    for k, t in tp.items():
        v = vars_b[k]
        got = ga[v.offset:v.offset + v.numel()].view(t.shape)
        assert float((got - t.grad).abs().max()) <= 2e-4 * max(1.0, float(t.grad.abs().max())), k


def test_u8_projection_kernels_exactness(dev):
    """The pieces: (q - 128) as bf16 is exact, the 3-way split reproduces (4/255) W to <= 2^-24 relative, r / x_tm are
    bit-identical to yt8m_dequant_l2norm_u8, and the assembled product equals the fp64 x.W to fp32 rounding at D = 1152."""
    import yt8m_amd._lib as L
    from yt8m_amd.ops import _p, _stream, _bf16_empty
    from oracle import np_ref
    rs = np.random.RandomState(6)
    B, F, D, N = 5, 7, 1152, 256
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = np.array([7, 0, 3, 7, 1], dtype=np.int32)
    W = (rs.randn(D, N) * 0.05).astype(np.float32)
    qd, nfd, Wd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev), torch.from_numpy(W).to(dev)
    lib = L.lib()
    Qb = _bf16_empty(F * B, 3 * D, dev)
    r = torch.empty((F * B,), dtype=torch.float32, device=dev)
    xtm = torch.empty((F, B, D), dtype=torch.float32, device=dev)
    L.check(lib.yt8m_u8_frames_to_bf16_tm(_p(qd), _p(nfd), B, F, D, 1e-12, 3, _p(Qb), Qb.stride(0), _p(xtm), _p(r), _stream()))
    ref = ops.dequant_l2norm(qd, nfd)
    assert torch.equal(xtm, ref.transpose(0, 1).contiguous())
    Qf = Qb.float().cpu().numpy().reshape(F, B, 3, D)
    exp = (q.astype(np.float32) - 128.0).transpose(1, 0, 2) * (np.arange(F)[:, None, None] < nf[None, :, None])
    for j in range(3):
        assert np.array_equal(Qf[:, :, j], exp)
    W3 = _bf16_empty(N, 3 * D, dev)
    L.check(lib.yt8m_split3_bf16_t(_p(Wd), N, D, N, 4.0 / 255.0, _p(W3), W3.stride(0), _stream()))
    w3 = W3.float().cpu().numpy().astype(np.float64).reshape(N, 3, D).sum(1).T
    aw = (W * np.float32(4.0 / 255.0)).astype(np.float64)
    assert np.abs(w3 - aw).max() <= 2.0 ** -24 * np.abs(aw).max()
    z, = ops.gemm_bf16_nt_grouped([dict(A=Qb, B=W3)])
    cs = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.colsum(Wd, cs)
    import yt8m_amd.seq_ops as seq_ops
    L.check(lib.yt8m_rowscale_bias_f32(_p(z), F * B, N, N, _p(r), _p(cs), seq_ops.U8_BETA, None, _stream()))
    x64 = np_ref.dequant_l2norm_folded(q, nf).transpose(1, 0, 2).reshape(F * B, D)
    zr = x64 @ W.astype(np.float64)
    assert np.abs(z.cpu().numpy() - zr).max() < 2e-6 * max(1.0, np.abs(zr).max())


# ---- RCCL in the C ABI (csrc/comm.hip) ------------------------------------------------------------------------------
def test_cabi_comm_single_rank(dev):
    """yt8m_comm_unique_id / init / size / allreduce (sum, mean) / broadcast / destroy on a 1-rank communicator, and the
    GradReducer driving a training step through it: identical to the step without a reducer."""
    import yt8m_amd.parallel as parallel
    comm = parallel.CabiComm(0, 1)
    assert comm.size() == (0, 1)
    t = torch.arange(1000, dtype=torch.float32, device=dev)
    ref = t.clone()
    comm.all_reduce(t).wait()
    comm.all_reduce(t, mean=True).wait()
    comm.broadcast(t, 0)
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    B, D, V = 32, 40, 57
    (x, y), = _toy_batches(dev, 1, B, D, V, 3)

    def run(reducer):
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g, reducer=reducer)
        for _ in range(2):
            tg.step(x, y)
        return g.params.clone()

    a = run(None)
    b = run(parallel.GradReducer(comm=comm))
    assert torch.equal(a, b)
    comm.close()


def _two_rank_worker(rank, world, port, q):
    try:
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import __graft_entry__
        __graft_entry__.load_package()
        import torch.distributed as dist
        import yt8m_amd.parallel as parallel
        import yt8m_amd.train as train_
        import yt8m_amd.video_level_models as vlm_
        from yt8m_amd.variables import reset_default_graph as rdg
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        parallel.init_from_env("nccl")
        dev = torch.device("cuda", rank)
        torch.cuda.set_device(dev)
        Bg, D, V = 64, 48, 65
        gen = torch.Generator(device="cpu").manual_seed(7)
        xs = [torch.rand((Bg, D), generator=gen) * 4 - 2 for _ in range(3)]
        ys = [torch.rand((Bg, V), generator=gen) < 0.05 for _ in range(3)]
        lo, hi = parallel.shard_batch(Bg, rank, world)
        g = rdg(device=dev, seed=0)
        tg = train_.TrainGraph(vlm_.MoeModel(), batch_size=Bg, graph=g, reducer=parallel.GradReducer())
        for x, y in zip(xs, ys):
            tg.step(x[lo:hi].to(dev), y[lo:hi].to(dev))
        out = g.params.detach().cpu()
        if rank == 0:
            g1 = rdg(device=dev, seed=0)
            t1 = train_.TrainGraph(vlm_.MoeModel(), batch_size=Bg, graph=g1)
            for x, y in zip(xs, ys):
                t1.step(x.to(dev), y.to(dev))
            ref = g1.params.detach().cpu()
            err = float((out - ref).abs().max() / ref.abs().max())
            q.put((rank, "ok" if err < 1e-5 else "FAIL rel err %g" % err))
        else:
            q.put((rank, "ok"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_two_rank_rccl_step_equals_one_rank_on_the_global_batch(dev):
    """SURVEY.md 8e on real RCCL: two ranks on shards of a global batch (bucketed all-reduce, 1/world folded into Adam, global-
    batch LR staircase) end at the 1-rank parameters on the concatenated batch.  Needs >= 2 GPUs (skipped on the 1-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---- no device allocation in a steady-state training step ---------------------------------------------------------------------
@pytest.mark.parametrize("model", ["LstmModel", "MoeModel"])
def test_steady_state_step_makes_no_device_allocation(dev, flags, model):
    """After the first steps every buffer of a training step comes from the caching allocator's pools and the recurrence scratch
    is the resident one (seq_ops._persist_ws): hipMalloc inside a step can wait on a driver fence for tens of ms and makes the step
    host-bound.  The allocator's device-allocation counter and its reserved bytes must not move over 6 more steps."""
    rs = np.random.RandomState(2)
    V = 23
    if model == "LstmModel":
        B, F, D = 32, 24, 64
        flags.lstm_cells, flags.lstm_pipeline_chunks = "512", 2          # persistent recurrence + x3 products engage
        x = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
        nf = torch.from_numpy(rs.randint(1, F + 1, size=B).astype(np.int32)).to(dev)
        mk = flm.LstmModel
    else:
        B, D = 256, 96
        x = torch.from_numpy(rs.randn(B, D).astype(np.float32)).to(dev)
        nf = None
        mk = vlm.MoeModel
    y = torch.from_numpy(rs.rand(B, V) < 0.2).to(dev)
    g = reset_default_graph(device=dev, seed=1)
    tg = train.TrainGraph(mk(), batch_size=B, graph=g)
    for _ in range(4):
        tg.step(x, y, nf)
    torch.cuda.synchronize()
    ms = torch.cuda.memory_stats()
    before = (ms.get("num_device_alloc", 0), ms["reserved_bytes.all.current"])
    for _ in range(6):
        tg.step(x, y, nf)
    torch.cuda.synchronize()
    ms = torch.cuda.memory_stats()
    assert (ms.get("num_device_alloc", 0), ms["reserved_bytes.all.current"]) == before


def test_vertical_and_frequent_supports_on_the_device(dev, flags, tmp_path):
    """support_type = "vertical,frequent,label" (W/losses.py:221-257) on device labels: labels . vm > 0.2 through the HIP GEMM with
    the table of --vertical_file, the first num_frequents classes, the labels themselves -- equal to the oracle bit for bit; and
    MultiTaskCrossEntropyLoss on those supports equals the oracle's loss."""
    import yt8m_amd.losses as losses
    from oracle import np_ref
    rs = np.random.RandomState(3)
    B, V, NV, NFQ = 19, 157, 7, 20
    lines = ["%d %d" % (c, rs.randint(NV)) for c in range(V) if c % 3 != 2] + ["", "1 2 3"]
    path = tmp_path / "vertical.tsv"
    path.write_text("\n".join(lines) + "\n")
    flags.vertical_file, flags.num_classes, flags.num_verticals, flags.num_frequents = str(path), V, NV, NFQ
    flags.support_type, flags.support_loss_percent = "vertical,frequent,label", 0.25
    y = rs.rand(B, V) < 0.05
    y[0] = False
    yd = torch.from_numpy(y).to(dev)
    vm = np_ref.load_vertical_mapping(lines, V, NV)
    want = np_ref.get_support_label_type(y, "vertical,frequent,label", num_frequents=NFQ, vertical_mapping=vm)
    got = losses.MultiTaskCrossEntropyLoss().get_support(yd)
    assert got.dtype == torch.float32 and tuple(got.shape) == (B, NV + NFQ + V)
    assert np.array_equal(got.cpu().numpy().astype(np.float64), want)
    p = rs.rand(B, V).astype(np.float32)
    sp = rs.rand(B, NV + NFQ + V).astype(np.float32)
    loss = losses.MultiTaskCrossEntropyLoss().calculate_loss(torch.from_numpy(p).to(dev), torch.from_numpy(sp).to(dev), yd)
    ref = np_ref.multitask_cross_entropy_loss(p.astype(np.float64), sp.astype(np.float64), y, want, 0.25)
    assert abs(float(loss) - ref) < 1e-5 * abs(ref)
