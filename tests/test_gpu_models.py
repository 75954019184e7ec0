"""-m gpu: plugin-level parity.  Each reference model class is driven through create_model() with injected
weights and compared with the oracle (forward <= 1e-3 on probabilities as north_star states -- in practice
~1e-6 -- and gradients against fp64 autograd); whole training steps (transform -> model -> loss -> clip -> Adam)
are compared step-for-step with the torch-CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref
import yt8m_amd.frame_level_models as flm
import yt8m_amd.losses as losses
import yt8m_amd.train as train
import yt8m_amd.video_level_models as vlm
from yt8m_amd.variables import reset_default_graph

pytestmark = pytest.mark.gpu
TOL_P = 1e-3     # north_star: "logits that match the TF1 reference within 1e-3 fp32"


def H(t):
    return t.detach().cpu().numpy().astype(np.float64)


def inject(g, params, dev):
    for k, v in params.items():
        assert k in g.vars, (k, list(g.vars))
        g.vars[k].data.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).to(dev).view(g.vars[k].data.shape))


def grads_of(g):
    return {k: H(v.grad) for k, v in g.vars.items() if v.trainable}


def randomise(g, rs, scale=0.3):
    P = {}
    for k, v in g.vars.items():
        if k.endswith("moving_variance") or k.endswith("gamma"):
            P[k] = (rs.rand(*v.shape) + 0.5).astype(np.float32)
        else:
            P[k] = (rs.randn(*v.shape) * scale).astype(np.float32)
    return P


def run_model(model, x, y, dev, nf=None, P=None, rs=None, multitask=False, **kw):
    """forward -> (optionally randomise / inject weights and forward again) -> loss -> backward."""
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(model, batch_size=x.shape[0], graph=g, transformer_class=__import__("yt8m_amd.feature_transform", fromlist=["x"]).IdenticalTransformer,
                          multitask=multitask, label_loss_fn=losses.MultiTaskCrossEntropyLoss() if multitask else None)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.from_numpy(y).to(dev)
    nfd = None if nf is None else torch.from_numpy(nf).to(dev)
    res = tg.forward(xd, yd, nfd)
    g.finalize()
    if P is None:
        P = randomise(g, rs)
    inject(g, P, dev)
    res = tg.forward(xd, yd, nfd)
    loss = tg.loss(res, yd)
    loss.backward()
    return g, res, loss, {k: v.astype(np.float64) for k, v in P.items()}


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def check_grads(g, tparams, tol=2e-4):
    got = grads_of(g)
    for k, t in tparams.items():
        if t.grad is None:
            continue
        ref = t.grad.numpy()
        err = np.abs(got[k] - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), (k, err, np.abs(ref).max())


def test_logistic_and_moe_models(dev, flags):
    rs = np.random.RandomState(0)
    B, Dm, V = 33, 70, 101
    x = rs.randn(B, Dm).astype(np.float32)
    y = rs.rand(B, V) < 0.05
    g, res, loss, P = run_model(vlm.LogisticModel(), x, y, dev, rs=rs)
    assert set(g.vars) == {"fully_connected/weights", "fully_connected/biases"}
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    pr = torch_ref.logistic(T(x), tp["fully_connected/weights"], tp["fully_connected/biases"])
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-5
    assert abs(float(loss) - lr.item()) < 1e-4 * abs(lr.item())
    check_grads(g, tp)
    for M in (1, 2, 4):
        flags.moe_num_mixtures = M
        g, res, loss, P = run_model(vlm.MoeModel(), x, y, dev, rs=rs)
        assert set(g.vars) == {"gates/weights", "experts/weights", "experts/biases"}
        assert g.vars["gates/weights"].shape == (Dm, V * (M + 1)) and g.vars["gates/weights"].l2 == 1e-8
        assert g.vars["experts/biases"].l2 == 0.0                     # biases are never regularised
        tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
        pr = torch_ref.moe(T(x), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], M)
        lr = torch_ref.cross_entropy(pr, T(y))
        lr.backward()
        assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-5
        assert np.abs(H(res["predictions"]) - np_ref.moe_model(x.astype(np.float64), P["gates/weights"], P["experts/weights"],
                                                                P["experts/biases"], M)).max() < 1e-5
        check_grads(g, tp)
    # sub_scope / num_mixtures kwargs and unknown kwargs are accepted (W/all_video_models/moe_model.py:12-19)
    g = reset_default_graph(device=dev)
    g.begin_step()
    out = vlm.MoeModel().create_model(torch.from_numpy(x).to(dev), vocab_size=V, num_mixtures=3, sub_scope="-x", foo=1, labels=None)
    assert "gates-x/weights" in g.vars and out["predictions"].shape == (B, V)


def test_deep_combine_chain_multitask(dev, flags):
    rs = np.random.RandomState(1)
    B, Dm, V, L, C = 9, 24, 31, 3, 8
    flags.deep_chain_layers, flags.deep_chain_relu_cells, flags.support_type = L, C, "label,label,label"
    flags.support_loss_percent = 0.05
    x = rs.randn(B, Dm).astype(np.float32)
    y = rs.rand(B, V) < 0.1
    g, res, loss, P = run_model(vlm.DeepCombineChainModel(), x, y, dev, rs=rs, multitask=True)
    assert "gates-prediction-0/weights" in g.vars and "relu-2/biases" in g.vars and "experts--main/biases" in g.vars
    assert g.vars["gates--main/weights"].shape == (Dm + L * C, V * 3)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.deep_combine_chain(T(x), tp, L, 2)
    ysup = T(np.tile(y, (1, L)))
    lr = 0.95 * torch_ref.cross_entropy(main, T(y)) + 0.05 * torch_ref.cross_entropy(sup, ysup)
    lr.backward()
    assert np.abs(H(res["predictions"]) - main.detach().numpy()).max() < 1e-5
    assert np.abs(H(res["support_predictions"]) - sup.detach().numpy()).max() < 1e-5
    assert abs(float(loss) - lr.item()) < 1e-4 * abs(lr.item())
    check_grads(g, tp)


def _lstm_ref_layers(tp, L):
    return [(tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
            for l in range(L)]


@pytest.mark.parametrize("chunks", [1, 4, 10])
@pytest.mark.parametrize("cls", ["LstmModel", "LstmMemoryModel"])
def test_lstm_models(dev, flags, cls, chunks, honour_lstm_chunks):
    rs = np.random.RandomState(2)
    B, F, Dm, Hh, V = 6, 10, 12, 8, 17
    flags.lstm_cells, flags.lstm_layers = str(Hh), 2
    flags.lstm_pipeline_chunks = chunks                               # layer pipeline over time chunks (1 = sequential)
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([10, 1, 5, 10, 3, 7], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(getattr(flm, cls)(), x, y, dev, nf=nf, rs=rs)
    assert g.vars["gates/weights"].shape[0] == (4 * Hh if cls == "LstmModel" else 2 * Hh)      # [c0|h0|c1|h1] vs [c0|c1]
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    layers = _lstm_ref_layers(tp, 2)
    if cls == "LstmModel":
        st = torch_ref.lstm_model_state(T(x), torch.from_numpy(nf), layers)
    else:
        _, c, _ = torch_ref.lstm_stack(T(x), torch.from_numpy(nf), layers)
        st = torch.cat(c, 1)
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)


def test_lstm_stack_pipelined_equals_sequential(dev, flags, honour_lstm_chunks, monkeypatch):
    """The layer-pipelined stack (time chunks on separate streams, fused step kernels: H % 128 == 0) against the same op
    with one chunk: forward results are bit-identical (same kernels, same per-step arithmetic); weight gradients differ only
    by the chunk-wise accumulation order of the hoisted dW GEMMs.  Repeated to catch stream races.  The persistent kernels are
    switched off: the subject is the per-step path and its hipGraph replay (H = 128 has a persistent form since round 4)."""
    import yt8m_amd.seq_ops as seq_ops
    monkeypatch.setattr(seq_ops, "PERSIST", False)
    from yt8m_amd.variables import xavier_uniform, zeros
    rs = np.random.RandomState(23)
    F, B, Dm, Hh = 37, 48, 64, 128
    x = torch.from_numpy(rs.randn(F, B, Dm).astype(np.float32)).to(dev)
    nf = torch.from_numpy(rs.randint(1, F + 1, size=B).astype(np.int32)).to(dev)
    dout = torch.from_numpy(rs.randn(F, B, Hh).astype(np.float32)).to(dev)
    res = {}
    for chunks in (1, 5, 5, 3):
        g = reset_default_graph(device=dev, seed=4)
        g.begin_step()
        wb = []
        for l, din in enumerate((Dm, Hh, Hh)):
            wb.append((g.get_variable("l%d/weights" % l, (din + Hh, 4 * Hh), xavier_uniform),
                       g.get_variable("l%d/biases" % l, (4 * Hh,), zeros)))
        g.finalize()
        xin = x.clone().requires_grad_(True)
        out, fin = seq_ops.lstm_stack(xin, nf, wb, chunks=chunks)
        loss = (out * dout).sum() + sum((c * 0.3).sum() + (h * 0.7).sum() for c, h in fin)
        loss.backward()
        torch.cuda.synchronize()
        cur = (out.detach().clone(), [t.detach().clone() for ch in fin for t in ch], xin.grad.clone(),
               {k: v.grad.clone() for k, v in g.vars.items()})
        if chunks == 1:
            res = cur
            continue
        assert torch.equal(cur[0], res[0])
        for a, b in zip(cur[1], res[1]):
            assert torch.equal(a, b)
        assert float((cur[2] - res[2]).abs().max()) <= 1e-6 * float(res[2].abs().max())
        for k in res[3]:
            assert float((cur[3][k] - res[3][k]).abs().max()) <= 2e-6 * max(float(res[3][k].abs().max()), 1e-6), k
    # the step chains went through the hipGraph replay cache (capture on first use, replay when the arguments repeat)
    import ctypes
    import yt8m_amd._lib as L
    h, c, f, e = (ctypes.c_int64(0) for _ in range(4))
    L.check(L.lib().yt8m_graph_cache_stats(ctypes.byref(h), ctypes.byref(c), ctypes.byref(f), ctypes.byref(e)))
    assert c.value > 0 and f.value == 0 and e.value <= 96
    L.check(L.lib().yt8m_graph_cache_clear())
    L.check(L.lib().yt8m_graph_cache_stats(None, None, None, ctypes.byref(e)))
    assert e.value == 0


def test_lstm_parallel_finaloutput_model(dev, flags):
    """parallel rgb / audio LSTM stacks (W/all_frame_models/lstm_parallel_finaloutput_model.py)."""
    rs = np.random.RandomState(6)
    B, F, V = 5, 9, 13
    fsz, hsz = [10, 6], [8, 4]
    flags.feature_sizes, flags.lstm_cells, flags.lstm_layers = "10,6", "8,4", 2
    nf = np.array([9, 1, 4, 9, 6], dtype=np.int32)
    x = rs.randn(B, F, sum(fsz)).astype(np.float32) * (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.2
    g, res, loss, P = run_model(flm.LstmParallelFinaloutputModel(), x, y, dev, nf=nf, rs=rs)
    assert "RNN1/multi_rnn_cell/cell_1/basic_lstm_cell/weights" in g.vars
    assert g.vars["gates/weights"].shape[0] == 2 * (8 + 4)                                  # h of 2 layers x 2 stacks
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    sets = [[(tp["RNN%d/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % (i, l)],
              tp["RNN%d/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % (i, l)]) for l in range(2)] for i in range(2)]
    st = torch_ref.lstm_parallel_finaloutput(T(x), torch.from_numpy(nf), sets, fsz)
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)
    flags.lstm_cells = "8"
    with pytest.raises(AssertionError):                                                       # reference assert (:39-41)
        run_model(flm.LstmParallelFinaloutputModel(), x, y, dev, nf=nf, rs=rs)


def test_lstm_positional_attention_max_pooling_model(dev, flags):
    """positional attention (W/all_frame_models/lstm_positional_attention_max_pooling_model.py)."""
    rs = np.random.RandomState(7)
    B, F, Dm, Hh, V, A, E = 4, 7, 10, 8, 11, 3, 5
    flags.lstm_cells, flags.lstm_layers, flags.lstm_attentions, flags.positional_embedding_size = str(Hh), 2, A, E
    nf = np.array([7, 2, 5, 7], dtype=np.int32)
    x = rs.randn(B, F, Dm).astype(np.float32) * (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.2
    g, res, loss, P = run_model(flm.LstmPositionalAttentionMaxPoolingModel(), x, y, dev, nf=nf, rs=rs)
    assert g.vars["positional_embedding"].shape == (1, F, E) and g.vars["positional_embedding"].l2 == 1e-8
    assert g.vars["attention-/weights"].shape == (Dm + E + Dm + Hh, A)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    pr = torch_ref.lstm_positional_attention_max_pooling(
        T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2), tp["positional_embedding"], tp["attention-/weights"],
        tp["attention-/biases"], tp["gates-sub-moe/weights"], tp["experts-sub-moe/weights"], tp["experts-sub-moe/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)


def test_cnn_deep_combine_chain_model(dev, flags):
    """einsum-CNN chain (W/all_frame_models/cnn_deep_combine_chain_model.py) under the multitask loss."""
    rs = np.random.RandomState(8)
    B, F, Dm, V, L, cells = 4, 6, 9, 12, 2, 5
    flags.deep_chain_layers, flags.deep_chain_relu_cells = L, cells
    flags.support_type = ",".join(["label"] * L)
    nf = np.array([6, 1, 4, 3], dtype=np.int32)
    x = rs.randn(B, F, Dm).astype(np.float32) * (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.2
    g, res, loss, P = run_model(flm.CnnDeepCombineChainModel(), x, y, dev, nf=nf, rs=rs, multitask=True)
    assert g.vars["cnn1cnn-filter-len3"].shape == (3 * Dm, 2 * cells) and g.vars["cnn0cnn-filter-len2"].l2 == 1e-8
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.cnn_deep_combine_chain(T(x), torch.from_numpy(nf), tp, L, 2, cells)
    assert np.abs(H(res["predictions"]) - main.detach().numpy()).max() < 1e-4
    assert np.abs(H(res["support_predictions"]) - sup.detach().numpy()).max() < 1e-4
    s = flags.support_loss_percent
    lr = (1 - s) * torch_ref.cross_entropy(main, T(y)) + s * torch_ref.cross_entropy(sup, T(y).repeat(1, L))
    lr.backward()
    assert abs(float(loss.detach()) - lr.item()) < 1e-4 * abs(lr.item())
    check_grads(g, tp, tol=5e-4)


@pytest.mark.parametrize("skinny", [False, True])
def test_lstm_attention_max_pooling_model(dev, flags, skinny, monkeypatch):
    rs = np.random.RandomState(3)
    B, F, Dm, Hh, V, A = 5, 9, 10, 6, 13, 3
    if skinny:                                   # attention FC through the streaming kernels (widths must be multiples of 4)
        import yt8m_amd.ops as ops
        monkeypatch.setattr(ops, "SKINNY_MIN_ROWS", 1)
        Dm, Hh = 12, 8
    flags.lstm_cells, flags.lstm_layers, flags.lstm_attentions = str(Hh), 2, A
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([9, 1, 4, 9, 2], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.2
    g, res, loss, P = run_model(flm.LstmAttentionMaxPoolingModel(), x, y, dev, nf=nf, rs=rs)
    assert {"attention-/weights", "attention-/biases", "gates-sub-moe/weights", "experts-sub-moe/biases"} <= set(g.vars)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    pr = torch_ref.lstm_attention_max_pooling(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2), tp["attention-/weights"],
                                              tp["attention-/biases"], tp["gates-sub-moe/weights"], tp["experts-sub-moe/weights"],
                                              tp["experts-sub-moe/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)


@pytest.mark.parametrize("gated", [False, True])
def test_netvlad_models(dev, flags, gated):
    rs = np.random.RandomState(4)
    B, F, Dm, K, Hf, V = 5, 8, 16, 4, 12, 11
    flags.netvlad_cluster_size, flags.netvlad_hidden_size = K, Hf
    nf = np.array([8, 1, 3, 8, 5], dtype=np.int32)
    x = np_ref.l2_normalize(rs.randn(B, F, Dm)) * (np.arange(F)[None, :, None] < nf[:, None, None])
    x = x.astype(np.float32)
    y = rs.rand(B, V) < 0.2
    g, res, loss, P = run_model((flm.GatedNetVLADModel if gated else flm.NetVLADModel)(), x, y, dev, nf=nf, rs=rs)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    h = torch_ref.netvlad_hidden(T(x), torch.from_numpy(nf), tp["netvlad/cluster_weights"], tp["netvlad/cluster_biases"],
                                 tp["netvlad/centres"], tp["netvlad/hidden/weights"], tp["netvlad/hidden/biases"],
                                 tp["netvlad/gating/weights"] if gated else None, tp["netvlad/gating/biases"] if gated else None)
    pr = torch_ref.moe(h, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)


def test_dbof_and_frame_logistic(dev, flags):
    rs = np.random.RandomState(5)
    B, F, Dm, V = 6, 12, 10, 9
    flags.dbof_cluster_size, flags.dbof_hidden_size, flags.iterations, flags.dbof_add_batch_norm = 16, 8, 5, False
    nf = np.array([12, 1, 6, 12, 3, 9], dtype=np.int32)
    x = (rs.randn(B, F, Dm) * (np.arange(F)[None, :, None] < nf[:, None, None])).astype(np.float32)
    y = rs.rand(B, V) < 0.2
    # frame-level logistic: average over num_frames (logistic_model.py:35-41)
    g, res, loss, P = run_model(flm.FrameLevelLogisticModel(), x, y, dev, nf=nf, rs=rs)
    avg = x.astype(np.float64).sum(1) / nf[:, None]
    pr = np_ref.logistic_model(avg, P["fully_connected/weights"], P["fully_connected/biases"])
    assert np.abs(H(res["predictions"]) - pr).max() < 1e-5
    # DBoF: sampling is random, so check the deterministic remainder by feeding iterations == all frames of
    # constant-per-video inputs (any sampled frame is the same row)
    xc = np.repeat(rs.randn(B, 1, Dm), F, axis=1).astype(np.float32)
    g, res, loss, P = run_model(flm.DbofModel(), xc, y, dev, nf=nf, rs=rs)
    assert [k for k in g.vars][:4] == ["Variable", "Variable_1", "Variable_2", "Variable_3"]
    hd = np_ref.dbof_model_hidden(xc[:, :5].astype(np.float64), P["Variable"], P["Variable_1"], P["Variable_2"], P["Variable_3"])
    pr = np_ref.moe_model(hd, P["gates/weights"], P["experts/weights"], P["experts/biases"], 2)
    assert np.abs(H(res["predictions"]) - pr).max() < 1e-4
    flags.dbof_add_batch_norm = True
    g, res, loss, P = run_model(flm.DbofModel(), xc, y, dev, nf=nf, rs=rs)
    assert "cluster_bn/gamma" in g.vars and not g.vars["cluster_bn/moving_mean"].trainable
    assert torch.isfinite(res["predictions"]).all()


def test_training_steps_match_cpu_restatement(dev, flags):
    """BASELINE config[1] shape family (MoeModel M=2 on L2-normalised video-level features): 5 optimiser steps on
    the GPU vs the torch-CPU restatement in fp64, same init, same batches -- loss, predictions and weights."""
    rs = np.random.RandomState(6)
    B, Dm, V, M = 48, 160, 311, 2
    g = reset_default_graph(device=dev, seed=5)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g, base_learning_rate=0.01)
    cpu = torch_ref.MoeTrainStepCPU(D=Dm, V=V, M=M, batch_size=B, dtype=torch.float64, base_lr=0.01)
    for step in range(5):
        x = (rs.randn(B, Dm) * 1.5).astype(np.float32)
        y = rs.rand(B, V) < 0.03
        if step == 0:
            tg.forward(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
            g.finalize()
            inject(g, {k: v.detach().numpy() for k, v in cpu.P.items()}, dev)
        out = tg.step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
        lc, pc = cpu.step(T(x), torch.from_numpy(y))
        assert abs(float(out["loss"]) - lc.item()) < 1e-4 * abs(lc.item()), step
        assert np.abs(H(out["predictions"]) - pc.numpy()).max() < TOL_P * 0.1
        for k, v in cpu.P.items():
            assert np.abs(H(g.vars[k].data) - v.detach().numpy()).max() < 2e-5, (step, k)
    assert tg.global_step == 5
    # reg loss report: sum l2 * 0.5 * |W|^2 over regularised weights (W/train.py:435-445)
    ref = sum(1e-8 * 0.5 * float((cpu.P[k].detach() ** 2).sum()) for k in ("gates/weights", "experts/weights"))
    assert tg.regularization_loss() == pytest.approx(ref, rel=1e-4)


def test_uint8_input_path_and_eval(dev, flags):
    """Frame-level uint8 batch through the DefaultTransformer (dequantise + pad + L2-normalise fused) into LstmModel,
    then GAP through the device top-k path vs the host metric on the same predictions."""
    import yt8m_amd.eval_util as eu
    rs = np.random.RandomState(7)
    B, F, Dm, V = 8, 6, 32, 40
    flags.lstm_cells, flags.lstm_layers = "8", 1
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    y = rs.rand(B, V) < 0.1
    y[:, 0] |= y.sum(1) == 0
    g = reset_default_graph(device=dev, seed=1)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    out = tg.step(torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev))
    P = {k: H(v.data) for k, v in g.vars.items()}
    assert torch.isfinite(out["loss"])
    p = tg.predict(torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev), vocab_size=V)
    xr = np_ref.dequant_l2norm_folded(q, nf)
    st = np_ref.lstm_model_state(xr, nf, [(P["RNN/multi_rnn_cell/cell_0/basic_lstm_cell/weights"], P["RNN/multi_rnn_cell/cell_0/basic_lstm_cell/biases"])])
    pr = np_ref.moe_model(st, P["gates/weights"], P["experts/weights"], P["experts/biases"], 2)
    assert np.abs(H(p) - pr).max() < 1e-4
    em = eu.EvaluationMetrics(V, 20)
    em.accumulate_device(p, torch.from_numpy(y).to(dev), 1.0)
    assert em.get()["gap"] == pytest.approx(eu.calculate_gap(H(p), y.astype(np.float64), 20), abs=1e-9)
    assert em.get()["avg_hit_at_one"] == pytest.approx(eu.calculate_hit_at_one(H(p), y.astype(np.float64)))


def test_data_parallel_step_single_rank_rccl(dev, flags):
    """The data-parallel step (RCCL all-reduce per bucket as gradients finish, per-bucket clip+Adam, un-grouped dW
    GEMMs) on a 1-rank `nccl` group must reproduce the plain single-process step: same loss, same weights."""
    import os
    import socket
    import torch.distributed as dist
    import yt8m_amd.parallel as parallel
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    created = False
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        created = True
    try:
        rs = np.random.RandomState(8)
        B, Dm, V = 256, 384, 1000
        xs = [(rs.randn(B, Dm)).astype(np.float32) for _ in range(3)]
        ys = [rs.rand(B, V) < 0.01 for _ in range(3)]
        outs = []
        for use_dp in (False, True):
            g = reset_default_graph(device=dev, seed=11)
            red = parallel.GradReducer(bucket_bytes=1 << 20) if use_dp else None
            tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g, reducer=red)
            ls = [float(tg.step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))["loss"]) for x, y in zip(xs, ys)]
            outs.append((ls, {k: v.data.clone() for k, v in g.vars.items()}))
            if use_dp:
                assert red.active and red.gscale == 1.0
        assert outs[0][0] == pytest.approx(outs[1][0], rel=1e-6)
        for k in outs[0][1]:
            assert float((outs[0][1][k] - outs[1][1][k]).abs().max()) < 1e-6, k
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("label_kind", ["bool", "f32"])
def test_fused_head_loss_equals_unfused(dev, flags, label_kind):
    """MoeModel's fused mixing + CrossEntropyLoss path (returns its own "loss", W/train.py:384-385) against the
    unfused model -> losses.CrossEntropyLoss path and the fp64 oracle: loss, predictions and every gradient."""
    rs = np.random.RandomState(12)
    B, Dm, V, M = 37, 50, 301, 2
    x = rs.randn(B, Dm).astype(np.float32)
    y = rs.rand(B, V) < 0.03
    ylab = y if label_kind == "bool" else (y * 0.9 + 0.01).astype(np.float32)        # soft labels take the f32 kernel
    res = {}
    for fused in (True, False):
        flags.fused_head_loss = fused
        g, r, loss, P = run_model(vlm.MoeModel(), x, ylab, dev, rs=np.random.RandomState(5))
        assert ("loss" in r) == fused
        res[fused] = (float(loss), H(r["predictions"]), grads_of(g), P)
    assert res[True][0] == pytest.approx(res[False][0], rel=1e-6)
    assert np.abs(res[True][1] - res[False][1]).max() < 1e-7
    for k in res[True][2]:
        assert np.abs(res[True][2][k] - res[False][2][k]).max() <= 1e-6 * max(1.0, np.abs(res[False][2][k]).max()), k
    P = res[True][3]
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    pr = torch_ref.moe(T(x), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], M)
    lr = torch_ref.cross_entropy(pr, T(np.asarray(ylab, dtype=np.float64)))
    lr.backward()
    assert abs(res[True][0] - lr.item()) < 1e-4 * abs(lr.item())
    for k, t in tp.items():
        assert np.abs(res[True][2][k] - t.grad.numpy()).max() <= 2e-4 * max(1.0, np.abs(t.grad.numpy()).max()), k
    # per-example weights are a loss-side feature: the trainer turns the fusion off for that call
    g = reset_default_graph(device=dev, seed=0)
    flags.fused_head_loss = True
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
    w = torch.rand(B, device=dev)
    out = tg.step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), weights=w)
    assert torch.isfinite(out["loss"])


@pytest.mark.parametrize("gated", [False, True])
def test_netvlad_u8_fused_equals_generic(dev, flags, gated):
    """NetVLAD plugin fed the reader's raw uint8 frames: the fused path (dequantise + l2-normalise folded into the
    pooling GEMMs) against (i) the generic fp32 path on the transformed input and (ii) the fp64 oracle -- predictions,
    loss and every gradient.  K = 64 clusters as the fused kernels require."""
    rs = np.random.RandomState(14)
    B, F, Dm, K, Hf, V = 6, 40, 128, 64, 24, 31
    flags.netvlad_cluster_size, flags.netvlad_hidden_size = K, Hf
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    nf = np.array([40, 1, 17, 40, 33, 8], dtype=np.int32)
    y = rs.rand(B, V) < 0.15
    model = flm.GatedNetVLADModel if gated else flm.NetVLADModel
    out = {}
    for fold in (True, False):
        flags.fold_dequant = fold
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(model(), batch_size=B, graph=g)
        qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
        tg.forward(qd, yd, nfd)
        g.finalize()
        if fold:
            P = randomise(g, np.random.RandomState(3), scale=0.4)
        inject(g, P, dev)
        res = tg.forward(qd, yd, nfd)
        loss = tg.loss(res, yd)
        loss.backward()
        out[fold] = (H(res["predictions"]), float(loss), grads_of(g))
    assert np.abs(out[True][0] - out[False][0]).max() < 2e-5
    assert out[True][1] == pytest.approx(out[False][1], rel=1e-5)
    for k in out[True][2]:
        ref = out[False][2][k]
        assert np.abs(out[True][2][k] - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-7), k
    x = np_ref.dequant_l2norm_folded(q, nf)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    h = torch_ref.netvlad_hidden(T(x), torch.from_numpy(nf), tp["netvlad/cluster_weights"], tp["netvlad/cluster_biases"],
                                 tp["netvlad/centres"], tp["netvlad/hidden/weights"], tp["netvlad/hidden/biases"],
                                 tp["netvlad/gating/weights"] if gated else None, tp["netvlad/gating/biases"] if gated else None)
    pr = torch_ref.moe(h, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(out[True][0] - pr.detach().numpy()).max() < 1e-4
    for k, t in tp.items():
        ref = t.grad.numpy()
        assert np.abs(out[True][2][k] - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("quantized", [False, True])
def test_config5_composite_model(dev, flags, quantized):
    """GatedNetVLADAttentionChainModel (BASELINE configs[4] composite) under the multitask loss
    against the oracle: predictions, support predictions, loss and every gradient; float input and raw uint8 input."""
    rs = np.random.RandomState(17)
    B, F, Dm, K, Hf, V, A, L = 4, 10, 64, 64, 16, 13, 3, 2
    flags.netvlad_cluster_size, flags.netvlad_hidden_size, flags.lstm_attentions = K, Hf, A
    flags.deep_chain_layers, flags.deep_chain_relu_cells = L, 8
    flags.support_type = ",".join(["label"] * L)
    flags.support_loss_percent = 0.1
    nf = np.array([10, 1, 4, 7], dtype=np.int32)
    y = rs.rand(B, V) < 0.2
    if quantized:
        q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
        x64 = np_ref.dequant_l2norm_folded(q, nf)
        inp = q
    else:
        x64 = np_ref.l2_normalize(rs.randn(B, F, Dm)) * (np.arange(F)[None, :, None] < nf[:, None, None])
        inp = x64.astype(np.float32)
        x64 = inp.astype(np.float64)
    g = reset_default_graph(device=dev, seed=0)
    tcls = __import__("yt8m_amd.feature_transform", fromlist=["x"])
    tg = train.TrainGraph(flm.GatedNetVLADAttentionChainModel(), batch_size=B, graph=g, multitask=True,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss(),
                          transformer_class=tcls.DefaultTransformer if quantized else tcls.IdenticalTransformer)
    xd, yd, nfd = torch.from_numpy(inp).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    tg.forward(xd, yd, nfd)
    g.finalize()
    P = randomise(g, rs, scale=0.4)
    inject(g, P, dev)
    res = tg.forward(xd, yd, nfd)
    loss = tg.loss(res, yd)
    loss.backward()
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.gated_netvlad_attention_chain(T(x64), torch.from_numpy(nf), tp, L, 2, A)
    assert np.abs(H(res["predictions"]) - main.detach().numpy()).max() < 1e-4
    assert np.abs(H(res["support_predictions"]) - sup.detach().numpy()).max() < 1e-4
    yt = T(y)
    lr = 0.9 * torch_ref.cross_entropy(main, yt) + 0.1 * torch_ref.cross_entropy(sup, yt.repeat(1, L))
    lr.backward()
    assert abs(float(loss.detach()) - lr.item()) < 1e-4 * abs(lr.item())
    check_grads(g, tp, tol=5e-4)


@pytest.mark.parametrize("mode", ["type1", "type2", "boosting", "type1_multitask"])
def test_distillation_losses(dev, flags, mode):
    """SURVEY.md 8f item 3, W/train.py:312-327,384-430: distillation labels blended into / replacing the label loss, re-formed
    labels (type 2) and boosting weights from the teacher's cross entropy -- loss and gradients against the oracle."""
    rs = np.random.RandomState(19)
    B, Dm, V = 9, 14, 23
    x = rs.randn(B, Dm).astype(np.float32)
    y = rs.rand(B, V) < 0.15
    teacher = rs.rand(B, V).astype(np.float32) * 0.9 + 0.05
    flags.distillation_features = True
    flags.distillation_percent = 0.3
    multitask = mode == "type1_multitask"
    if mode.startswith("type1"):
        flags.distillation_type = 1
    elif mode == "type2":
        flags.distillation_type = 2
    else:
        flags.distillation_type = 0
        flags.distillation_as_boosting = True
    if multitask:
        flags.deep_chain_layers, flags.deep_chain_relu_cells, flags.support_type = 2, 4, "label,label"
    model = vlm.DeepCombineChainModel() if multitask else vlm.MoeModel()
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(model, batch_size=B, graph=g, multitask=multitask,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss() if multitask else None,
                          transformer_class=__import__("yt8m_amd.feature_transform", fromlist=["x"]).IdenticalTransformer)
    xd, yd, td = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(teacher).to(dev)
    tg.forward(xd, yd)
    g.finalize()
    P = randomise(g, rs)
    inject(g, P, dev)
    before = {k: v.data.clone() for k, v in g.vars.items()}
    out = tg.step(xd, yd, distill_labels_batch=td)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    yt, tt = T(y), T(teacher)
    if multitask:
        main, sup = torch_ref.deep_combine_chain(T(x), tp, 2, 2)

        def L(lab, w=None):
            s = flags.support_loss_percent
            return (1 - s) * torch_ref.cross_entropy(main, lab, w) + s * torch_ref.cross_entropy(sup, lab.repeat(1, 2), w)
    else:
        pr = torch_ref.moe(T(x), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)

        def L(lab, w=None):
            return torch_ref.cross_entropy(pr, lab, w)
    if mode.startswith("type1"):
        lr = 0.7 * L(yt) + 0.3 * L(tt)
    elif mode == "type2":
        lr = L(torch_ref.reform_distill_labels(yt, tt, 0.3))
    else:
        lr = L(yt, torch_ref.weights_by_predictions(yt, tt))
    lr.backward()
    assert abs(float(out["loss"]) - lr.item()) < 1e-4 * abs(lr.item())
    # one Adam step from the same start: compare the parameter update direction through the gradient the step used
    got = grads_of(g)
    for k, t in tp.items():
        if t.grad is not None:
            ref = t.grad.numpy()
            assert np.abs(got[k] - ref).max() <= 5e-4 * max(1.0, np.abs(ref).max()), k
    assert any(float((g.vars[k].data - before[k]).abs().max()) > 0 for k in before)


def test_gap_on_heldout_shard_matches_cpu_training(dev, flags):
    """North-star acceptance check, at a size the CPU oracle trains in seconds: MoeModel trained on a synthetic
    teacher shard by the HIP path and by the torch-CPU restatement from the SAME initial weights on the SAME batches; GAP@20 on
    a disjoint held-out shard must agree within 0.001 (and be far above the untrained model's)."""
    import yt8m_amd.eval_util as eu
    D_, V_, M_, B_, steps, held = 64, 300, 2, 256, 60, 2048
    gen = torch.Generator().manual_seed(7)
    Wt = torch.randn(D_, V_, generator=gen) / D_ ** 0.5

    def shard(n, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(n, D_, generator=g) * 4.0 - 2.0
        logit = x @ Wt * 3.0 - 3.0 + 0.5 * torch.randn(n, V_, generator=g)
        tau = torch.quantile(logit.flatten()[:200000], 1.0 - 3.4 / V_)        # ~3.4 positives per video
        return x, logit > tau
    xtr, ytr = shard(B_ * steps, 11)
    xho, yho = shard(held, 12)                                               # disjoint seed = held-out shard
    assert 2.0 < float(yho.float().sum(1).mean()) < 5.0
    cpu = torch_ref.MoeTrainStepCPU(D=D_, V=V_, M=M_, batch_size=B_, dtype=torch.float32, seed=3)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B_, graph=g)
    tg.forward(xtr[:B_].to(dev), ytr[:B_].to(dev))
    g.finalize()
    inject(g, {k: v.detach().numpy() for k, v in cpu.P.items()}, dev)
    p0 = tg.predict(xho.to(dev), vocab_size=V_)
    gap0 = eu.calculate_gap(p0.cpu().numpy(), yho.numpy().astype(np.float32), 20)
    for i in range(steps):
        xb, yb = xtr[i * B_:(i + 1) * B_], ytr[i * B_:(i + 1) * B_]
        lh = tg.step(xb.to(dev), yb.to(dev))["loss"]
        lc, _ = cpu.step(xb, yb)
        assert abs(float(lh) - float(lc)) < 2e-3 * abs(float(lc)), (i, float(lh), float(lc))
    ph = tg.predict(xho.to(dev), vocab_size=V_).cpu()
    with torch.no_grad():
        pc = torch_ref.moe(torch_ref.l2_normalize(xho, 1), cpu.P["gates/weights"], cpu.P["experts/weights"], cpu.P["experts/biases"], M_)
    gh = eu.calculate_gap(ph.numpy(), yho.numpy().astype(np.float32), 20)
    gc = eu.calculate_gap(pc.numpy(), yho.numpy().astype(np.float32), 20)
    assert abs(gh - gc) < 1e-3, (gh, gc)
    assert gh > gap0 + 0.05 and gh > 0.2, (gap0, gh)
    # the device metric path agrees with the host one on the same predictions
    em = eu.EvaluationMetrics(V_, 20)
    em.accumulate_device(ph.to(dev), yho.to(dev), 0.0)
    assert abs(em.get()["gap"] - gh) < 1e-3


def _bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("fused", [True, False])
def test_moe_head_bf16(dev, flags, fused, monkeypatch):
    """compute_dtype=bfloat16 (BASELINE config 5): the head's GEMMs take bf16 operands with fp32 accumulation.  Forward is
    checked two ways: against the fp64 oracle evaluated on the bf16-rounded operands (tight: only accumulation order
    differs) and against the unrounded oracle at north_star's 1e-3; gradients against fp64 autograd at bf16 resolution
    (operands of dW = x^T dZ carry 2^-9 relative rounding each).  Master weights / loss / optimiser stay fp32."""
    import yt8m_amd.ops as ops
    monkeypatch.setattr(ops, "BF16_MIN_ROWS", 2)                     # the production threshold (512 rows) is a cost model
    rs = np.random.RandomState(31)
    B, Dm, V, M = 64, 96, 250, 2
    x = rs.randn(B, Dm).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)                    # the path's inputs are L2-normalised
    y = rs.rand(B, V) < 0.03
    flags.fused_head_loss = fused
    flags.compute_dtype = "bfloat16"
    g, r, loss, P = run_model(vlm.MoeModel(), x, y, dev, rs=np.random.RandomState(5))
    assert ("loss" in r) == fused
    got_p, got_g = H(r["predictions"]), grads_of(g)
    # (a) oracle on rounded operands
    pr_r = torch_ref.moe(_bf16_round(x), _bf16_round(P["gates/weights"]), _bf16_round(P["experts/weights"]),
                         T(P["experts/biases"]), M)
    assert np.abs(got_p - pr_r.numpy()).max() < 2e-6
    # (b) unrounded oracle, north_star tolerance; loss and gradients
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    pr = torch_ref.moe(T(x), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], M)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(got_p - pr.detach().numpy()).max() < TOL_P
    assert abs(float(loss) - lr.item()) < 2e-3 * abs(lr.item())
    for k, t in tp.items():
        ref = t.grad.numpy()
        assert np.abs(got_g[k] - ref).max() <= 1e-2 * np.abs(ref).max(), (k, np.abs(got_g[k] - ref).max(), np.abs(ref).max())
    # fp32 path on the same weights: the bf16 result is a perturbation of it, not a different function
    flags.compute_dtype = "float32"
    g2, r2, loss2, _ = run_model(vlm.MoeModel(), x, y, dev, P={k: v.astype(np.float32) for k, v in P.items()})
    assert np.abs(H(r2["predictions"]) - pr.detach().numpy()).max() < 1e-5
    assert 0 < np.abs(H(r2["predictions"]) - got_p).max() < TOL_P
    # odd reduction lengths cannot be packed in bf16 pairs: the op falls back to the exact fp32 MFMA path, loudly documented
    flags.compute_dtype = "bfloat16"
    xo = rs.randn(7, 33).astype(np.float32)
    yo = rs.rand(7, 11) < 0.2
    g3, r3, _, P3 = run_model(vlm.MoeModel(), xo, yo, dev, rs=np.random.RandomState(6))
    po = torch_ref.moe(T(xo), T(P3["gates/weights"]), T(P3["experts/weights"]), T(P3["experts/biases"]), M)
    assert np.abs(H(r3["predictions"]) - po.numpy()).max() < 1e-5


def test_bf16_training_tracks_fp32(dev, flags, monkeypatch):
    """Ten steps of the config-1 shape at reduced size: the bf16-operand path's loss curve stays within 1e-3 relative of
    the fp32 path's (fp32 master weights + fp32 Adam, so rounding does not accumulate in the parameters)."""
    import yt8m_amd.ops as ops
    monkeypatch.setattr(ops, "BF16_MIN_ROWS", 2)
    rs = np.random.RandomState(41)
    B, Dm, V = 128, 128, 400
    xs = [rs.randn(B, Dm).astype(np.float32) for _ in range(10)]
    ys = [rs.rand(B, V) < 0.02 for _ in range(10)]
    curves = {}
    for dt in ("float32", "bfloat16"):
        flags.compute_dtype = dt
        g = reset_default_graph(device=dev, seed=3)
        tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
        curves[dt] = [float(tg.step(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))["loss"]) for x, y in zip(xs, ys)]
    a, b = np.array(curves["float32"]), np.array(curves["bfloat16"])
    assert a[-1] < a[0]
    assert np.abs(a - b).max() <= 1e-3 * np.abs(a).max(), (a, b)


def _small_lstm_params(rs, Dm, Hh, V):
    """xavier-scale weights (the randomise() default of 0.3 saturates a 256-wide cell and amplifies operand rounding)"""
    P = {}
    d_in = Dm
    for l in range(2):
        P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l] = (rs.randn(d_in + Hh, 4 * Hh) / np.sqrt(d_in + Hh)).astype(np.float32)
        P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l] = (rs.randn(4 * Hh) * 0.1).astype(np.float32)
        d_in = Hh
    P["gates/weights"] = (rs.randn(4 * Hh, V * 3) * 0.05).astype(np.float32)
    P["experts/weights"] = (rs.randn(4 * Hh, V * 2) * 0.05).astype(np.float32)
    P["experts/biases"] = (rs.randn(V * 2) * 0.1).astype(np.float32)
    return P


@pytest.mark.parametrize("chunks,Hh", [(1, 128), (3, 128), (1, 256), (3, 256)])
def test_lstm_model_bf16_projections(dev, flags, chunks, Hh, monkeypatch, honour_lstm_chunks):
    """--compute_dtype=bfloat16 on LstmModel: the hoisted products of the stack (input projection, dW, dx) take bf16 operands,
    and -- for H % 256 == 0 -- the recurrent product too (csrc/lstm_bf16.hip: bf16 h / dz / W_h operands, fp32 state and
    accumulation).  Predictions stay within bf16 operand noise of the fp32 oracle and every gradient within 6 % of its scale
    (8-bit mantissas through two layers and twelve steps of back-propagation; measured 3 %)."""
    import yt8m_amd.ops as ops
    monkeypatch.setattr(ops, "BF16_MIN_ROWS", 2)
    monkeypatch.setattr(ops, "BF16_MIN_MACS", 1)
    rs = np.random.RandomState(51)
    B, F, Dm, V = 8, 12, 16, 17
    flags.lstm_cells, flags.lstm_layers, flags.lstm_pipeline_chunks = str(Hh), 2, chunks
    flags.compute_dtype = "bfloat16"
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([12, 1, 5, 12, 3, 7, 12, 9], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(flm.LstmModel(), x, y, dev, nf=nf, rs=np.random.RandomState(53) if Hh < 256 else None,
                                P=None if Hh < 256 else _small_lstm_params(rs, Dm, Hh, V))
    tp = {k: T(v * 1.0).requires_grad_(True) for k, v in P.items()}
    with torch.no_grad():   # value emulation: both LSTM products on bf16-rounded operands (exactly what the device computes)
        ste = torch_ref.lstm_model_state(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2),
                                         bf16_operands=True if Hh % 256 == 0 else "input")
        pre = torch_ref.moe(torch_ref.bf16_round(ste), torch_ref.bf16_round(tp["gates/weights"]),
                            torch_ref.bf16_round(tp["experts/weights"]), tp["experts/biases"], 2)
    assert np.abs(H(res["predictions"]) - pre.numpy()).max() < 2e-3
    st = torch_ref.lstm_model_state(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2))
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 3e-2
    got = grads_of(g)
    for k, t in tp.items():
        ref = t.grad.numpy()
        assert np.abs(got[k] - ref).max() <= 6e-2 * max(np.abs(ref).max(), 1e-3), k


@pytest.mark.parametrize("cls", ["GruPoolingModel", "LayerNormLstmMemoryModel"])
def test_gru_and_layernorm_models_bf16_hoisted_products(dev, flags, cls, monkeypatch):
    """--compute_dtype=bfloat16 for the other recurrent cells: hoisted products on bf16 operands (ops.gemm_any), recurrence
    fp32; the bf16 step tracks the fp32 step of the same weights (loss within 1 %, every gradient within 6 % of its scale)."""
    import yt8m_amd.ops as ops
    monkeypatch.setattr(ops, "BF16_MIN_ROWS", 2)
    monkeypatch.setattr(ops, "BF16_MIN_MACS", 1)
    rs = np.random.RandomState(52)
    B, F, Dm, Hh, V = 8, 12, 32, 64, 17
    flags.gru_cells, flags.gru_layers, flags.lstm_cells, flags.lstm_layers = Hh, 2, str(Hh), 2
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([12, 1, 5, 12, 3, 7, 12, 9], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g32, res32, loss32, P = run_model(getattr(flm, cls)(), x, y, dev, nf=nf, rs=rs)
    ref = grads_of(g32)
    flags.compute_dtype = "bfloat16"
    g16, res16, loss16, _ = run_model(getattr(flm, cls)(), x, y, dev, nf=nf, P={k: v.astype(np.float32) for k, v in P.items()})
    assert abs(float(loss16) - float(loss32)) < 1e-2 * abs(float(loss32))
    assert np.abs(H(res16["predictions"]) - H(res32["predictions"])).max() < 3e-2
    got = grads_of(g16)
    for k, r in ref.items():
        assert np.abs(got[k] - r).max() <= 6e-2 * max(np.abs(r).max(), 1e-3), k


def test_tfrecord_to_training_step(dev, flags, tmp_path):
    """End to end through the widened path: fabricated frame-level TFRecord shard -> native reader -> pinned host ->
    device uint8 -> fused dequantise/normalise -> LstmModel step; the transform is checked against the oracle on the
    bytes that came out of the file."""
    from oracle import tfrecord_ref as tr
    import yt8m_amd.readers as readers
    import yt8m_amd.ops as ops
    rs = np.random.RandomState(21)
    names, sizes = ["rgb", "audio"], [24, 8]
    vids = []
    for i in range(6):
        k = int(rs.randint(1, 12))
        vids.append(dict(video_id=("v%d" % i).encode(), labels=[int(rs.randint(0, 50))],
                         frames={n: rs.randint(0, 256, size=(k, s)).astype(np.uint8) for n, s in zip(names, sizes)}))
    p = str(tmp_path / "train.tfrecord")
    tr.write_frame_shard(p, vids, names)
    rd = readers.YT8MFrameFeatureReader(num_classes=50, feature_sizes=sizes, feature_names=names, max_frames=10)
    flags.lstm_cells, flags.lstm_layers = "8", 1
    g = reset_default_graph(device=dev, seed=2)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=6, graph=g)
    n = 0
    for ids, q, lab, nf in rd.prepare_reader(p, batch_size=6, device=dev):
        eq, enf, elab = tr.expected_frame_batch(vids, names, sizes, 10, 50)
        assert q.is_cuda and np.array_equal(q.cpu().numpy(), eq) and np.array_equal(nf.cpu().numpy(), enf)
        x = ops.dequant_l2norm(q, nf)
        assert np.abs(H(x) - np_ref.dequant_l2norm_folded(eq, enf)).max() < 1e-6
        out = tg.step(q, lab, nf)
        assert torch.isfinite(out["loss"]) and out["predictions"].shape == (6, 50)
        n += len(ids)
    assert n == 6
    # the multi-threaded prefetcher (pinned slots lent zero-copy, async H2D) delivers the same bytes
    got = list(rd.prepare_reader([p], batch_size=4, device=dev, num_threads=2))
    eq, enf, elab = tr.expected_frame_batch(vids, names, sizes, 10, 50)
    assert [len(b[0]) for b in got] == [4, 2]
    assert np.array_equal(torch.cat([b[1] for b in got]).cpu().numpy(), eq)
    assert np.array_equal(torch.cat([b[3] for b in got]).cpu().numpy(), enf)
    assert np.array_equal(torch.cat([b[2] for b in got]).cpu().numpy(), elab.astype(bool))


def _step_seeds(n, step=1, graph_seed=0, rank=0):
    """Keys the graph's random stream hands out in forward pass `step` (run_model's second forward is pass 1)."""
    from yt8m_amd.variables import random_seed
    return [random_seed(graph_seed, rank, step, c) for c in range(n)]


def test_deep_combine_chain_with_dropout(dev, flags):
    """--dropout --keep_prob: tf.nn.dropout on every sub-model input (W/all_video_models/deep_combine_chain_model.py:57-58),
    not on the main head; masks replayed in backward."""
    rs = np.random.RandomState(11)
    B, Dm, V, L, C = 9, 24, 31, 3, 8
    flags.deep_chain_layers, flags.deep_chain_relu_cells, flags.support_type = L, C, "label,label,label"
    flags.support_loss_percent = 0.05
    flags.dropout, flags.keep_prob = True, 0.7
    x = rs.randn(B, Dm).astype(np.float32)
    y = rs.rand(B, V) < 0.1
    g, res, loss, P = run_model(vlm.DeepCombineChainModel(), x, y, dev, rs=rs, multitask=True)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.deep_combine_chain(T(x), tp, L, 2, dropout_spec=(0.7, _step_seeds(L)))
    lr = 0.95 * torch_ref.cross_entropy(main, T(y)) + 0.05 * torch_ref.cross_entropy(sup, T(np.tile(y, (1, L))))
    lr.backward()
    assert np.abs(H(res["predictions"]) - main.detach().numpy()).max() < 1e-5
    assert np.abs(H(res["support_predictions"]) - sup.detach().numpy()).max() < 1e-5
    check_grads(g, tp)
    # without dropout the predictions differ (the masks really were applied) and evaluation uses keep_prob = 1
    main0, _ = torch_ref.deep_combine_chain(T(x), tp, L, 2)
    assert np.abs(main0.detach().numpy() - main.detach().numpy()).max() > 1e-4
    tg = train.TrainGraph(vlm.DeepCombineChainModel(), batch_size=B, graph=g, multitask=True,
                          transformer_class=__import__("yt8m_amd.feature_transform", fromlist=["x"]).IdenticalTransformer)
    pe = tg.predict(torch.from_numpy(x).to(dev), vocab_size=V)
    assert np.abs(H(pe) - main0.detach().numpy()).max() < 1e-5


@pytest.mark.parametrize("chunks", [1, 4])
def test_lstm_memory_model_with_dropout_wrapper(dev, flags, chunks, honour_lstm_chunks):
    """DropoutWrapper(BasicLSTMCell, input_keep_prob) on both layers (W/all_frame_models/lstm_memory_model.py:36-45): masks on
    the layer inputs only, a fresh one per time step, the recurrent state untouched; chunking does not change the masks."""
    rs = np.random.RandomState(12)
    B, F, Dm, Hh, V = 6, 10, 12, 8, 17
    flags.lstm_cells, flags.lstm_layers, flags.lstm_pipeline_chunks = str(Hh), 2, chunks
    flags.dropout, flags.keep_prob = True, 0.6
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([10, 1, 5, 10, 3, 7], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(flm.LstmMemoryModel(), x, y, dev, nf=nf, rs=rs)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    _, c, _ = torch_ref.lstm_stack(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2), dropout_spec=(0.6, _step_seeds(2)))
    pr = torch_ref.moe(torch.cat(c, 1), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)
    _, c0, _ = torch_ref.lstm_stack(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 2))
    assert np.abs(torch.cat(c0, 1).detach().numpy() - torch.cat(c, 1).detach().numpy()).max() > 1e-3


def test_noise_level_flag(dev, flags):
    """--noise_level: N(0, level^2) on the chain's relu outputs / the LSTM memory (W/train.py:349-352,571): training forward
    differs from the noiseless one by the oracle's Philox normal stream; evaluation is noiseless."""
    from oracle import philox
    rs = np.random.RandomState(13)
    B, F, Dm, Hh, V = 5, 6, 12, 8, 17
    flags.lstm_cells, flags.lstm_layers, flags.noise_level = str(Hh), 1, 0.25
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.full(B, F, dtype=np.int32)
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(flm.LstmMemoryModel(), x, y, dev, nf=nf, rs=rs)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    _, c, _ = torch_ref.lstm_stack(T(x), torch.from_numpy(nf), _lstm_ref_layers(tp, 1))
    st = c[0] + T(philox.add_noise(np.zeros((B, Hh)), 0.25, _step_seeds(1)[0]))
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)


def _gru_ref_layers(tp, L):
    s = "RNN/multi_rnn_cell/cell_%d/gru_cell/%s"
    return [(tp[s % (l, "gates/weights")], tp[s % (l, "gates/biases")], tp[s % (l, "candidate/weights")],
             tp[s % (l, "candidate/biases")]) for l in range(L)]


@pytest.mark.parametrize("cls", ["GruPoolingModel", "GruWithPoolingModel"])
def test_gru_models(dev, flags, cls):
    """tf.contrib.rnn.GRUCell stack (W/all_frame_models/gru_pooling_model.py, gru_with_pooling_model.py): ragged num_frames
    (copy-through + zero outputs), mean over the video's frames, [pooled || h_0 || h_1] for the WithPooling variant."""
    rs = np.random.RandomState(21)
    B, F, Dm, Hh, V = 6, 9, 12, 8, 17
    flags.gru_cells, flags.gru_layers = Hh, 2
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([9, 1, 5, 9, 3, 7], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(getattr(flm, cls)(), x, y, dev, nf=nf, rs=rs)
    assert g.vars["RNN/multi_rnn_cell/cell_1/gru_cell/gates/weights"].shape == (2 * Hh, 2 * Hh)
    assert g.vars["gates/weights"].shape[0] == (Hh if cls == "GruPoolingModel" else 3 * Hh)
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    outs, hfin = torch_ref.gru_stack(T(x), torch.from_numpy(nf), _gru_ref_layers(tp, 2))
    pooled = outs.sum(1) / T(np.maximum(nf, 1)).unsqueeze(1)
    st = pooled if cls == "GruPoolingModel" else torch.cat([pooled] + hfin, 1)
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)
    # fresh variables: the gate bias starts at 1 (TF-1.0 GRUCell), everything else as slim / zeros
    g2 = reset_default_graph(device=dev, seed=0)
    g2.begin_step()
    getattr(flm, cls)().create_model(torch.from_numpy(x).to(dev), vocab_size=V, num_frames=torch.from_numpy(nf).to(dev))
    assert float(g2.vars["RNN/multi_rnn_cell/cell_0/gru_cell/gates/biases"].data.min()) == 1.0
    assert float(g2.vars["RNN/multi_rnn_cell/cell_0/gru_cell/candidate/biases"].data.abs().max()) == 0.0


@pytest.mark.parametrize("keep", [1.0, 0.7])
def test_layernorm_lstm_memory_model(dev, flags, keep):
    """tf.contrib.rnn.LayerNormBasicLSTMCell stack (W/all_frame_models/layernorm_lstm_memory_model.py): five layer norms per
    step with their gamma / beta gradients, normalised cell state carried, recurrent dropout on the candidate with --dropout."""
    rs = np.random.RandomState(22)
    B, F, Dm, Hh, V = 6, 8, 12, 10, 17
    flags.lstm_cells, flags.lstm_layers = str(Hh), 2
    if keep < 1:
        flags.dropout, flags.keep_prob = True, keep
    x = rs.randn(B, F, Dm).astype(np.float32)
    nf = np.array([8, 1, 5, 8, 3, 7], dtype=np.int32)
    x *= (np.arange(F)[None, :, None] < nf[:, None, None])
    y = rs.rand(B, V) < 0.15
    g, res, loss, P = run_model(flm.LayerNormLstmMemoryModel(), x, y, dev, nf=nf, rs=rs)
    s = "RNN/multi_rnn_cell/cell_%d/layer_norm_basic_lstm_cell/%s"
    assert g.vars[s % (1, "weights")].shape == (2 * Hh, 4 * Hh) and (s % (0, "biases")) not in g.vars
    assert g.vars["gates/weights"].shape[0] == 2 * Hh
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    layers = [(tp[s % (l, "weights")], [tp[s % (l, n + "/gamma")] for n in flm.LN_GATES],
               [tp[s % (l, n + "/beta")] for n in flm.LN_GATES]) for l in range(2)]
    _, c, _ = torch_ref.lnlstm_stack(T(x), torch.from_numpy(nf), layers,
                                     dropout_spec=None if keep >= 1 else (keep, _step_seeds(2)))
    pr = torch_ref.moe(torch.cat(c, 1), tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(H(res["predictions"]) - pr.detach().numpy()).max() < 1e-4
    check_grads(g, tp, tol=5e-4)
    g2 = reset_default_graph(device=dev, seed=0)
    g2.begin_step()
    flm.LayerNormLstmMemoryModel().create_model(torch.from_numpy(x).to(dev), vocab_size=V, num_frames=torch.from_numpy(nf).to(dev))
    assert float(g2.vars[s % (0, "state/gamma")].data.min()) == 1.0 and float(g2.vars[s % (0, "input/beta")].data.abs().max()) == 0.0
