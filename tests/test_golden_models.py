"""Replays the committed golden fixture tests/golden/models_kat.json (seeded tiny cases, expected outputs from the fp64 numpy
restatement at the time the fixture was written -- tests/golden/make_models_golden.py):
  * CPU: both restatements (oracle/np_ref.py, oracle/torch_ref.py) still reproduce the stored numbers;
  * GPU: the HIP path reproduces them through the C ABI.
The reference cannot generate these vectors (Python 2 / TF 1.0); the fixture pins the restatement against drift."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import model_cases  # noqa: E402
from oracle import np_ref, torch_ref  # noqa: E402

KAT = json.load(open(os.path.join(HERE, "golden", "models_kat.json")))


def expected(name, key):
    e = KAT[name][key]
    return np.asarray(e["values"], dtype=np.float64).reshape(e["shape"])


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


@pytest.mark.parametrize("name", model_cases.CASES)
def test_numpy_restatement_reproduces_fixture(name):
    got = model_cases.case_outputs(name, np_ref)
    assert set(got) == set(KAT[name])
    for k, v in got.items():
        assert np.abs(np.asarray(v) - expected(name, k)).max() < 1e-12, (name, k)


def test_torch_restatement_reproduces_fixture():
    c = model_cases.case_inputs("moe")
    p = torch_ref.moe(T(c["x"]), T(c["Wg"]), T(c["We"]), T(c["be"]), c["M"])
    assert np.abs(p.numpy() - expected("moe", "predictions")).max() < 1e-12
    c = model_cases.case_inputs("chain")
    main, sup = torch_ref.deep_combine_chain(T(c["x"]), {k: T(v) for k, v in c["P"].items()}, c["L"], c["M"])
    assert np.abs(main.numpy() - expected("chain", "predictions")).max() < 1e-12
    assert np.abs(sup.numpy() - expected("chain", "support_predictions")).max() < 1e-12
    c = model_cases.case_inputs("lstm")
    out, cs, hs = torch_ref.lstm_stack(T(c["x"]), T(c["nf"]), [(T(W), T(b)) for W, b in c["layers"]])
    assert np.abs(out.numpy() - expected("lstm", "outputs")).max() < 1e-12
    c = model_cases.case_inputs("gru")
    out, hs = torch_ref.gru_stack(T(c["x"]), T(c["nf"]), [tuple(T(a) for a in lay) for lay in c["layers"]])
    assert np.abs(out.numpy() - expected("gru", "outputs")).max() < 1e-12 and np.abs(hs[1].numpy() - expected("gru", "h1")).max() < 1e-12
    c = model_cases.case_inputs("lnlstm")
    out, cs, hs = torch_ref.lnlstm_stack(T(c["x"]), T(c["nf"]), [(T(W), [T(g) for g in ga], [T(b) for b in be]) for W, ga, be in c["layers"]])
    assert np.abs(out.numpy() - expected("lnlstm", "outputs")).max() < 1e-11 and np.abs(cs[1].numpy() - expected("lnlstm", "c1")).max() < 1e-11
    c = model_cases.case_inputs("netvlad")
    v = torch_ref.netvlad(T(c["x"]), T(c["nf"]), T(c["Wc"]), T(c["bc"]), T(c["centres"]))
    v = v[0] if isinstance(v, tuple) else v
    assert np.abs(v.numpy().reshape(-1) - expected("netvlad", "vlad").reshape(-1)).max() < 1e-12
    c = model_cases.case_inputs("xent")
    assert abs(float(torch_ref.cross_entropy(T(c["p"]), T(c["y"]))) - float(expected("xent", "loss"))) < 1e-12
    assert abs(float(torch_ref.cross_entropy(T(c["p"]), T(c["y"]), weights=T(c["w"]))) - float(expected("xent", "weighted_loss"))) < 1e-12


def test_torch_restatement_reproduces_round4_cases():
    c = model_cases.case_inputs("logistic")
    assert np.abs(torch_ref.logistic(T(c["x"]), T(c["W"]), T(c["b"])).numpy() - expected("logistic", "predictions")).max() < 1e-12
    c = model_cases.case_inputs("attention")
    lay = [(T(W), T(b)) for W, b in c["layers"]]
    out, _, _ = torch_ref.lstm_stack(T(c["x"]), T(c["nf"]), lay)
    pooled = torch_ref.attention_pool(T(c["x"]), out, T(c["nf"]), T(c["Wa"]), T(c["ba"]))
    pooled = pooled[0] if isinstance(pooled, tuple) else pooled
    assert np.abs(pooled.numpy() - expected("attention", "pooled")).max() < 1e-12
    p = torch_ref.lstm_attention_max_pooling(T(c["x"]), T(c["nf"]), lay, T(c["Wa"]), T(c["ba"]), T(c["Wg"]), T(c["We"]), T(c["be"]), c["M"])
    assert np.abs(p.numpy() - expected("attention", "predictions")).max() < 1e-12
    c = model_cases.case_inputs("dbof_bn")
    h = torch_ref.dbof_model_bn(T(c["xs"]), {k: T(v) for k, v in c["P"].items()})
    assert np.abs(h.numpy() - expected("dbof_bn", "hidden")).max() < 1e-12
    c = model_cases.case_inputs("multitask")
    sl = expected("multitask", "support_labels")
    loss = torch_ref.cross_entropy(T(c["p"]), T(c["y"])) * (1 - c["percent"]) + torch_ref.cross_entropy(T(c["sp"]), T(sl)) * c["percent"]
    assert abs(float(loss) - float(expected("multitask", "loss"))) < 1e-12
    c = model_cases.case_inputs("trainstep")                       # torch_ref.TFAdam: same LR staircase, l2, per-tensor clip, TF-Adam
    params = {k: T(v).clone().requires_grad_(True) for k, v in c["P"].items()}
    opt = torch_ref.TFAdam(params, set(c["regularised"]), base_lr=c["base_lr"], batch_size=c["batch_size"], l2=c["l2"], clip=c["clip"])
    opt.step_no = c["step"]
    for k, (m, v) in c["state"].items():
        opt.m[k], opt.v[k] = T(m).clone(), T(v).clone()
    for k, t in params.items():
        t.grad = T(c["G"][k]).clone()
    opt.step()
    for k, t in params.items():
        assert np.abs(t.detach().numpy() - expected("trainstep", "param:" + k)).max() < 1e-12, k
        assert np.abs(opt.m[k].numpy() - expected("trainstep", "m:" + k)).max() < 1e-13 and \
            np.abs(opt.v[k].numpy() - expected("trainstep", "v:" + k)).max() < 1e-13
    assert float(expected("trainstep", "lr")) == pytest.approx(0.01 * 0.95)          # 4500 * 1024 examples: one decay (W/train.py:303-308)


def _vars(dev, arrays):
    from yt8m_amd.variables import reset_default_graph, zeros
    g = reset_default_graph(device=dev)
    g.begin_step()
    vs = [g.get_variable("v%d" % i, a.shape, zeros) for i, a in enumerate(arrays)]
    g.finalize()
    for v, a in zip(vs, arrays):
        v.data.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev))
    return vs


@pytest.mark.gpu
def test_hip_path_reproduces_fixture(dev):
    import yt8m_amd.ops as ops
    import yt8m_amd.seq_ops as seq_ops

    def D(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dev)

    def Hn(t):
        return t.detach().cpu().numpy().astype(np.float64)

    c = model_cases.case_inputs("moe")
    Wg, We, be = _vars(dev, [c["Wg"], c["We"], c["be"]])
    p = ops.moe_head(D(c["x"]), Wg, We, be, 11, c["M"])
    assert np.abs(Hn(p) - expected("moe", "predictions")).max() < 1e-5
    c = model_cases.case_inputs("lstm")
    flat = [a for lay in c["layers"] for a in lay]
    vs = _vars(dev, flat)
    xt = D(c["x"]).transpose(0, 1).contiguous()
    nf = torch.from_numpy(c["nf"].astype(np.int32)).to(dev)
    out, fin = seq_ops.lstm_stack(xt, nf, [(vs[0], vs[1]), (vs[2], vs[3])], chunks=2)
    assert np.abs(Hn(out).transpose(1, 0, 2) - expected("lstm", "outputs")).max() < 1e-5
    st = torch.cat([t for pair in fin for t in pair], dim=1)
    assert np.abs(Hn(st) - expected("lstm", "state")).max() < 1e-5
    c = model_cases.case_inputs("gru")
    x_tm = D(c["x"]).transpose(0, 1).contiguous()
    hs = []
    vs = _vars(dev, [a for lay in c["layers"] for a in lay])
    for l in range(2):
        x_tm, h = seq_ops.gru_layer(x_tm, *vs[4 * l:4 * l + 4], nf)
        hs.append(h)
    assert np.abs(Hn(x_tm).transpose(1, 0, 2) - expected("gru", "outputs")).max() < 1e-5
    assert np.abs(Hn(hs[0]) - expected("gru", "h0")).max() < 1e-5 and np.abs(Hn(hs[1]) - expected("gru", "h1")).max() < 1e-5
    c = model_cases.case_inputs("lnlstm")
    arrays = []
    for W, ga, be_ in c["layers"]:
        arrays += [W] + list(ga) + list(be_)
    vs = _vars(dev, arrays)
    x_tm = D(c["x"]).transpose(0, 1).contiguous()
    cs = []
    for l in range(2):
        v = vs[11 * l:11 * l + 11]
        x_tm, cfin, hfin = seq_ops.lnlstm_layer(x_tm, v[0], v[1:6], v[6:11], nf)
        cs.append(cfin)
    assert np.abs(Hn(x_tm).transpose(1, 0, 2) - expected("lnlstm", "outputs")).max() < 2e-5
    assert np.abs(Hn(cs[0]) - expected("lnlstm", "c0")).max() < 2e-5 and np.abs(Hn(cs[1]) - expected("lnlstm", "c1")).max() < 2e-5
    c = model_cases.case_inputs("netvlad")                        # generic float path: FC -> masked softmax -> pooling -> finish
    Wc, bc, cen = _vars(dev, [c["Wc"], c["bc"], c["centres"]])
    xn = D(c["x"])
    nfv = torch.from_numpy(c["nf"].astype(np.int32)).to(dev)
    a = seq_ops.masked_softmax_rows(ops.linear(xn, Wc, bc), nfv)
    assert np.abs(Hn(a) - expected("netvlad", "assignment")).max() < 1e-5
    vlad = seq_ops.vlad_finish(seq_ops.pool_tn(a, xn), a, cen)
    v = ops.l2_normalize(vlad.reshape(xn.shape[0], -1))
    assert np.abs(Hn(v) - expected("netvlad", "vlad")).max() < 1e-5
    c = model_cases.case_inputs("chain")                          # DeepCombineChainModel through the plugin surface
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph
    FLAGS.reset()
    FLAGS.deep_chain_layers, FLAGS.deep_chain_relu_cells, FLAGS.moe_num_mixtures = c["L"], 5, c["M"]
    g = reset_default_graph(device=dev)
    g.begin_step()
    model = vlm.DeepCombineChainModel()
    model.create_model(D(c["x"]), vocab_size=11)
    g.finalize()
    for k, v_ in c["P"].items():
        g.vars[k].data.copy_(D(v_).view(g.vars[k].data.shape))
    g.begin_step()
    res = model.create_model(D(c["x"]), vocab_size=11)
    FLAGS.reset()
    assert np.abs(Hn(res["predictions"]) - expected("chain", "predictions")).max() < 1e-5
    assert np.abs(Hn(res["support_predictions"]) - expected("chain", "support_predictions")).max() < 1e-5
    c = model_cases.case_inputs("xent")
    loss, _ = ops.xent_fwd(D(c["p"]), D(c["y"]), None, want_dp=False, upstream=1.0)
    assert abs(float(loss) - float(expected("xent", "loss"))) < 1e-4 * abs(float(expected("xent", "loss")))


@pytest.mark.gpu
def test_hip_path_reproduces_round4_cases(dev):
    """Logistic, attention pooling (+ the whole LstmAttentionMaxPoolingModel), DBoF with batch norm, the multitask loss and one
    clip + Adam + LR-staircase step through the product's plugin surface / ops against the committed fp64 numbers."""
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.losses as losses
    import yt8m_amd.ops as ops
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.feature_transform import IdenticalTransformer
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph

    def D(a, dt=np.float32):
        return torch.from_numpy(np.asarray(a, dtype=dt)).to(dev)

    def Hn(t):
        return t.detach().cpu().numpy().astype(np.float64)

    def plugin(model, x, P, V=11, nf=None, **flags):
        FLAGS.reset()
        for k, v in flags.items():
            setattr(FLAGS, k, v)
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(model, batch_size=x.shape[0], graph=g, transformer_class=IdenticalTransformer)
        y = torch.zeros((x.shape[0], V), dtype=torch.bool, device=dev)
        nfd = None if nf is None else D(nf, np.int32)
        tg.forward(D(x), y, nfd)
        g.finalize()
        assert set(P) <= set(g.vars), (sorted(P), sorted(g.vars))
        for k, v in P.items():
            g.vars[k].data.copy_(D(v).view(g.vars[k].data.shape))
        res = tg.forward(D(x), y, nfd)
        FLAGS.reset()
        return g, tg, res

    c = model_cases.case_inputs("logistic")
    _, _, res = plugin(vlm.LogisticModel(), c["x"], {"fully_connected/weights": c["W"], "fully_connected/biases": c["b"]})
    assert np.abs(Hn(res["predictions"]) - expected("logistic", "predictions")).max() < 1e-5

    c = model_cases.case_inputs("attention")
    P = {"attention-/weights": c["Wa"], "attention-/biases": c["ba"], "gates-sub-moe/weights": c["Wg"], "experts-sub-moe/weights": c["We"],
         "experts-sub-moe/biases": c["be"]}
    for l, (W, b) in enumerate(c["layers"]):
        P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l] = W
        P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l] = b
    _, _, res = plugin(flm.LstmAttentionMaxPoolingModel(), c["x"], P, nf=c["nf"], lstm_cells="8", lstm_layers=2, lstm_attentions=3)
    assert np.abs(Hn(res["predictions"]) - expected("attention", "predictions")).max() < 1e-5

    c = model_cases.case_inputs("dbof_bn")                        # slim.batch_norm in training mode around the two DBoF products
    names = ["input_bn", "cluster_bn", "hidden1_bn"]
    arrays = []
    for n in names:
        C = c["P"][n + "/gamma"].shape[0]
        arrays += [c["P"][n + "/gamma"], c["P"][n + "/beta"], np.zeros(C), np.ones(C)]
    vs = _vars(dev, arrays + [c["P"]["Variable"], c["P"]["Variable_1"]])
    B, S, Dd = c["xs"].shape

    def bn(x, i):
        return ops.batch_norm(x, vs[4 * i], vs[4 * i + 1], vs[4 * i + 2], vs[4 * i + 3], True)

    r = bn(D(c["xs"]).view(-1, Dd), 0)
    a = bn(r @ vs[12].data, 1).clamp(0, 6).view(B, S, -1)
    h = bn(ops.frame_pool(a, "max") @ vs[13].data, 2).clamp(0, 6)
    assert np.abs(Hn(h) - expected("dbof_bn", "hidden")).max() < 2e-5

    c = model_cases.case_inputs("multitask")
    FLAGS.reset()
    FLAGS.support_type, FLAGS.num_frequents, FLAGS.support_loss_percent = c["support_type"], c["num_frequents"], c["percent"]
    loss = losses.MultiTaskCrossEntropyLoss().calculate_loss(D(c["p"]), D(c["sp"]), D(c["y"], bool))
    FLAGS.reset()
    assert abs(float(loss) - float(expected("multitask", "loss"))) < 1e-5 * abs(float(expected("multitask", "loss")))

    c = model_cases.case_inputs("trainstep")                       # the optimiser slice alone: arenas filled by hand, one fused pass
    import math
    g, tg, _ = plugin(vlm.LogisticModel(), model_cases.case_inputs("logistic")["x"], c["P"])
    for k, var in g.vars.items():
        var.grad.copy_(D(c["G"][k]).view(var.grad.shape))
        var.grad_written = True
        m, v = c["state"][k]
        g.adam_m[var.offset:var.offset + var.numel()].copy_(D(m).view(-1))
        g.adam_v[var.offset:var.offset + var.numel()].copy_(D(v).view(-1))
    g.l2.copy_(torch.tensor([c["l2"] if v.name in c["regularised"] else 0.0 for v in g.trainable_variables()], device=dev))
    lr = train.exponential_decay(c["base_lr"], c["step"], c["batch_size"], 4000000, 0.95)
    assert lr == pytest.approx(float(expected("trainstep", "lr")), rel=1e-12)
    t = c["step"] + 1
    ops.sqnorm_and_adam(g, lr * math.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t), gscale=1.0, clip=c["clip"], beta1=0.9, beta2=0.999, eps=1e-8)
    for k, var in g.vars.items():
        e = expected("trainstep", "param:" + k)
        assert np.abs(Hn(var.data).reshape(e.shape) - e).max() < 1e-6, k
