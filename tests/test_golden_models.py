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
