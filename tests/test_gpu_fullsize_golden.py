"""-m gpu: the five BASELINE.json configurations at their full sizes through the HIP path, forward AND backward, against the
committed checksums of the fp64 restatement (tests/golden/fullsize_kat.json, written by tests/golden/make_fullsize_golden.py;
SURVEY.md 8c last row).  Inputs and weights are regenerated from seeds (tests/golden/fullsize_cases.py); compared per tensor:
sum and abs-sum, and the 20 largest-magnitude entries by VALUE at the stored positions (an index comparison would flip on
near-ties)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fullsize_cases as fc  # noqa: E402

import yt8m_amd.frame_level_models as flm  # noqa: E402
import yt8m_amd.losses as losses  # noqa: E402
import yt8m_amd.train as train  # noqa: E402
import yt8m_amd.video_level_models as vlm  # noqa: E402
from yt8m_amd.variables import reset_default_graph  # noqa: E402

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(HERE, "golden", "fullsize_kat.json")))


def _check(name, got, ref, rel, overlap=16):
    """got: device tensor; ref: fixture checksum.  Tolerances are relative to the tensor's own mean magnitude (sum / abs-sum) and
    to its largest entry (top-20 values)."""
    assert got.numel() == ref["n"], (name, got.numel(), ref["n"])
    g = got.detach().double().flatten()
    # floor: a tensor whose true value is zero (the attention biases: a constant added to an attention's logits over all frames does
    # not move its softmax over frames, so their gradient vanishes identically -- fp64 gives 1e-17, fp32 products 1e-9) is compared on
    # the scale of fp32 rounding of O(1) terms, not on its own
    scale = ref["abs_sum"] + 1e-6 * ref["n"]
    assert abs(float(g.sum()) - ref["sum"]) <= rel * scale, (name, "sum", float(g.sum()), ref["sum"], scale)
    assert abs(float(g.abs().sum()) - ref["abs_sum"]) <= rel * scale, (name, "abs_sum", float(g.abs().sum()), ref["abs_sum"])
    if ref["abs_sum"] < 1e-9 * ref["n"]:
        return                                                      # nothing but rounding noise to rank
    idx = torch.tensor(ref["top_idx"], device=g.device)
    vals = g[idx].cpu().numpy()
    top = np.asarray(ref["top_val"])
    assert np.abs(vals - top).max() <= rel * 10 * np.abs(top).max() + 1e-30, (name, "top-20 values", vals[:4], top[:4])
    # the device's own largest entries are (nearly) the same set: at least 16 of 20 positions agree
    mine = set(torch.topk(g.abs(), 20).indices.cpu().tolist())
    assert len(mine & set(ref["top_idx"])) >= overlap, (name, "top-20 positions", sorted(mine)[:5], sorted(ref["top_idx"])[:5])


MODELS = {"c0_logistic": vlm.LogisticModel, "c1_moe": vlm.MoeModel, "c2_netvlad": flm.NetVLADModel, "c3_lstm": flm.LstmModel,
          "c4_composite_bf16": flm.GatedNetVLADAttentionChainModel}


@pytest.mark.parametrize("cfg", fc.CONFIGS)
def test_full_size_configuration_matches_the_fp64_checksums(dev, flags, cfg):
    if cfg not in KAT:
        pytest.skip("no fixture for %s (tests/golden/make_fullsize_golden.py %s)" % (cfg, cfg))
    ref = KAT[cfg]
    bf16 = cfg.endswith("bf16")
    multitask = cfg.startswith("c4")
    if bf16:
        flags.compute_dtype = "bfloat16"
    if multitask:
        flags.deep_chain_layers, flags.deep_chain_relu_cells, flags.lstm_attentions = fc.CH_L, fc.CH_C, fc.A
        flags.support_type, flags.support_loss_percent = ",".join(["label"] * fc.CH_L), 0.1
    I = fc.make_inputs(cfg)
    B = fc.BATCH[cfg]
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(MODELS[cfg](), batch_size=B, graph=g, multitask=multitask,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss() if multitask else None)
    x, y = torch.from_numpy(I["x"]).to(dev), torch.from_numpy(I["y"]).to(dev)
    nf = None if I["nf"] is None else torch.from_numpy(I["nf"]).to(dev)
    tg.forward(x, y, nf)
    g.finalize()
    P = fc.make_params(cfg)
    assert {k: tuple(v.data.shape) for k, v in g.vars.items()} == {k: tuple(v.shape) for k, v in P.items()}
    for k, v in P.items():
        g.vars[k].data.copy_(torch.from_numpy(v).to(dev))
    del P
    res = tg.forward(x, y, nf)
    loss = tg.loss(res, y)
    loss.backward()
    torch.cuda.synchronize()
    # fp32 path: 1e-4 of the mean magnitude (six-product bf16-pipe GEMMs + fp32 accumulation over up to 38 400 rows); the bf16
    # configuration rounds both operands of every head product to 8 bits: 2e-2
    rel = 2e-2 if bf16 else 1e-4
    assert abs(float(loss) - ref["loss"]) <= (5e-3 if bf16 else 1e-5) * abs(ref["loss"]), (float(loss), ref["loss"])
    _check("predictions", res["predictions"], ref["predictions"], rel)
    if multitask:
        _check("support_predictions", res["support_predictions"], ref["support_predictions"], rel)
    for k, c in ref["grads"].items():
        # fp32 configurations: 5e-4 of the tensor's mean magnitude.  The bf16 configuration rounds the operands of EVERY large product
        # to 8 bits on the device (heads, relu cells, the f16 single-operand NetVLAD pooling) while the fixture's value emulation
        # rounds the heads' only: gradients agree in scale and direction (tests/test_gpu_round2.py::test_config5_composite_bf16_engaged
        # bounds them the same way), measured up to 11 % on the abs-sum of the attention weights' gradient -- bound 20 %
        if bf16:
            _check("grad " + k, g.vars[k].grad, c, 0.2, overlap=8)
        else:
            _check("grad " + k, g.vars[k].grad, c, 5e-4)
