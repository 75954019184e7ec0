"""Writes tests/golden/fullsize_kat.json: checksums of predictions, loss and EVERY gradient tensor of the five BASELINE.json
configurations at their full sizes (tests/golden/fullsize_cases.py), computed by the fp64 torch restatement (oracle/torch_ref.py
autograd; SURVEY.md 8c last row).  The reference itself cannot produce them (Python 2 / TensorFlow 1.0): like models_kat.json this
pins the RESTATEMENT and gives the HIP path one whole-configuration comparison, backward pass included, at the real shapes.
Takes a few minutes and ~25 GB of host memory:   python tests/golden/make_fullsize_golden.py [config ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import torch_ref  # noqa: E402
import fullsize_cases as fc  # noqa: E402

torch.set_num_threads(os.cpu_count() or 1)
dst = os.path.join(HERE, "fullsize_kat.json")
out = json.load(open(dst)) if os.path.exists(dst) else {}


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def frames(q, nf):
    x = torch_ref.l2_normalize(torch_ref.dequantize(torch.from_numpy(q), torch.float64), 2)
    mask = torch.arange(q.shape[1])[None, :] < torch.from_numpy(nf.astype(np.int64))[:, None]
    return x * mask[:, :, None].to(x.dtype)                       # zero padding AFTER dequantise (W/readers.py:178-187)


for cfg in (sys.argv[1:] or fc.CONFIGS):
    t0 = time.time()
    P = {k: T(v).requires_grad_(True) for k, v in fc.make_params(cfg).items()}
    I = fc.make_inputs(cfg)
    y = T(I["y"])
    nf = None if I["nf"] is None else torch.from_numpy(I["nf"].astype(np.int64))
    sup = None
    if cfg == "c0_logistic":
        p = torch_ref.logistic(torch_ref.l2_normalize(T(I["x"]), 1), P["fully_connected/weights"], P["fully_connected/biases"])
    elif cfg == "c1_moe":
        p = torch_ref.moe(torch_ref.l2_normalize(T(I["x"]), 1), P["gates/weights"], P["experts/weights"], P["experts/biases"], fc.M)
    elif cfg == "c2_netvlad":
        h = torch_ref.netvlad_hidden(frames(I["x"], I["nf"]), nf, P["netvlad/cluster_weights"], P["netvlad/cluster_biases"],
                                     P["netvlad/centres"], P["netvlad/hidden/weights"], P["netvlad/hidden/biases"])
        p = torch_ref.moe(h, P["gates/weights"], P["experts/weights"], P["experts/biases"], fc.M)
    elif cfg == "c3_lstm":
        layers = [(P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
                  for l in range(2)]
        st = torch_ref.lstm_model_state(frames(I["x"], I["nf"]), nf, layers)
        p = torch_ref.moe(st, P["gates/weights"], P["experts/weights"], P["experts/biases"], fc.M)
    else:
        p, sup = torch_ref.gated_netvlad_attention_chain(frames(I["x"], I["nf"]), nf, P, fc.CH_L, fc.M, fc.A, bf16_heads=True)
    if sup is None:
        loss = torch_ref.cross_entropy(p, y)
    else:                                                         # MultiTaskCrossEntropyLoss, support_type "label" x L, 10 % (W/losses.py:271-279)
        loss = 0.9 * torch_ref.cross_entropy(p, y) + 0.1 * torch_ref.cross_entropy(sup, y.repeat(1, fc.CH_L))
    loss.backward()
    rec = {"batch": fc.BATCH[cfg], "loss": float(loss), "predictions": fc.checksum(p.detach().numpy()), "grads": {}}
    if sup is not None:
        rec["support_predictions"] = fc.checksum(sup.detach().numpy())
    for k, v in P.items():
        rec["grads"][k] = fc.checksum(v.grad.numpy())
    rec["seconds"] = round(time.time() - t0, 1)
    out[cfg] = rec
    print(cfg, "loss %.6f" % rec["loss"], "%.0f s" % rec["seconds"], flush=True)
    del P, p, loss
    json.dump(out, open(dst, "w"), indent=0, sort_keys=True)
print("wrote", dst)
