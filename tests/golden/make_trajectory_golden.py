"""Writes tests/golden/trajectory_kat.json: STEPS consecutive training steps of BASELINE configs[3] (LstmModel 2 x 1024, MoE head) at
its full size [128, 300, 1152] -- forward, CrossEntropyLoss, backward, + l2 * w, per-tensor clip_by_norm, TF-Adam with the staircase
learning rate (W/train.py:301-311,435-466; W/utils.py:164-174) -- computed by the fp64 torch restatement (oracle/torch_ref.py: the
models through autograd, TFAdam).  Per step: the label loss and every tensor's gradient norm as the clip sees it (data gradient +
l2 * w); after the last step: checksums of every parameter tensor.  A different batch per step (fullsize_cases.make_inputs(cfg, step)).
VERDICT r5 #6 / #8: the single-pass fixtures pin one forward + backward per configuration; this one pins the optimiser trajectory at
full size, and the HIP replay (tests/test_gpu_trajectory.py) runs it with the f16 ("h2") recurrences and products engaged.
The reference itself cannot produce it (Python 2 / TensorFlow 1.0).  ~12 minutes, ~25 GB:  python tests/golden/make_trajectory_golden.py"""
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
CFG, STEPS = "c3_lstm", int(os.environ.get("TRAJ_STEPS", "5"))
# --base_learning_rate of the reference's LSTM runs (W/training_scripts/run-lstm-memory-cell1024.sh:11); the staircase is made to step
# INSIDE the trajectory (every 256 examples = 2 steps; the default 4 000 000 would never move in five steps)
HYPER = dict(base_lr=0.0008, batch_size=fc.BATCH[CFG], l2=1e-8, clip=1.0, decay_examples=256, decay=0.95)
REG = ["gates/weights", "experts/weights"]                                     # slim l2_regularizer(1e-8) sites of MoeModel (W/all_video_models/moe_model.py:43-53)
dst = os.path.join(HERE, "trajectory_kat.json")


def T(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def frames(q, nf):
    x = torch_ref.l2_normalize(torch_ref.dequantize(torch.from_numpy(q), torch.float64), 2)
    mask = torch.arange(q.shape[1])[None, :] < torch.from_numpy(nf.astype(np.int64))[:, None]
    return x * mask[:, :, None].to(x.dtype)


P = {k: T(v).requires_grad_(True) for k, v in fc.make_params(CFG).items()}
opt = torch_ref.TFAdam(P, REG, **HYPER)
rec = {"config": CFG, "steps": [], "hyper": HYPER, "regularised": REG}
t00 = time.time()
for s in range(STEPS):
    t0 = time.time()
    I = fc.make_inputs(CFG, s)
    nf = torch.from_numpy(I["nf"].astype(np.int64))
    layers = [(P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    st = torch_ref.lstm_model_state(frames(I["x"], I["nf"]), nf, layers)
    p = torch_ref.moe(st, P["gates/weights"], P["experts/weights"], P["experts/biases"], fc.M)
    loss = torch_ref.cross_entropy(p, T(I["y"]))
    loss.backward()
    norms = {}
    with torch.no_grad():
        for k, w in P.items():
            g = w.grad + (HYPER["l2"] * w if k in REG else 0.0)
            norms[k] = float(g.norm())
    rec["steps"].append({"loss": float(loss), "grad_norms": norms,
                         "learning_rate": torch_ref.exponential_decay(HYPER["base_lr"], s, HYPER["batch_size"], HYPER["decay_examples"],
                                                                      HYPER["decay"])})
    opt.step()
    print("step %d loss %.6f  %.0f s" % (s, float(loss), time.time() - t0), flush=True)
    del st, p, loss
rec["params"] = {k: fc.checksum(v.detach().numpy()) for k, v in P.items()}
rec["seconds"] = round(time.time() - t00, 1)
json.dump(rec, open(dst, "w"), indent=0, sort_keys=True)
print("wrote", dst)
