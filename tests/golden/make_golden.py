"""Generates tests/golden/metrics_kat.json by IMPORTING the reference's pure-numpy metric code.

Run in the build container only (needs /root/reference; never runs on the GPU box):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Only numbers (inputs as seeds/small literals + outputs) are written; no reference source is stored.
The reference modules imported are W/eval_util.py, W/average_precision_calculator.py and
W/mean_average_precision_calculator.py (W = /root/reference/youtube-8m-wangheda); eval_util's only
TensorFlow import (`gfile`, eval_util.py:19, unused) is satisfied by an empty stub module.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/youtube-8m-wangheda"
for n in ["tensorflow", "tensorflow.python", "tensorflow.python.platform", "tensorflow.python.platform.gfile"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["tensorflow.python.platform"].gfile = sys.modules["tensorflow.python.platform.gfile"]
sys.path.insert(0, REF)
import average_precision_calculator as apc  # noqa: E402
import eval_util  # noqa: E402
import mean_average_precision_calculator as mapc  # noqa: E402


def f(x):
    return float(x)


out = {"_generator": "tests/golden/make_golden.py", "_python": sys.version.split()[0], "_numpy": np.__version__}

# C1-C3
p4, a4 = [.9, .8, .7, .6], [1, 0, 1, 0]
out["C1_ap"] = f(apc.AveragePrecisionCalculator.ap(np.array(p4), np.array(a4)))
out["C2_ap_at_2"] = f(apc.AveragePrecisionCalculator.ap_at_n(np.array(p4), np.array(a4), n=2))
out["C3_ap_tot4"] = f(apc.AveragePrecisionCalculator.ap_at_n(np.array(p4), np.array(a4), n=None, total_num_positives=4))

# C4
p = np.array([[.9, .1, .8, .3], [.2, .7, .6, .1]], dtype=np.float32)
y = np.array([[1, 0, 0, 1], [0, 1, 0, 0]], dtype=np.float32)
out["C4"] = {"p": p.tolist(), "y": y.tolist(), "gap_top2": f(eval_util.calculate_gap(p, y, top_k=2)),
             "hit1": f(eval_util.calculate_hit_at_one(p, y)),
             "perr": f(eval_util.calculate_precision_at_equal_recall_rate(p, y))}

# C5: teacher-style data, RandomState(1234)
rs = np.random.RandomState(1234)
W = rs.randn(32, 4716)
z = rs.randn(256, 32).astype(np.float32)
logit = (z @ W / np.sqrt(32) - 3)
yy = (logit + 0.5 * rs.randn(256, 4716) > 0.2)
pp = (1 / (1 + np.exp(-logit))).astype(np.float32)
out["C5"] = {"seed": 1234, "gap20": f(eval_util.calculate_gap(pp, yy.astype(np.float32), 20)),
             "hit1": f(eval_util.calculate_hit_at_one(pp, yy.astype(np.float32))),
             "perr": f(eval_util.calculate_precision_at_equal_recall_rate(pp, yy.astype(np.float32))),
             "mean_labels": f(yy.sum(1).mean())}

# C6: RandomState(0) uniform
rs = np.random.RandomState(0)
p6 = rs.rand(64, 4716).astype(np.float32)
y6 = (rs.rand(64, 4716) > 0.999)
out["C6"] = {"seed": 0, "gap20": f(eval_util.calculate_gap(p6, y6.astype(np.float32), 20)),
             "hit1": f(eval_util.calculate_hit_at_one(p6, y6.astype(np.float32))),
             "perr": f(eval_util.calculate_precision_at_equal_recall_rate(p6, y6.astype(np.float32)))}

# C7: Dequantize endpoints (formula W/utils.py:35-38 evaluated by hand: utils.py itself needs TF)
out["C7"] = {"deq0": 0 * (4 / 255.0) + (4 / 512.0 - 2), "deq255": 255 * (4 / 255.0) + (4 / 512.0 - 2)}

# C8: small random cases with several top_k, incl. k > classes and rows with no positives
cases = []
for seed, (B, V, k, dens) in enumerate([(5, 11, 3, 0.3), (7, 40, 20, 0.1), (3, 8, 20, 0.5), (16, 100, 5, 0.02)]):
    rs = np.random.RandomState(100 + seed)
    pc = rs.rand(B, V).astype(np.float32)
    yc = (rs.rand(B, V) < dens)
    yc[:, 0] |= (yc.sum(1) == 0)  # PERR divides by #labels: keep >= 1 label per row
    ycf = yc.astype(np.float32)
    cases.append({"seed": 100 + seed, "B": B, "V": V, "k": k, "dens": dens,
                  "gap": f(eval_util.calculate_gap(pc, ycf, k)),
                  "hit1": f(eval_util.calculate_hit_at_one(pc, ycf)),
                  "perr": f(eval_util.calculate_precision_at_equal_recall_rate(pc, ycf))})
out["C8"] = cases

# C9: EvaluationMetrics accumulate/get over 3 batches (W/eval_util.py:167-254)
rs = np.random.RandomState(7)
em = eval_util.EvaluationMetrics(50, 20)
batches = []
for b in range(3):
    pb = rs.rand(8, 50).astype(np.float32)
    yb = (rs.rand(8, 50) < 0.1)
    yb[:, 0] |= (yb.sum(1) == 0)
    lb = float(rs.rand())
    em.accumulate(pb, yb.astype(np.float32), lb)
    batches.append(lb)
res = em.get()
out["C9"] = {"seed": 7, "losses": batches, "avg_hit_at_one": f(res["avg_hit_at_one"]), "avg_perr": f(res["avg_perr"]),
             "avg_loss": f(res["avg_loss"]), "gap": f(res["gap"]), "map": f(np.mean(res["aps"]))}

# C10: MeanAveragePrecisionCalculator (W/mean_average_precision_calculator.py:44-112)
rs = np.random.RandomState(11)
mc = mapc.MeanAveragePrecisionCalculator(6)
pm = rs.rand(12, 6)
ym = (rs.rand(12, 6) < 0.4).astype(np.float64)
mc.accumulate([pm[:, i] for i in range(6)], [ym[:, i] for i in range(6)], [None] * 6)
out["C10"] = {"seed": 11, "aps": [f(a) for a in mc.peek_map_at_n()]}

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics_kat.json")
with open(dst, "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print("wrote", dst)
for k in ("C1_ap", "C2_ap_at_2", "C3_ap_tot4"):
    print(k, out[k])
print("C4", out["C4"]["gap_top2"], out["C4"]["hit1"], out["C4"]["perr"])
print("C5", out["C5"]["gap20"], "C6", out["C6"]["gap20"])
