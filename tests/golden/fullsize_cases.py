"""Full-size cases, one per BASELINE.json config (SURVEY.md 8c last row: "one full-size smoke vector per config stored as
checksums").  Shared by tests/golden/make_fullsize_golden.py (fp64 restatement -> tests/golden/fullsize_kat.json) and
tests/test_gpu_fullsize_golden.py (the HIP path replays them).  Everything is regenerated from seeds with numpy's frozen
RandomState streams; only the checksums (sum, abs-sum, the 20 largest-magnitude entries) live in the fixture.

Sizes: D = 1152, V = 4716, M = 2, F = 300, H = 1024, K = 64 as BASELINE.json names them.  Batches: configs[0] 128 and configs[1]
1024 as quoted; the frame-level configs at 128 videos (the per-GPU batch of the headline for configs[3]; for configs[2] / [4] the
fp64 restatement of 1024 videos needs > 60 GB of autograd tape on this container's host -- the shapes of every weight and of every
per-video tensor are the full ones)."""
import numpy as np

D, V, M, F, H, K, HID, A, CH_L, CH_C = 1152, 4716, 2, 300, 1024, 64, 1024, 8, 3, 128

CONFIGS = ["c0_logistic", "c1_moe", "c2_netvlad", "c3_lstm", "c4_composite_bf16"]
BATCH = {"c0_logistic": 128, "c1_moe": 1024, "c2_netvlad": 128, "c3_lstm": 128, "c4_composite_bf16": 128}
SEED = {"c0_logistic": 900, "c1_moe": 901, "c2_netvlad": 902, "c3_lstm": 903, "c4_composite_bf16": 904}


def _moe_spec(d_in, scope_g="gates", scope_e="experts"):
    return [(scope_g + "/weights", (d_in, V * (M + 1)), "xavier"), (scope_e + "/weights", (d_in, V * M), "xavier"),
            (scope_e + "/biases", (V * M,), "small")]


def param_spec(cfg):
    """[(TF variable name, shape, init)] in creation order.  xavier: U(+-sqrt(6 / (fan_in + fan_out))); small: 0.05 U(-1, 1);
    unit: U(-1, 1) / sqrt(fan_in); sharp: 6 U(-1, 1) (cluster logits of unit-norm frames with a spread of ~3.5: a peaked soft
    assignment, so that the assignment's own gradients are not vanishing)."""
    if cfg == "c0_logistic":
        return [("fully_connected/weights", (D, V), "xavier"), ("fully_connected/biases", (V,), "small")]
    if cfg == "c1_moe":
        return _moe_spec(D)
    nv = [("netvlad/cluster_weights", (D, K), "sharp"), ("netvlad/cluster_biases", (K,), "small"), ("netvlad/centres", (K, D), "unit"),
          ("netvlad/hidden/weights", (K * D, HID), "xavier"), ("netvlad/hidden/biases", (HID,), "small")]
    if cfg == "c2_netvlad":
        return nv + _moe_spec(HID)
    if cfg == "c3_lstm":
        out, d = [], D
        for l in range(2):
            out += [("RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l, (d + H, 4 * H), "xavier"),
                    ("RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l, (4 * H,), "small")]
            d = H
        return out + _moe_spec(4 * H)
    if cfg == "c4_composite_bf16":
        out = nv + [("netvlad/gating/weights", (HID, HID), "xavier"), ("netvlad/gating/biases", (HID,), "small"),
                    ("attention-/weights", (2 * D, A), "unit"), ("attention-/biases", (A,), "small")]
        d = HID + D
        for i in range(CH_L):
            s = "prediction-%d" % i
            out += _moe_spec(d, "gates-" + s, "experts-" + s)
            out += [("relu-%d/weights" % i, (V, CH_C), "xavier"), ("relu-%d/biases" % i, (CH_C,), "small")]
            d += CH_C
        return out + _moe_spec(d, "gates--main", "experts--main")
    raise KeyError(cfg)


def make_params(cfg):
    rs = np.random.RandomState(SEED[cfg])
    P = {}
    for name, shape, init in param_spec(cfg):
        u = (rs.random_sample(shape) * 2.0 - 1.0)
        if init == "xavier":
            u *= np.sqrt(6.0 / (shape[0] + shape[1]))
        elif init == "unit":
            u /= np.sqrt(shape[0])
        elif init == "sharp":
            u *= 6.0
        else:
            u *= 0.05
        P[name] = u.astype(np.float32)
    return P


def make_inputs(cfg, step=0):
    """The batch as the reader hands it over: video-level float features in the dequantised range, frame-level raw uint8 +
    num_frames (ragged, incl. 1 and F); labels ~ Bernoulli(3.4 / V) with at least one per video.  step: which batch of a training
    trajectory (tests/golden/make_trajectory_golden.py); step 0 is the batch of the single-pass fixtures."""
    rs = np.random.RandomState(SEED[cfg] + 5000 + 7919 * step)
    B = BATCH[cfg]
    y = rs.random_sample((B, V)) < 3.4 / V
    y[np.arange(B), rs.randint(0, V, size=B)] = True
    if cfg in ("c0_logistic", "c1_moe"):
        return dict(x=(rs.random_sample((B, D)) * 4.0 - 2.0).astype(np.float32), y=y, nf=None)
    q = rs.randint(0, 256, size=(B, F, D), dtype=np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    return dict(x=q, y=y, nf=nf)


def checksum(a, k=20):
    a = np.asarray(a, dtype=np.float64).ravel()
    idx = np.argsort(-np.abs(a), kind="stable")[:k]
    return {"n": int(a.size), "sum": float(a.sum()), "abs_sum": float(np.abs(a).sum()), "top_idx": [int(i) for i in idx],
            "top_val": [float(a[i]) for i in idx]}
