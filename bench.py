#!/usr/bin/env python
"""bench.py -- training videos/sec of the hot path on N MI355X (BASELINE.json metric).

Headline workload (config.workload): BASELINE.json configs[3] -- the frame-level configuration the metric is quoted on
("training videos/sec on synthetic [B,300,1152] tensors"): LstmModel, 2 x BasicLSTMCell(1024) under dynamic_rnn over
F = 300 frames of raw uint8 [B,300,1152] features, MoE head (2 mixtures, 4716 labels) on [c0|h0|c1|h1], batch 128 PER
GPU, fp32, full training step:
    dequantise + l2-normalise -> hoisted input projections + persistent recurrence (2 layers) -> MoE head ->
    CrossEntropyLoss -> backward (recurrence, dx, dW) -> [RCCL gradient all-reduce for N>1] -> + l2*w -> per-tensor
    clip -> TF-Adam.
Inputs are synthetic and already resident in HBM (a pool of distinct batches cycled through).  One "step" = one such
pass over one batch.  N>1: one process per GPU (torchrun contract; `python bench.py --gpus N` without a torchrun
environment re-launches itself under torch.distributed.run), weak scaling, the RCCL world size is printed.

Extra lines in the same JSON ("extra", rank 0, N=1 only; each with its own per-family hipEvent times and roofline):
  configs[1]  MoeModel (2 mixtures) on video-level [1024,1152] features, fp32 (>= 200 timed steps: the step is ~1 ms)
  configs[2]  NetVLADModel (64 clusters) on raw uint8 [B,300,1152] frames + MoE head, fp32
  configs[3]  in bfloat16 operand mode (a labelled VARIANT, never the headline)
  configs[4]  GatedNetVLADAttentionChainModel in bfloat16 at B = 1024 (its own dtype; the per-GPU share of 8192 on 8 GPUs)
  configs[3]  at per-GPU batch 256 and 512 (batch sweep of the headline configuration, fp32)
Other legs (rank 0):
  roofline     : hipEvent timing of every kernel family inside the library (separate pass after the timed region);
                 algorithmic FLOPs / family time vs the fp32 matrix peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline : the torch-CPU fp32 restatement of the same step (oracle/torch_ref.py, kind "port") on the host cores,
                 bounded sample (B = 32, thread count by probe, per-frame port and its torch.nn.LSTM twin: the faster is
                 `value`); N=1 only.
  reducer      : N > 1 (or --force-reducer): algorithm / bucket / CU-reserve settings and, per rank, the time-out word of the
                 persistent recurrences, placement statistics and a traced bucket timeline of two extra steps.
  gap_at_20    : BASELINE.json's second metric -- GAP@20 on a held-out synthetic teacher shard after 768 training steps
                 of a fresh MoeModel (outside the timed region, ~1.5 s; N=1 only; --no-gap skips it).
"""
import argparse
import ctypes
import gc
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

D_IN, VOCAB, MIX, FRAMES = 1152, 4716, 2, 300
LSTM_H, LSTM_L = 1024, 2
PEAK_F32_MATRIX_TFLOPS = 157.3
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md); only used by bf16 VARIANT lines
PEAK_HBM_GBS = 8000.0
FAMILIES = ["gemm", "moe_fused", "elementwise", "optimizer", "lstm_recurrence", "netvlad", "lstm_recurrence_bwd", "gemm_x3", "gemm_x1x3",
            "vlad_rows", "vlad_cols",          # the streaming kernels of "netvlad", timed inside it, bytes declared
            "netvlad_fwd",                     # the whole forward pooling call with SURVEY.md 8(d)'s bytes (frames once + parameters)
            "gemm_h2",                         # fp32 products as three f16 MFMA products of two-plane half images (round 5)
            "gemm_h1x2"]                       # ... with a one-plane exact operand (uint8 frames minus 128): two products
X3_PRODUCTS = 6.0                    # bf16 MFMA products per fp32 product in csrc/gemm_x3.hip


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["lstm", "moe", "netvlad"], default="lstm",
                    help="headline workload; default = BASELINE configs[3] (the frame-level config the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the config's: lstm 128, moe 1024, netvlad 1024)")
    ap.add_argument("--pool", type=int, default=0, help="distinct synthetic batches resident in HBM (0 = per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[1] / configs[2] / bf16-variant extra lines")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="lower bound of the cpu_baseline sample (it also runs >= --cpu-steps steps)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed steps of the cpu_baseline leg at least (SURVEY.md 8d: >= 10)")
    ap.add_argument("--cpu-max-seconds", type=float, default=120.0, help="hard wall budget of the cpu_baseline sample, shared by its two "
                    "implementations (it stops early and reports the steps it did)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the reference's arithmetic (the headline).  bf16 = bf16 MFMA operands, fp32 accumulate / "
                         "master weights / Adam: a separately labelled line, never the headline.")
    ap.add_argument("--no-gap", action="store_true", help="skip the GAP@20 leg (BASELINE.json's second metric)")
    ap.add_argument("--force-reducer", action="store_true", help="exercise the RCCL reducer even at world size 1 (test aid)")
    # data-parallel knobs (also YT8M_DP_ALGO / YT8M_DP_RESERVED_CUS / YT8M_DP_LAYER_BUCKETS); recorded in the line's "reducer" object
    ap.add_argument("--dp-algo", choices=["allreduce", "rs_ag"], default=None, help="gradient collective per bucket")
    ap.add_argument("--dp-reserved-cus", type=int, default=None, help="CUs the persistent recurrences leave to RCCL's kernels")
    ap.add_argument("--dp-layer-buckets", type=int, choices=[0, 1], default=None, help="report the LSTM layers' gradients separately")
    ap.add_argument("--dp-bucket-mb", type=int, default=None, help="all-reduce bucket threshold in MiB (default 32)")
    a = ap.parse_args()
    for flag, env in ((a.dp_algo, "YT8M_DP_ALGO"), (a.dp_reserved_cus, "YT8M_DP_RESERVED_CUS"), (a.dp_layer_buckets, "YT8M_DP_LAYER_BUCKETS")):
        if flag is not None:                             # read at import time by parallel.py / seq_ops.py, inherited by the ranks
            os.environ[env] = str(flag)
    return a


# ---- self-launch: `python bench.py --gpus N` must run N ranks even without torchrun -------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def maybe_relaunch(a):
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


# ---- workloads ---------------------------------------------------------------------------------------------------------
def make_video_pool(n, B, dev, seed):
    """Video-level synthetic inputs: a wide spread in the dequantised range [-2, 2]; labels ~ Bernoulli(3.4/V)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.rand((B, D_IN), device=dev, generator=gen) * 4.0 - 2.0
        y = torch.rand((B, VOCAB), device=dev, generator=gen) < (3.4 / VOCAB)
        out.append((x, y, None))
    return out


def make_frame_pool(n, B, dev, seed):
    """Frame-level synthetic inputs: raw uint8 [B,300,1152] exactly as the reader hands them over (W/readers.py:159-187),
    every video 300 frames long (no work is skipped by the num_frames masks)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = []
    for _ in range(n):
        x = torch.randint(0, 256, (B, FRAMES, D_IN), device=dev, generator=gen, dtype=torch.uint8)
        y = torch.rand((B, VOCAB), device=dev, generator=gen) < (3.4 / VOCAB)
        nf = torch.full((B,), FRAMES, device=dev, dtype=torch.int32)
        out.append((x, y, nf))
    return out


def lstm_flops(B):
    """Algorithmic FLOPs of one configs[3] training step (per family)."""
    H, L, F, D = LSTM_H, LSTM_L, FRAMES, D_IN
    proj_fwd = 2.0 * F * B * (D * 4 * H + (L - 1) * H * 4 * H)
    dw = 2.0 * F * B * ((D + H) * 4 * H + (L - 1) * (H + H) * 4 * H)
    dx = 2.0 * F * B * (L - 1) * 4 * H * H
    S = 2 * L * H
    head = 3 * 2.0 * B * S * VOCAB * (2 * MIX + 1)
    rec = 2 * L * 2.0 * F * B * H * 4 * H             # forward + backward recurrent products
    return {"gemm": proj_fwd + dw + dx + head, "lstm_recurrence": rec / 2, "lstm_recurrence_bwd": rec / 2}


def moe_flops(B):
    return {"gemm": 2.0 * 2.0 * B * D_IN * VOCAB * (2 * MIX + 1)}


def netvlad_flops(B, K=64, hidden=1024):
    F, D = FRAMES, D_IN
    pool = 2 * 2 * 2.0 * B * F * D * K                 # assignment + aggregation, forward and backward
    fc = 3 * 2.0 * B * (K * D) * hidden                # hidden FC fwd, dW, dx
    head = 3 * 2.0 * B * hidden * VOCAB * (2 * MIX + 1)
    return {"netvlad": pool, "gemm": fc + head}


WORKLOADS = {
    "lstm": dict(name="BASELINE configs[3]: LstmModel (2 x BasicLSTMCell(1024), dynamic_rnn over F=300) on raw uint8 "
                      "[B,300,1152] frames + MoeModel head (2 mixtures, V=4716)",
                 batch=128, pool=4, frame=True, flops=lstm_flops),
    "moe": dict(name="BASELINE configs[1]: MoeModel (2 mixtures) on video-level features, D=1152, V=4716",
                batch=1024, pool=8, frame=False, flops=moe_flops),
    "netvlad": dict(name="BASELINE configs[2]: NetVLADModel (64 clusters, hidden 1024) on raw uint8 [B,300,1152] frames + "
                         "MoeModel head (2 mixtures, V=4716)",
                    batch=1024, pool=2, frame=True, flops=netvlad_flops),
    # extra line only (bf16, the config's own dtype; per-GPU share of the global batch of 8192 on 8 GPUs)
    "composite": dict(name="BASELINE configs[4]: GatedNetVLADAttentionChainModel (gated NetVLAD 64 clusters + attention pooling + 3-layer "
                           "chained MoE, multitask loss) on raw uint8 [B,300,1152] frames",
                      batch=1024, pool=1, frame=True, flops=lambda B: {}),
}


def build(workload, B, world, rank, dev, reducer, bf16):
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph
    FLAGS.reset()
    if bf16:
        FLAGS.compute_dtype = "bfloat16"
    if os.environ.get("YT8M_FUSED_HEAD_LOSS") == "0":     # A/B aid
        FLAGS.fused_head_loss = False
    for kv in os.environ.get("YT8M_SET", "").split(","):  # e.g. YT8M_SET=lstm_pipeline_chunks=1
        if "=" in kv:
            k, v = kv.split("=", 1)
            cur = getattr(FLAGS, k)
            setattr(FLAGS, k, (v == "1") if isinstance(cur, bool) else type(cur)(v))
    multitask = None
    if workload == "composite":                           # SURVEY.md App. B / tools/model_bench.py "config5"
        FLAGS.deep_chain_layers, FLAGS.deep_chain_relu_cells, FLAGS.support_type = 3, 128, ",".join(["label"] * 3)
        multitask = True
    model = {"lstm": flm.LstmModel, "moe": vlm.MoeModel, "netvlad": flm.NetVLADModel,
             "composite": flm.GatedNetVLADAttentionChainModel}[workload]()
    g = reset_default_graph(device=dev, seed=0)
    import yt8m_amd.losses as losses
    tg = train.TrainGraph(model, batch_size=B * world, graph=g, reducer=reducer, multitask=multitask,
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss() if multitask else None)
    w = make_pool(workload, B, dev, rank)
    return g, tg, w


def make_pool(workload, B, dev, rank, n=None):
    cfg = WORKLOADS[workload]
    n = n or cfg["pool"]
    return (make_frame_pool if cfg["frame"] else make_video_pool)(n, B, dev, seed=1234 + rank)


def family_times(lib, steps):
    n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
    fam = {}
    fl = ctypes.c_double(0.0)
    for fid, name in enumerate(FAMILIES):
        lib.yt8m_prof_get(fid, ctypes.byref(n), ctypes.byref(ms))
        lib.yt8m_prof_get_flops(fid, ctypes.byref(fl))
        if n.value:
            fam[name] = {"launches_per_step": n.value / float(steps), "ms_per_step": ms.value / steps,
                         "avg_launch_ms": ms.value / n.value}
            if fl.value > 0:
                fam[name]["declared_flops_per_step"] = fl.value / steps       # counted by the library at launch time
            lib.yt8m_prof_get_bytes(fid, ctypes.byref(fl))
            if fl.value > 0:
                fam[name]["declared_bytes_per_step"] = fl.value / steps       # algorithmic HBM bytes, same mechanism
    return fam


def family_peak(name, bf16, fwd_x3=False, bwd_f16=False):
    """(chip-level peak TFLOP/s, what it is) of the matrix pipe a family's kernels ISSUE on, in the units its algorithmic FLOPs are
    counted in (fp32-equivalent FLOPs: a family that spends k bf16 / f16 MFMA products per fp32 product has peak 2500 / k).
    VERDICT r2: never price a bf16-pipe kernel against the fp32 peak, nor the other way round."""
    bfp = PEAK_BF16_MATRIX_TFLOPS
    if name == "gemm_x3":
        return bfp / X3_PRODUCTS, ("bf16 MFMA pipe, fp32-equivalent: dense peak %.0f / %d products per fp32 product "
                                   "(v_mfma_f32_32x32x16_bf16 on three-plane split operands)" % (bfp, X3_PRODUCTS))
    if name == "gemm_h2":
        return bfp / 3.0, ("f16 MFMA pipe, fp32-equivalent: dense peak %.0f / 3 products per fp32 product (v_mfma_f32_32x32x16_f16 on "
                           "two-plane half splits of both operands: hi hi + hi lo + lo hi)" % bfp)
    if name == "gemm_h1x2":
        return bfp / 2.0, ("f16 MFMA pipe, fp32-equivalent: dense peak %.0f / 2 products per fp32 product (one exact half plane -- uint8 "
                           "frames minus 128 -- against a two-plane half split)" % bfp)
    if name == "gemm_x1x3":
        return bfp / 3.0, ("bf16 MFMA pipe, fp32-equivalent: dense peak %.0f / 3 products per fp32 product (one exact bf16 plane "
                           "-- uint8 frames minus 128 -- against a three-plane split operand)" % bfp)
    if name == "netvlad":
        if bf16:
            return bfp, "f16 MFMA pipe, one product per element pair (v_mfma_f32_16x16x32_f16, nsplit = 1): dense peak"
        return bfp / 2.0, ("f16 MFMA pipe, fp32-equivalent: dense peak %.0f / 2 products per fp32 product (exact f16 (q - 128) "
                           "against the f16 hi + lo split of the fp32 operand)" % bfp)
    if name == "lstm_recurrence":
        if bf16:
            return bfp, "bf16 MFMA pipe (lstm_step_fwd16_kernel: bf16 h / W_h operands, one product): dense peak"
        if fwd_x3 == "f16":
            return bfp / 3.0, ("f16 MFMA pipe, fp32-equivalent: dense peak %.0f / 3 products per fp32 product (v_mfma_f32_16x16x32_f16 on "
                               "two-plane half splits of h and W_h; the kernel is paced by its publish -> fetch chain, not by this pipe)" % bfp)
        if fwd_x3:
            return bfp / X3_PRODUCTS, ("bf16 MFMA pipe, fp32-equivalent: dense peak %.0f / %d products per fp32 product "
                                       "(v_mfma_f32_16x16x32_bf16 on three-plane splits of h and W_h)" % (bfp, X3_PRODUCTS))
    if name == "lstm_recurrence_bwd" and bwd_f16 and not bf16:
        return bfp / 3.0, ("f16 MFMA pipe, fp32-equivalent: dense peak %.0f / 3 products per fp32 product (v_mfma_f32_16x16x32_f16 on two-plane "
                           "half splits of dz and W_h^T; the kernel is bound by the delivery of the exchanged dz -- 1 MiB per workgroup and "
                           "step through one CU's L2 port -- not by this pipe: roofline.exchange)" % bfp)
    if name == "lstm_recurrence_bwd" and bf16:
        return bfp, "bf16 MFMA pipe (lstm_step_bwd16_kernel: bf16 dz / W_h operands, one product): dense peak"
    if bf16 and name == "gemm":
        return bfp, "bf16 MFMA pipe: dense peak"
    return PEAK_F32_MATRIX_TFLOPS, "fp32 MFMA pipe (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): dense peak of the whole chip"


def fwd_form(lib, B, H):
    """False (fp32 pipe), True (six bf16 products) or "f16" (three f16 products: the native stack's default with the h2 products)."""
    if os.environ.get("YT8M_PERSIST_STEP_IMAGES", "1") == "0" or not lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, H):
        return False
    h2 = os.environ.get("YT8M_STACK_H2", "1") != "0"
    if os.environ.get("YT8M_STACK_H2_RECUR_FWD", "1" if h2 else "0") != "0" and os.environ.get("YT8M_PERSIST_FWD_H2", "1") != "0":
        return "f16"
    return True


def bwd_on_f16_pipe(lib, B, H):
    """True when the native stack's backward recurrences of this shape run the three-f16-product form (yt8m_lstm_persist_bwd_h2 /
    _ex with a weight scale word; knobs YT8M_STACK_H2_RECUR, YT8M_PERSIST_BWD_H2)."""
    return (os.environ.get("YT8M_STACK_H2_RECUR", "1") != "0" and os.environ.get("YT8M_PERSIST_BWD_H2", "1") != "0" and
            os.environ.get("YT8M_PERSIST_STEP_IMAGES", "1") != "0" and bool(lib.yt8m_lstm_persist_bwd_on_f16_pipe(B, H)))


def exchange_view(row, B, H, cus, f16, pair=True):
    """What bounds the backward recurrence in its f16 form (profiles/r6_pmc_recur_tcc.txt): a workgroup (one per CU) draws the dz its K
    range needs for its 64 rows through the CU's vector-memory window -- the L1 keeps ~64 line requests (128 B each) in flight, so a CU
    receives 64 x 128 B per mean request latency.  The counters give 277 cycles per request for this kernel (82 % L2 hits, 8 % fabric reads
    at ~805 cycles); with every request an L2 hit (~230 cycles) the window delivers 64 x 128 / 230 = 35.6 B/clk: the bound quoted here.
    (Earlier rounds quoted the L2 port's 64 B/clk, which no CU reaches at these latencies.)  Round 6's K-split pairs halve the bytes a
    workgroup draws per step (pair=True): achieved is then measured on half of 64 rows x 4H x 4 B."""
    if not f16 or not row:
        return None
    per_wg = 64.0 * 4 * H * 4 * (0.5 if pair else 1.0)
    steps_per_launch = row["algorithmic_flops_per_launch"] / (2.0 * B * H * 4 * H)
    us_per_step = row["avg_launch_ms"] * 1e3 / max(steps_per_launch, 1e-9)
    ach = per_wg / (us_per_step * 1e-6) / 1e9
    peak = 64.0 * 128.0 / 230.0 * 2.1                      # GB/s per CU: 64 requests x 128 B per 230-cycle L2 hit at ~2.1 GHz
    return {"bytes_per_workgroup_and_step": per_wg, "us_per_step": us_per_step, "achieved_GBps_per_cu": ach, "peak_GBps_per_cu": peak,
            "frac": ach / peak, "workgroups": cus, "k_split_pairs": bool(pair),
            "is": "dz_t drawn by each of the launch's workgroups per time step over the launch's average step time, against what one CU's "
                  "64-request vector-memory window delivers when every request is an L2 hit (64 x 128 B / 230 cycles at 2.1 GHz; "
                  "profiles/r6_pmc_recur_tcc.txt): the bound this kernel runs against"}


def roofline_from(fam, flops, bf16, extra_note=None, bwd_cus=None, fwd_x3=False, step_ms=None, bwd_f16=False):
    """Dominant family = the one with the largest hipEvent time among the MFMA families that have an algorithmic FLOP count
    (declared by the library at launch time for the GEMM and recurrence entry points, else the workload's formula).
    achieved = algorithmic FLOPs of that family per step / its time per step (= FLOPs per launch / average launch).
    `frac` is ALWAYS against the whole chip's peak of the pipe the family issues on; a family whose launches occupy only part
    of the chip (the half-chip backward recurrence) also reports `frac_of_occupied_cus` next to it.
    `blended_bound`: sum over the MFMA families of (algorithmic FLOPs / chip peak of their pipe) = the time the step's matrix
    work would take with every pipe at its peak and nothing overlapped, against the measured wall time of a step."""
    rows = {}
    for name, v in fam.items():
        if name in ("vlad_rows", "vlad_cols", "netvlad_fwd"):        # sub-scopes of "netvlad": reported under roofline.hbm
            continue
        f = v.get("declared_flops_per_step") or flops.get(name)
        if f and v["ms_per_step"] > 0:
            peak, what = family_peak(name, bf16, fwd_x3, bwd_f16)
            ach = f / (v["ms_per_step"] * 1e-3) / 1e12
            rows[name] = {"achieved": ach, "peak": peak, "peak_is": what, "frac": ach / peak, "ms_per_step": v["ms_per_step"],
                          "launches_per_step": v["launches_per_step"], "avg_launch_ms": v["avg_launch_ms"],
                          "algorithmic_flops_per_step": f,
                          "algorithmic_flops_per_launch": f / max(v["launches_per_step"], 1e-9)}
            if name == "lstm_recurrence_bwd" and bwd_cus and not bf16 and bwd_cus < 256:
                rows[name]["occupied_cus"] = bwd_cus
                rows[name]["frac_of_occupied_cus"] = ach / (peak * bwd_cus / 256.0)
                rows[name]["occupancy_note"] = ("launched on %d of 256 CUs (two of them run side by side for part of the backward pass, "
                                                "the weight-gradient GEMMs take the other CUs): `frac` is against the WHOLE chip" % bwd_cus)
    if not rows:
        return None
    dom = max(rows, key=lambda k: rows[k]["ms_per_step"])
    r = rows[dom]
    roof = {"bound": "mfma", "kernel": dom, "achieved": r["achieved"], "peak": r["peak"], "peak_is": r["peak_is"], "unit": "TFLOP/s",
            "frac": r["frac"], "traffic": None, "traffic_source": None,
            "launches_per_step": r["launches_per_step"], "avg_launch_ms": r["avg_launch_ms"],
            "algorithmic_flops_per_launch": r["algorithmic_flops_per_launch"],
            "families": rows,
            "other_families": {k: v for k, v in fam.items() if k not in rows and k not in ("vlad_rows", "vlad_cols", "netvlad_fwd")}}
    for k in ("occupied_cus", "frac_of_occupied_cus", "occupancy_note"):
        if k in r:
            roof[k] = r[k]
    bound_ms = sum(v["algorithmic_flops_per_step"] / (v["peak"] * 1e12) * 1e3 for v in rows.values())
    roof["blended_bound"] = {"ms_per_step": bound_ms,
                             "is": "sum over the MFMA families of algorithmic FLOPs / whole-chip peak of the pipe each issues on",
                             "frac": (bound_ms / step_ms) if step_ms else None}
    if extra_note:
        roof["note"] = extra_note
    return roof


def timed_run(tg, pool, steps, warmup, world, dev, dist):
    """The measured region, and nothing else: W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier +
    torch.cuda.synchronize() on both sides; max over ranks.  (The per-step / host-profile diagnostics that used to live here are
    tools/bench_diag.py: they wrap this function's `run` from the outside.)"""

    def run(k, base):
        for i in range(k):
            x, y, nf = pool[(base + i) % len(pool)]
            tg.step(x, y, nf)

    run(max(warmup, 1), 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, run


def profile_pass(lib, run, steps, rank):
    if rank == 0:
        lib.yt8m_prof_reset()
        lib.yt8m_prof_enable(1)
    run(steps, 0)                      # every rank runs the same extra steps (the all-reduce is collective)
    torch.cuda.synchronize()
    fam = None
    if rank == 0:
        lib.yt8m_prof_enable(0)
        fam = family_times(lib, steps)
    return fam


def netvlad_hbm(fam, bf16):
    """The two streaming kernels of the fused NetVLAD pooling against the HBM roof ("NetVLAD: report both"):
    achieved = the algorithmic bytes the library declares per launch (uint8 frames once + the transposed assignment + the
    aggregate / partials) / hipEvent time; next to it the same launches as a fraction of the f16 MFMA pipe."""
    out = {"peak_GBps": PEAK_HBM_GBS, "kernels": {}}
    peak, what = family_peak("netvlad", bf16)
    for k in ("vlad_rows", "vlad_cols"):
        v = fam[k]
        by, fl = v.get("declared_bytes_per_step", 0.0), v.get("declared_flops_per_step", 0.0)
        gbps = by / (v["ms_per_step"] * 1e-3) / 1e9
        out["kernels"][k] = {"launches_per_step": v["launches_per_step"], "avg_launch_us": v["avg_launch_ms"] * 1e3,
                             "algorithmic_bytes_per_launch": by / max(v["launches_per_step"], 1e-9), "achieved_GBps": gbps,
                             "frac_of_hbm": gbps / PEAK_HBM_GBS,
                             "mfma_TFLOPs": fl / (v["ms_per_step"] * 1e-3) / 1e12, "frac_of_mfma_pipe": fl / (v["ms_per_step"] * 1e-3) / 1e12 / peak}
    fw = fam.get("netvlad_fwd")
    if fw:                     # the forward pooling against the HBM roof on SURVEY.md 8(d)'s bytes (345 600 B / video + params)
        by = fw.get("declared_bytes_per_step", 0.0)
        gb = by / (fw["ms_per_step"] * 1e-3) / 1e9
        out["forward_8d"] = {"us": fw["ms_per_step"] * 1e3, "bytes_8d": by, "achieved_GBps": gb, "frac_of_hbm_8d_bytes": gb / PEAK_HBM_GBS,
                             "is": "whole yt8m_netvlad_fwd_u8 call (pack + rows + cols, or the single-pass kernel) / (uint8 frames once + W_c, b_c)"}
    out["mfma_pipe"] = what
    out["north_star_note"] = ("BASELINE.json asks for >= 70 % of the bf16 MFMA roofline on the assignment GEMM.  That GEMM does 128 FLOP per "
                              "uint8 input byte ([B*300,1152] x [1152,64]), below the ~312 FLOP/B ridge of a 2.5 PFLOP/s pipe over 8 TB/s: "
                              "reading the frames alone caps it at <= 41 % of the MFMA peak, assignment + aggregation fused in one pass "
                              "over the frames at <= 82 % (SURVEY.md section 7).  The bound that applies is HBM: frac_of_hbm is the "
                              "roofline fraction of these kernels, frac_of_mfma_pipe is printed for completeness.")
    return out


def extra_line(workload, dev, lib, bf16=False, steps=None, warmup=None, batch=None, tag=""):
    """One more configuration, rank 0 / N=1 only: its own timed region (>= 200 steps when a step is < 5 ms) + family times."""
    cfg = WORKLOADS[workload]
    B = batch or cfg["batch"]
    g, tg, pool = build(workload, B, 1, 0, dev, None, bf16)
    x, y, nf = pool[0]
    for _ in range(2):
        tg.step(x, y, nf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tg.step(x, y, nf)
    torch.cuda.synchronize()
    probe = time.perf_counter() - t0
    steps = steps or (200 if probe < 5e-3 else max(10, min(50, int(1.0 / probe))))
    el, run = timed_run(tg, pool, steps, warmup or 3, 1, dev, None)
    fam = profile_pass(lib, run, min(steps, 10), 0)
    bwd_cus = None
    fwd_x3 = False
    if workload == "lstm" and not bf16:
        if lib.yt8m_lstm_persist_bwd_supported(B, LSTM_H):
            bwd_cus = int(os.environ.get("YT8M_PERSIST_CUS_BWD", "128"))
        fwd_x3 = fwd_form(lib, B, LSTM_H)
    bwd_f16 = workload == "lstm" and not bf16 and bwd_on_f16_pipe(lib, B, LSTM_H)
    roof = roofline_from(fam, cfg["flops"](B), bf16, bwd_cus=bwd_cus, fwd_x3=fwd_x3, step_ms=el / steps * 1e3, bwd_f16=bwd_f16)
    if roof and bwd_f16:
        roof["exchange"] = exchange_view(roof.get("families", {}).get("lstm_recurrence_bwd"), B, LSTM_H, bwd_cus, True)
    if roof is None and fam:
        roof = {"families": {}, "other_families": fam}
    if workload == "netvlad" and fam and "vlad_rows" in fam:
        roof["hbm"] = netvlad_hbm(fam, bf16)
    del tg, g, pool
    gc.collect()                         # Graph <-> Variable <-> WeightImages are reference cycles: release the arenas and images now
    torch.cuda.empty_cache()
    return {"workload": cfg["name"] + (" -- bf16-operand VARIANT" if bf16 else ", fp32") + tag, "dtype": "bf16" if bf16 else "f32",
            "per_gpu_batch": B, "steps": steps, "ms_per_step": el / steps * 1e3, "value": steps * B / el, "unit": "videos/s",
            "roofline": roof}


def gap_leg(dev, B=1024, train_steps=768, heldout=16384, signal=3.0):
    """BASELINE.json's second metric, outside the timed region: a fresh MoeModel (M = 2, D = 1152, V = 4716) is
    trained for `train_steps` steps of B videos on a synthetic teacher shard (fixed W_t ~ N(0, 1/sqrt(D)), logit = x.W_t * 3
    - 3 + 0.5 N(0,1), threshold at ~3.4 positives per video) and evaluated with GAP@20 on a disjoint held-out shard; the
    per-video top-20 runs on the device.  (tests/test_gpu_models.py::test_gap_on_heldout_shard_matches_cpu_training checks
    the same procedure against the CPU oracle's training at a reduced size: |dGAP| < 0.001.)"""
    import yt8m_amd.eval_util as eval_util
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph
    FLAGS.reset()
    gen = torch.Generator(device=dev).manual_seed(4242)
    Wt = torch.randn((D_IN, VOCAB), device=dev, generator=gen) / D_IN ** 0.5

    def batch():
        x = torch.rand((B, D_IN), device=dev, generator=gen) * 4.0 - 2.0
        logit = (x @ Wt) * signal - 3.0 + 0.5 * torch.randn((B, VOCAB), device=dev, generator=gen)
        return x, logit

    x0, l0 = batch()
    tau = torch.quantile(l0.flatten()[:1 << 22], 1.0 - 3.4 / VOCAB)
    g = reset_default_graph(device=dev, seed=1)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)

    def evaluate(n):
        em = eval_util.EvaluationMetrics(VOCAB, 20)
        egen_state = gen.get_state()
        gen.manual_seed(99991)                                   # disjoint seed = held-out shard
        for _ in range(n // B):
            x, logit = batch()
            em.accumulate_device(tg.predict(x, vocab_size=VOCAB), logit > tau, 0.0)
        gen.set_state(egen_state)
        return em.get()

    tg.forward(x0, l0 > tau)
    tg.ensure_finalized() if hasattr(tg, "ensure_finalized") else g.finalize()
    before = evaluate(min(heldout, 4 * B))["gap"]
    pos = 0.0
    for _ in range(train_steps):
        x, logit = batch()
        y = logit > tau
        pos += float(y.float().sum(1).mean())
        tg.step(x, y)
    m = evaluate(heldout)
    state = (Wt.cpu(), float(tau), {k: v.data.detach().cpu().clone() for k, v in g.vars.items()})
    return {"_state": state, "value": m["gap"], "hit_at_one": m["avg_hit_at_one"], "perr": m.get("avg_perr"), "untrained": before,
            "train_steps": train_steps, "batch": B, "heldout_videos": (heldout // B) * B,
            "positives_per_video": pos / train_steps, "data": "synthetic teacher shard (MoeModel, configs[1])"}


def gap_twin(dev, D_=64, V_=300, M_=2, B_=256, steps=60, held=2048, state=None):
    """The acceptance form of the north-star's second target ("GAP@20 within 0.001 of the reference on a held-out synthetic
    shard"), at a size the CPU port trains in seconds: the SAME MoeModel from the SAME initial weights on the SAME teacher-shard
    batches through the HIP path and through the torch-CPU restatement (the checker: oracle/torch_ref.py, TF1 is not runnable);
    GAP@20 of both on a disjoint held-out shard."""
    import numpy as np
    import yt8m_amd.eval_util as eval_util
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph
    from oracle import torch_ref
    FLAGS.reset()
    gen = torch.Generator().manual_seed(7)
    Wt = torch.randn(D_, V_, generator=gen) / D_ ** 0.5
    tau0 = None
    if state is not None:                                # continue from the GAP leg's trained model on ITS teacher shard
        Wt, tau0, init = state

    def shard(n, seed):
        g_ = torch.Generator().manual_seed(seed)
        x = torch.rand(n, D_, generator=g_) * 4.0 - 2.0
        logit = x @ Wt * 3.0 - 3.0 + 0.5 * torch.randn(n, V_, generator=g_)
        tau = tau0 if tau0 is not None else torch.quantile(logit.flatten()[:200000], 1.0 - 3.4 / V_)
        return x, logit > tau

    xtr, ytr = shard(B_ * steps, 11)
    xho, yho = shard(held, 12)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = torch_ref.MoeTrainStepCPU(D=D_, V=V_, M=M_, batch_size=B_, dtype=torch.float32, seed=3)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B_, graph=g)
    tg.forward(xtr[:B_].to(dev), ytr[:B_].to(dev))
    tg.ensure_finalized()
    if state is not None:
        with torch.no_grad():
            for k, v in cpu.P.items():
                v.copy_(init[k].view(v.shape))
    for k, v in cpu.P.items():
        g.vars[k].data.copy_(v.detach().to(dev).view(g.vars[k].data.shape))
    for i in range(steps):
        xb, yb = xtr[i * B_:(i + 1) * B_], ytr[i * B_:(i + 1) * B_]
        tg.step(xb.to(dev), yb.to(dev))
        cpu.step(xb, yb)
    ph = tg.predict(xho.to(dev), vocab_size=V_).cpu().numpy()
    with torch.no_grad():
        pc = torch_ref.moe(torch_ref.l2_normalize(xho, 1), cpu.P["gates/weights"], cpu.P["experts/weights"], cpu.P["experts/biases"],
                           M_).numpy()
    yh = yho.numpy().astype(np.float32)
    gh, gc = eval_util.calculate_gap(ph, yh, 20), eval_util.calculate_gap(pc, yh, 20)
    return {"gap_hip": gh, "gap_cpu_port": gc, "abs_diff": abs(gh - gc), "target": 0.001, "within_target": bool(abs(gh - gc) < 1e-3),
            "config": "MoeModel D=%d V=%d M=%d, %d steps x %d videos, held-out %d videos, same initial weights and batches%s" %
                      (D_, V_, M_, steps, B_, held, "" if state is None else
                       " -- continued from the GAP leg's trained weights (fresh Adam state on both sides) on its teacher shard")}


def gap_twin_frames(dev, D_=64, H_=256, L_=2, F_=40, V_=300, M_=2, B_=128, steps=100, held=2048, base_lr=0.001):
    """The same acceptance check on a FRAME-LEVEL model (W/eval_util.py:102-120): LstmModel (L_ x BasicLSTMCell(H_)
    under dynamic_rnn over <= F_ ragged frames of raw uint8 features + MoE head) trained from the SAME initial weights on the
    SAME batches by the HIP path and by the torch-CPU restatement (oracle/torch_ref.LstmTrainStepCPU: the checker), GAP@20 of
    both on a disjoint held-out shard.  The teacher labels a video from its base frame (the frames are the base + uniform byte
    noise), so the recurrent stack has something to integrate.  base_learning_rate 0.001 on both sides: at the reference's default
    0.01 this small model saturates to p = 0 within ten steps on either implementation and there is no ranking left to compare."""
    import numpy as np
    import yt8m_amd.eval_util as eval_util
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.train as train
    from yt8m_amd.flags import FLAGS
    from yt8m_amd.variables import reset_default_graph
    from oracle import torch_ref
    FLAGS.reset()
    FLAGS.lstm_cells, FLAGS.lstm_layers = H_, L_
    gen = torch.Generator().manual_seed(17)
    Wt = torch.randn(D_, V_, generator=gen) / D_ ** 0.5
    tau = [None]

    def shard(n, seed):
        g_ = torch.Generator().manual_seed(seed)
        base = torch.randint(0, 256, (n, 1, D_), generator=g_)
        q = (base + torch.randint(-12, 13, (n, F_, D_), generator=g_)).clamp_(0, 255).to(torch.uint8)
        nf = torch.randint(F_ // 2, F_ + 1, (n,), generator=g_, dtype=torch.int32)
        q = q * (torch.arange(F_)[None, :, None] < nf[:, None, None]).to(torch.uint8)      # the reader pads with zero BYTES
        logit = torch_ref.dequantize(base[:, 0].to(torch.uint8)) @ Wt * 4.0 - 3.0 + 0.5 * torch.randn(n, V_, generator=g_)
        if tau[0] is None:
            tau[0] = torch.quantile(logit.flatten()[:200000], 1.0 - 3.4 / V_)
        return q, nf, logit > tau[0]

    qtr, ntr, ytr = shard(B_ * steps, 21)
    qho, nho, yho = shard(held, 22)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cpu = torch_ref.LstmTrainStepCPU(D=D_, H=H_, L=L_, V=V_, M=M_, batch_size=B_, dtype=torch.float32, seed=5, base_lr=base_lr)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B_, graph=g, base_learning_rate=base_lr)
    tg.forward(qtr[:B_].to(dev), ytr[:B_].to(dev), ntr[:B_].to(dev))
    tg.ensure_finalized()
    assert set(cpu.P) == set(v for v in g.vars), (sorted(cpu.P), sorted(g.vars))
    for k, v in cpu.P.items():
        g.vars[k].data.copy_(v.detach().to(dev).view(g.vars[k].data.shape))
    worst = 0.0
    for i in range(steps):
        sl = slice(i * B_, (i + 1) * B_)
        lh = float(tg.step(qtr[sl].to(dev), ytr[sl].to(dev), ntr[sl].to(dev))["loss"])
        lc = float(cpu.step(qtr[sl], ntr[sl], ytr[sl])[0])
        worst = max(worst, abs(lh - lc) / max(abs(lc), 1e-30))
    ph = torch.cat([tg.predict(qho[i:i + B_].to(dev), nho[i:i + B_].to(dev), vocab_size=V_).cpu() for i in range(0, held, B_)]).numpy()
    with torch.no_grad():
        x = torch_ref.l2_normalize(torch_ref.dequantize(qho), 2)
        layers = [(cpu.P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l],
                   cpu.P["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l]) for l in range(L_)]
        pc = torch_ref.moe(torch_ref.lstm_model_state(x, nho, layers), cpu.P["gates/weights"], cpu.P["experts/weights"],
                           cpu.P["experts/biases"], M_).numpy()
    yh = yho.numpy().astype(np.float32)
    gh, gc = eval_util.calculate_gap(ph, yh, 20), eval_util.calculate_gap(pc, yh, 20)
    FLAGS.reset()
    return {"gap_hip": gh, "gap_cpu_port": gc, "abs_diff": abs(gh - gc), "target": 0.001,
            "within_target": bool(abs(gh - gc) < 1e-3 and gc > 0.05),          # ... of a model that LEARNED the shard (untrained: ~0.001)
            "max_rel_loss_diff_while_training": worst, "max_abs_prediction_diff": float(np.abs(ph - pc).max()),
            "config": "LstmModel %d x BasicLSTMCell(%d) over <= %d ragged uint8 frames (D=%d) + MoE head V=%d M=%d, %d steps x %d videos, "
                      "base_learning_rate %g, held-out %d videos, same initial weights and batches" % (L_, H_, F_, D_, V_, M_, steps, B_, base_lr, held)}


def _pick_threads(probe_fn, candidates=None):
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best, best_t = None, None
    for cores in sorted({c for c in (candidates or (usable, 64, 32, 16)) if 1 <= c <= usable}, reverse=True):
        torch.set_num_threads(cores)
        probe_fn()
        t0 = time.perf_counter()
        probe_fn()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = cores, dt
    torch.set_num_threads(best)
    return best, usable


def cpu_baseline(workload, seconds, min_steps=10, max_seconds=120.0):
    """Times the torch-CPU fp32 restatement of the same training step on the host cores (a reported baseline, not a target; TF1
    itself is not runnable here).  Bounded sample.  Frame-level step: B = 32 videos x 300 frames, thread count
    chosen by a probe (a 12-frame cut of the same step at every candidate count), and TWO implementations of the same function
    side by side -- the per-frame port (oracle/torch_ref.LstmTrainStepCPU: a Python loop over frames like dynamic_rnn's
    while_loop) and its torch.nn.LSTM twin (LstmTrainStepOneDNN: PyTorch's fused CPU LSTM; tests/test_oracle_thirdparty.py proves
    it computes the same cell) -- `value` is the FASTER of the two: a baseline should not lose because its loop is in Python."""
    from oracle import torch_ref
    gen = torch.Generator().manual_seed(1)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if workload == "moe":
        B = 1024
        x = torch.rand((B, D_IN), generator=gen) * 4.0 - 2.0
        y = torch.rand((B, VOCAB), generator=gen) < (3.4 / VOCAB)
        probe = torch_ref.MoeTrainStepCPU(D=D_IN, V=VOCAB, M=MIX, batch_size=128, dtype=torch.float32, seed=0)
        cores, usable = _pick_threads(lambda: probe.step(x[:128], y[:128]))
        st = torch_ref.MoeTrainStepCPU(D=D_IN, V=VOCAB, M=MIX, batch_size=B, dtype=torch.float32, seed=0)
        r = _time_steps(lambda: st.step(x, y), seconds, min_steps, max_seconds)
        return {"value": r["steps"] * B / r["seconds"], "unit": "videos/s", "cores": cores, "kind": "port", "timed_steps": r["steps"],
                "batch": B, "seconds": r["seconds"], "usable_cores": usable,
                "sample": "%d steps of the same fp32 MoeModel training step at B=%d on torch-CPU (oracle/torch_ref.py; TF1 itself is not "
                          "runnable here), %.1f s, %d threads of %d usable cores" % (r["steps"], B, r["seconds"], cores, usable)}
    if workload != "lstm":
        return None
    B = 32
    q = torch.randint(0, 256, (B, FRAMES, D_IN), generator=gen, dtype=torch.uint8)
    y = torch.rand((B, VOCAB), generator=gen) < (3.4 / VOCAB)
    nf = torch.full((B,), FRAMES, dtype=torch.int32)
    impls = {}
    budget = max_seconds / 2.0
    for name, cls, steps_min in (("per_frame_port", torch_ref.LstmTrainStepCPU, 3), ("nn_lstm_twin", torch_ref.LstmTrainStepOneDNN, min_steps)):
        st = cls(D=D_IN, H=LSTM_H, L=LSTM_L, V=VOCAB, M=MIX, batch_size=B, dtype=torch.float32, seed=0)
        short_q, short_nf = q[:, :12].contiguous(), torch.full((B,), 12, dtype=torch.int32)
        cores, _ = _pick_threads(lambda: st.step(short_q, short_nf, y), candidates=(min(usable, 128), 64, 32, 16))
        r = _time_steps(lambda: st.step(q, nf, y), min(seconds, budget), steps_min, budget)
        impls[name] = {"value": r["steps"] * B / r["seconds"], "unit": "videos/s", "cores": cores, "timed_steps": r["steps"],
                       "seconds": r["seconds"], "is": cls.__doc__.split(".")[0].strip()[:160]}
        if name == "nn_lstm_twin" and cores != usable:
            # BASELINE.md promised set_num_threads(os.cpu_count()); the probe picks fewer because oneDNN's LSTM does not scale to every
            # core of this host.  What all cores give is reported beside the chosen count -- from the 12-frame probe step, in a child
            # process under a wall-clock limit: a full step at 256 threads ran for more than 15 MINUTES on a round-6 box (the first
            # form of this leg timed two full steps in-process and took the whole bench line with it).
            impls[name]["all_cores"] = _all_cores_probe(usable, cores)
        del st
    best = max(impls, key=lambda k: impls[k]["value"])
    bi = impls[best]
    return {"value": bi["value"], "unit": "videos/s", "cores": bi["cores"], "kind": "port", "implementation": best,
            "timed_steps": bi["timed_steps"], "batch": B, "seconds": bi["seconds"], "usable_cores": usable, "implementations": impls,
            "all_cores": impls.get("nn_lstm_twin", {}).get("all_cores"),
            "sample": "%d steps of the same fp32 LstmModel (2x1024, F=300) + MoE head training step at B=%d on torch-CPU (oracle/"
                      "torch_ref.py, %s; TF1 itself is not runnable here), %.1f s, %d threads (chosen by probe) of %d usable cores; the "
                      "other implementation is timed beside it under `implementations`" % (bi["timed_steps"], B, best, bi["seconds"],
                                                                                           bi["cores"], usable)}


def _cpu_probe_child(threads):
    """(child of _all_cores_probe) one 12-frame cut of the cpu_baseline LSTM step at `threads` threads; prints its seconds."""
    from oracle import torch_ref
    torch.set_num_threads(threads)
    gen = torch.Generator().manual_seed(1)
    B = 32
    q = torch.randint(0, 256, (B, 12, D_IN), generator=gen, dtype=torch.uint8)
    y = torch.rand((B, VOCAB), generator=gen) < (3.4 / VOCAB)
    nf = torch.full((B,), 12, dtype=torch.int32)
    st = torch_ref.LstmTrainStepOneDNN(D=D_IN, H=LSTM_H, L=LSTM_L, V=VOCAB, M=MIX, batch_size=B, dtype=torch.float32, seed=0)
    st.step(q, nf, y)
    t0 = time.perf_counter()
    st.step(q, nf, y)
    print(json.dumps({"probe_seconds": time.perf_counter() - t0}), flush=True)


def _all_cores_probe(usable, chosen, limit=45.0):
    """The 12-frame probe step of the nn.LSTM twin at ALL usable cores and at the chosen count, each in a child process killed after
    `limit` seconds: {"threads", "probe_seconds" | "timed_out_after_s", "chosen_threads", "chosen_probe_seconds"}."""
    import subprocess
    out = {"threads": usable, "chosen_threads": chosen, "what": "one 12-frame cut of the same training step (B = 32), child process"}
    for key, n in (("probe_seconds", usable), ("chosen_probe_seconds", chosen)):
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe-child", str(n)], capture_output=True, text=True,
                               timeout=limit, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
            lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
            out[key] = json.loads(lines[-1])["probe_seconds"] if lines else None
        except subprocess.TimeoutExpired:
            out["timed_out_after_s" if n == usable else "chosen_timed_out_after_s"] = limit
        except Exception as e:  # noqa: BLE001 -- a reported side number must never take the line down
            out[key + "_error"] = repr(e)[:120]
    return out


def _time_steps(stepf, seconds, min_steps, max_seconds):
    stepf()                                          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        stepf()
        n += 1
        el = time.perf_counter() - t0
        if (el >= seconds and n >= min_steps) or n >= 200 or el >= max_seconds:
            return {"steps": n, "seconds": el}


def library_identity():
    """Which library produced the numbers: path (YT8M_LIB overrides the in-tree build), sha256 of the file, and every YT8M_*
    environment variable of the process -- tuning / timing-experiment builds and knobs must be visible in the line."""
    import hashlib
    import yt8m_amd._lib as L
    h = hashlib.sha256()
    with open(L.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return {"path": os.path.relpath(L.LIB_PATH, ROOT) if L.LIB_PATH.startswith(ROOT) else L.LIB_PATH,
            "in_tree_default": os.path.abspath(L.LIB_PATH) == os.path.join(ROOT, "youtube-8m_amd", "libyt8m_hip.so"),
            "sha256": h.hexdigest(), "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("YT8M_")}}


LINE_LIMIT = 8192                    # the driver keeps ~8.9 KB of stdout: the ONE line must fit
SIDECAR = "bench_extra.json"


def _r(x, nd=5):
    """floats to `nd` significant digits (the line is a summary; the sidecar keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def _short(s, n):
    return s if s is None or len(s) <= n else s[:n - 3] + "..."


def compact_line(out, sidecar=SIDECAR):
    """The ONE stdout line of the contract, <= LINE_LIMIT bytes: headline fields + `roofline` (dominant kernel and a compact
    {frac, ms_per_step} map of the families) + `cpu_baseline` + `gap_at_20` + library sha256.  Everything else of `out` (the extra
    configurations with their per-family detail, placement, data-parallel trace, notes) stays in the sidecar file named here."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step",
                                "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    c = out.get("config") or {}
    line["config"] = {k: c.get(k) for k in ("workload", "per_gpu_batch", "global_batch", "frames", "parallelism", "params")}
    line["config"]["workload"] = _short(line["config"]["workload"], 260)
    r = out.get("roofline")
    if r:
        cr = {k: (r.get(k) if k in ("achieved", "peak", "frac") else _r(r.get(k)))
              for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step",
                                        "avg_launch_ms", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch",
                                        "occupied_cus", "frac_of_occupied_cus") if r.get(k) is not None or k == "traffic"}
        cr["peak_is"] = _short(r.get("peak_is"), 90)
        if r.get("traffic_source"):
            cr["traffic_source"] = _short(r["traffic_source"], 120)
        cr["families"] = {k: {"frac": _r(v["frac"], 4), "ms_per_step": _r(v["ms_per_step"], 4), "peak": _r(v["peak"], 4)}
                          for k, v in (r.get("families") or {}).items()}
        cr["other_ms_per_step"] = {k: _r(v["ms_per_step"], 4) for k, v in (r.get("other_families") or {}).items()}
        if r.get("blended_bound"):
            cr["blended_bound"] = {k: _r(r["blended_bound"].get(k), 4) for k in ("ms_per_step", "frac")}
        if r.get("exchange"):                                              # what the dominant kernel runs against when it is not its matrix pipe
            cr["exchange"] = {k: _r(r["exchange"].get(k), 4) for k in ("achieved_GBps_per_cu", "peak_GBps_per_cu", "frac", "us_per_step")}
        line["roofline"] = cr
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "implementation", "batch", "timed_steps",
                                                           "seconds", "usable_cores") if k in cb}
        if cb.get("all_cores"):
            line["cpu_baseline"]["all_cores"] = {k: _r(v) for k, v in cb["all_cores"].items()}
        line["cpu_baseline"]["sample"] = _short(cb.get("sample"), 240)
    else:
        line["cpu_baseline"] = None
    gp = out.get("gap_at_20")
    if gp:
        cg = {k: _r(gp.get(k)) for k in ("value", "hit_at_one", "perr", "train_steps", "batch", "heldout_videos", "error") if k in gp}
        for k, v in gp.items():
            if isinstance(v, dict) and k.startswith(("cpu_twin", "frame_twin")):
                cg[k] = ({"error": _short(v["error"], 120)} if "error" in v else
                         {kk: _r(v.get(kk)) for kk in ("gap_hip", "gap_cpu_port", "abs_diff", "within_target")})
        line["gap_at_20"] = cg
    else:
        line["gap_at_20"] = None
    ex = []
    for e in out.get("extra") or []:
        if "error" in e:
            ex.append({"workload": _short(e.get("workload"), 60), "error": _short(e["error"], 120)})
            continue
        er = e.get("roofline") or {}
        row = {"workload": _short(e["workload"], 60), "dtype": e["dtype"], "per_gpu_batch": e["per_gpu_batch"], "steps": e["steps"],
               "ms_per_step": _r(e["ms_per_step"]), "value": _r(e["value"])}
        if er.get("kernel"):
            row["dominant"] = {"kernel": er["kernel"], "frac": _r(er.get("frac"), 4), "bound": er.get("bound")}
        if er.get("blended_bound"):
            row["blended_frac"] = _r(er["blended_bound"].get("frac"), 4)
        if er.get("hbm"):
            row["hbm"] = {k: {"frac_of_hbm": _r(v.get("frac_of_hbm"), 4), "avg_launch_us": _r(v.get("avg_launch_us"), 4)}
                          for k, v in er["hbm"]["kernels"].items()}
            f8 = er["hbm"].get("forward_8d")
            if f8:
                row["hbm"]["forward_8d"] = {"us": _r(f8["us"], 4), "frac_of_hbm_8d_bytes": _r(f8["frac_of_hbm_8d_bytes"], 4)}
        ex.append(row)
    line["extra"] = ex
    lib = out.get("library") or {}
    line["library"] = {"sha256": lib.get("sha256"), "in_tree_default": lib.get("in_tree_default"), "env": lib.get("env")}
    rd = out.get("reducer")
    if rd:
        line["reducer"] = {k: rd.get(k) for k in ("algo", "reserved_cus", "reserve_rule", "emulated_footprint", "world", "bucket_MiB", "layer_buckets", "transport",
                                                  "forced_at_world_1")}
        pr = rd.get("per_rank") or []
        line["reducer"]["persist_timeouts"] = sum(1 for x in pr if x and x.get("persist_timeout"))
        traces = [x["dp_trace"] for x in pr if x and isinstance(x.get("dp_trace"), dict) and "exposed_allreduce_ms" in x["dp_trace"]]
        if traces:
            # the rank whose all-reduce sticks out furthest behind its backward pass, bucket by bucket: [MB, enqueued, landed, exposed] in ms
            # after the start of the backward pass (exposed = how far the bucket's landing lies behind the end of the backward pass)
            w = max(traces, key=lambda t: t["exposed_allreduce_ms"])
            line["reducer"]["exposed_allreduce_ms_max"] = _r(w["exposed_allreduce_ms"], 4)
            line["reducer"]["backward_ms"] = _r(w["backward_ms"], 4)
            line["reducer"]["buckets_MB_enq_land_exposed_ms"] = [[_r(b["MB"], 4), _r(b["enqueued_ms"], 3), _r(b["landed_ms"], 3),
                                                                  _r(max(0.0, b["landed_ms"] - w["backward_ms"]), 3)] for b in w["buckets"]][:8]
    line["sidecar"] = sidecar
    # the contract is the size: shed the optional detail, in this order, until the line fits
    for drop in (lambda: line["library"].pop("env", None), lambda: line["roofline"] and line["roofline"].pop("other_ms_per_step", None),
                 lambda: [e.pop("hbm", None) for e in line["extra"]], lambda: line.update(extra=[{"workload": e["workload"], "ms_per_step":
                                                                                                e.get("ms_per_step")} for e in line["extra"]]),
                 lambda: line.pop("extra", None), lambda: line.pop("gap_at_20", None)):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        drop()
    return line


def write_sidecar(out):
    """Full detail next to the line (and under gpurun_out/ when that exists, so a gpurun call brings it back)."""
    paths = [os.path.join(ROOT, SIDECAR)]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", SIDECAR))
    for p in paths:
        try:
            with open(p, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:                                   # a read-only tree must not cost the line
            sys.stderr.write("bench: could not write %s: %r\n" % (p, e))
    return paths[0]


_T0 = time.perf_counter()


def note(msg):
    """progress on stderr (the JSON line on stdout is printed last: a leg that overruns must be visible in the log)"""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-probe-child":
        _cpu_probe_child(int(sys.argv[2]))
        return
    a = parse()
    maybe_relaunch(a)
    __graft_entry__.load_package()
    import yt8m_amd._lib as L
    import yt8m_amd.parallel as parallel
    import torch.distributed as dist

    rank, world, local = parallel.init_from_env()
    assert world == a.gpus, "--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the measured path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = L.lib()
    rccl_ranks = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1
    assert rccl_ranks == a.gpus, "RCCL world size %d != --gpus %d" % (rccl_ranks, a.gpus)

    cfg = WORKLOADS[a.workload]
    B = a.batch or cfg["batch"]
    bf16 = a.dtype == "bf16"
    if a.force_reducer and world == 1 and not dist.is_initialized():
        # a 1-rank RCCL group: the whole data-parallel machinery (bucketed async all-reduce, per-bucket clip + Adam, the CU headroom of
        # the persistent recurrences) runs on one GPU -- what a single-GPU box can measure of the N > 1 path
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                device_id=dev)
    reducer = None
    if world > 1 or a.force_reducer:
        reducer = parallel.GradReducer(**({"bucket_bytes": a.dp_bucket_mb << 20} if a.dp_bucket_mb else {}))
    g, tg, pool = build(a.workload, B, world, rank, dev, reducer, bf16)
    if a.pool:
        pool = make_pool(a.workload, B, dev, rank, a.pool)
    note("built %s, B = %d per GPU, world %d" % (a.workload, B, world))
    el, run = timed_run(tg, pool, a.steps, a.warmup, world, dev, dist)
    note("timed region: %.2f ms/step" % (el / a.steps * 1e3))
    params = sum(v.numel() for v in g.trainable_variables())
    import yt8m_amd.seq_ops as seq_ops
    persist_timeout = None
    try:
        seq_ops.check_persist_errors()          # a persistent launch that timed out must fail the bench, not skew it
    except Exception as e:                      # ... but every rank's word goes into the line first (a SCALE run must explain itself)
        persist_timeout = repr(e)
    placement = None
    if a.workload == "lstm":
        # diagnostics of the timed region (+ warm-up): persistent recurrence workgroups that did not land on the XCD their index
        # suggests lose the L2 sharing of the state fetch -- tells an unlucky placement from a slow kernel
        nl, nw, off = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        lib.yt8m_lstm_persist_placement_stats(ctypes.byref(nl), ctypes.byref(nw), ctypes.byref(off), 1)
        placement = {"persistent_launches": nl.value, "workgroups": nw.value, "workgroups_off_their_xcd": off.value}
    # data-parallel timeline of two more (untimed) steps: when each gradient bucket's collective was enqueued and landed relative to
    # the backward pass, the all-reduce time left exposed behind it -- per rank (parallel.GradReducer.trace)
    dp_trace = None
    if reducer is not None and reducer.active:
        reducer.trace = True
        run(2, 0)
        dp_trace = reducer.trace_report()
        reducer.trace = False
    per_rank = {"rank": rank, "ms_per_step_local": None, "persist_timeout": persist_timeout, "placement": placement, "dp_trace": dp_trace}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank)
        per_rank_all = gathered
    else:
        per_rank_all = [per_rank]
    if persist_timeout is not None:
        raise RuntimeError("rank %d: %s" % (rank, persist_timeout))
    placement = placement if rank == 0 else None

    roof = None
    if not a.no_roofline:
        fam = profile_pass(lib, run, min(a.steps, 5 if a.workload != "moe" else 20), rank)
        if rank == 0:
            bwd_cus = None
            if a.workload == "lstm" and lib.yt8m_lstm_persist_bwd_supported(B, LSTM_H):
                bwd_cus = int(os.environ.get("YT8M_PERSIST_CUS_BWD", "128"))
            fwd_x3 = fwd_form(lib, B, LSTM_H) if (a.workload == "lstm" and not bf16) else False
            bwd_f16 = a.workload == "lstm" and not bf16 and bwd_on_f16_pipe(lib, B, LSTM_H)
            roof = roofline_from(fam, cfg["flops"](B), bf16, bwd_cus=bwd_cus, fwd_x3=fwd_x3, step_ms=el / a.steps * 1e3, bwd_f16=bwd_f16)
            if roof and bwd_f16:
                pair = os.environ.get("YT8M_PERSIST_BWD_PAIR", "0") != "0" and (B + 15) // 16 // 2 <= 8
                roof["exchange"] = exchange_view(roof.get("families", {}).get("lstm_recurrence_bwd"), B, LSTM_H, bwd_cus, True, pair=pair)
            if a.workload == "moe" and B == 1024 and not bf16 and roof:
                try:                   # HBM-side bytes per GEMM launch from the committed PMC passes (profiles/r1_pmc_traffic.md)
                    pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
                    ks = [v for k, v in pm["kernels"].items() if k.startswith("gemm_grouped_kernel")]
                    roof["traffic"] = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in ks) / len(ks)
                    roof["traffic_unit"] = "bytes/launch (L2-miss side, PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_pmc_traffic.md)"
                    roof["traffic_source"] = "profiles/r1_pmc_traffic.json -- a committed rocprofv3 --pmc pass, NOT measured in this run"
                    roof["algorithmic_bytes_per_launch"] = 4.0 * (B * D_IN + D_IN * VOCAB * (2 * MIX + 1) + B * VOCAB * (2 * MIX + 1))
                except Exception:
                    pass
            elif a.workload == "lstm" and roof:
                try:                   # PMC passes of the headline step, committed with the round's profiles
                    src = next(f for f in ("r6_pmc_traffic_lstm.json", "r5_pmc_traffic_lstm.json", "r4_pmc_traffic_lstm.json", "r3_pmc_traffic_lstm.json", "r2_pmc_traffic_lstm.json")
                               if os.path.exists(os.path.join(ROOT, "profiles", f)))
                    pm = json.load(open(os.path.join(ROOT, "profiles", src)))
                    key = roof["kernel"]
                    roof["traffic_source"] = ("profiles/%s -- committed rocprofv3 --pmc passes of this workload (tools/pmc_run_lstm.py), "
                                              "NOT measured in this run" % src)
                    roof["traffic"] = pm["families"][key]["hbm_bytes_per_launch"]
                    roof["traffic_unit"] = pm["unit"]
                    roof["traffic_kernel"] = pm["families"][key]["kernel"]
                    roof["algorithmic_bytes_per_launch"] = pm["families"][key].get("algorithmic_bytes_per_launch")
                except Exception:
                    pass
            if roof:
                # whole-step view: every algorithmic FLOP of the step over the WALL time of the timed region (the families above
                # are hipEvent durations of launches that share the chip across streams, so they overlap and add up to more)
                tot = sum(cfg["flops"](B).values())
                roof["step_level"] = {"algorithmic_flops_per_step": tot, "achieved": tot / (el / a.steps) / 1e12, "unit": "TFLOP/s",
                                      "is": "all algorithmic (fp32-equivalent) FLOPs of the step over wall time: a rate, not a roofline "
                                            "fraction -- the step's products run on two pipes; see blended_bound"}
    del tg, g, pool
    gc.collect()
    torch.cuda.empty_cache()

    extra = []
    if rank == 0 and world == 1 and not a.no_extra and a.workload == "lstm" and not bf16:
        for wl, b16, bb in (("moe", False, None), ("netvlad", False, None), ("lstm", True, None), ("composite", True, None),
                            # per-GPU batch sweep of the headline configuration: more rows per workgroup = more independent
                            # chains per persistent workgroup; tells the 8-GPU run which per-GPU batch to use
                            ("lstm", False, 256), ("lstm", False, 512)):
            try:
                extra.append(extra_line(wl, dev, lib, bf16=b16, batch=bb, tag=" -- per-GPU batch sweep, B = %d" % bb if bb else ""))
                note("extra line %s%s%s done" % (wl, " bf16" if b16 else "", " B=%d" % bb if bb else ""))
            except Exception as e:                                    # an extra line must never break the headline
                extra.append({"workload": WORKLOADS[wl]["name"], "error": repr(e)})

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        note("cpu_baseline starts")
        cpu = cpu_baseline(a.workload, a.cpu_seconds, a.cpu_steps, a.cpu_max_seconds)
        note("cpu_baseline done: %s" % (cpu and cpu.get("sample")))

    gap = None
    if rank == 0 and world == 1 and not a.no_gap and not bf16:
        try:
            gap = gap_leg(dev)
            state = gap.pop("_state")
            note("gap leg done")
            try:                                                      # the acceptance check on a frame-level model
                gap["frame_twin_lstm"] = gap_twin_frames(dev)
            except Exception as e:
                gap["frame_twin_lstm"] = {"error": repr(e)}
            note("frame_twin_lstm done")
            for key, kw in (("cpu_twin", {}),
                            # the same acceptance check at the FULL model size (D = 1152, V = 4716), continued from the leg's trained model
                            ("cpu_twin_full_size", dict(D_=D_IN, V_=VOCAB, M_=MIX, B_=256, steps=24, held=2048, state=state))):
                try:
                    gap[key] = gap_twin(dev, **kw)
                except Exception as e:
                    gap[key] = {"error": repr(e)}
                note("%s done" % key)
        except Exception as e:                                    # never let the secondary metric break the bench line
            gap = {"value": None, "error": repr(e)}

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {"metric": "training videos/sec", "value": a.steps * B * world / el, "unit": "videos/s",
               "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "%s, %s training step (fwd+bwd+clip+Adam%s)"
                                      % (cfg["name"], "bf16-operand VARIANT (not the fp32 headline) of the" if bf16 else "fp32",
                                         "+RCCL all-reduce" if world > 1 else ""),
                          "per_gpu_batch": B, "global_batch": B * world, "frames": FRAMES if cfg["frame"] else None,
                          "parallelism": "dp%d" % world, "params": params,
                          "arithmetic": ("bf16 operands, fp32 accumulate" if bf16 else
                                         "fp32 values throughout, fp32 accumulation everywhere; large matrix products run on the 16-bit MFMA "
                                         "pipe from EXACT-TO-2^-22 splits of their fp32 operands: three f16 products of two half planes under "
                                         "power-of-two scales (h2: recurrent stack projections, dx, weight gradients, backward recurrence; two "
                                         "products where one operand is the uint8 frame) or six bf16 products of three planes (x3: forward "
                                         "recurrence, MoE / NetVLAD heads) -- per product term 2^-21 relative, the size of three fp32 roundings, "
                                         "inside the error of an fp32 FMA chain (tests/test_gpu_h2.py, test_gpu_h2_recur.py against fp64); "
                                         "YT8M_STACK_H2=0 YT8M_STACK_H2_RECUR=0 YT8M_GEMM_H2=0 select the six-product / fp32-pipe forms")},
               "roofline": roof, "cpu_baseline": cpu, "gap_at_20": gap, "extra": extra, "placement": placement,
               "library": library_identity(),
               "reducer": None if reducer is None else {"algo": reducer.algo, "reserved_cus": reducer.reserve_cus, "reserve_rule": reducer.reserve_rule,
                                                        "emulated_footprint": reducer.emulate, "world": reducer.world,
                                                        "bucket_MiB": reducer.bucket_elems * 4 / 2 ** 20,
                                                        "layer_buckets": bool(seq_ops.DP_LAYER_BUCKETS),
                                                        "transport": "CabiComm" if reducer.comm is not None else "torch.distributed",
                                                        "forced_at_world_1": bool(a.force_reducer and world == 1),
                                                        "per_rank": per_rank_all}}
    else:
        out = None
    rccl_loaded = dist.is_initialized()
    if rccl_loaded:
        dist.destroy_process_group()
    if out is not None:
        write_sidecar(out)
        sys.stdout.flush()
        print(json.dumps(compact_line(out)), flush=True)
    if rccl_loaded:
        # librccl writes a version banner ("RCCL version : ...", five lines) to stdout when the process exits: the contract is ONE
        # line, so every rank leaves without running the C runtime's exit handlers (everything of ours is flushed above)
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
