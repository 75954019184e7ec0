#!/usr/bin/env python
"""bench.py -- training videos/sec of the hot path on N MI355X (BASELINE.json metric).

Workload (config.workload): BASELINE.json configs[1] -- MoeModel (2 mixtures) on video-level features
(D=1152 -> V=4716 labels), batch 1024 PER GPU, fp32, full training step:
    L2-normalise -> MoE head (2 GEMMs + mixing) -> CrossEntropyLoss -> backward (2 dW GEMMs, bias sums)
    -> [RCCL gradient all-reduce for N>1] -> + l2*w -> per-tensor clip -> TF-Adam.
Inputs are synthetic and already resident in HBM (a pool of distinct batches cycled through).
One "step" = one such pass over one batch.  N>1: one process per GPU (torchrun contract), weak scaling.

Extra legs (rank 0):
  roofline     : hipEvent timing of the dominant kernel family (gemm_f32) inside the library, in a separate
                 pass after the timed region; algorithmic FLOPs per launch / average launch duration vs the fp32
                 matrix peak (157.3 TFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline : the torch-CPU fp32 restatement of the same step (oracle/torch_ref.py, kind "port") on the host
                 cores, bounded sample; N=1 only.
  gap_at_20    : BASELINE.json's second metric -- GAP@20 on a held-out synthetic teacher shard after 768 further training
                 steps of a fresh model (outside the timed region, ~1.5 s; N=1 only; --no-gap skips it).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

D_IN, VOCAB, MIX = 1152, 4716, 2
PEAK_F32_MATRIX_TFLOPS = 157.3
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md); only used by the --dtype bf16 variant line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch")
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic batches resident in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline).  bf16 = same step with bf16 MFMA operands for the head "
                         "GEMMs (fp32 accumulate / master weights / Adam): a separate, labelled line, never the headline.")
    ap.add_argument("--no-gap", action="store_true", help="skip the GAP@20 leg (BASELINE.json's second metric)")
    ap.add_argument("--force-reducer", action="store_true", help="exercise the RCCL reducer even at world size 1 (test aid)")
    return ap.parse_args()


def make_pool(n, B, dev, seed):
    """Video-level synthetic inputs: x = mean over frames of dequantised uint8 ~ N(0.008, small) is
    too degenerate to train on, so use a wide spread in the dequantised range [-2, 2]; labels ~ Bernoulli(3.4/4716)."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    xs, ys = [], []
    for _ in range(n):
        xs.append((torch.rand((B, D_IN), device=dev, generator=gen) * 4.0 - 2.0))
        ys.append((torch.rand((B, VOCAB), device=dev, generator=gen) < (3.4 / VOCAB)))
    return xs, ys


def gap_leg(dev, B, train_steps=768, heldout=16384, signal=3.0):
    """BASELINE.json's second metric, outside the timed region: a fresh MoeModel (M = 2, D = 1152, V = 4716) is
    trained for `train_steps` steps of B videos on a synthetic teacher shard (fixed W_t ~ N(0, 1/sqrt(D)), logit = x.W_t * 3
    - 3 + 0.5 N(0,1), threshold at ~3.4 positives per video) and evaluated with GAP@20 on a disjoint held-out shard; the
    per-video top-20 runs on the device.  (tests/test_gpu_models.py::test_gap_on_heldout_shard_matches_cpu_training checks
    the same procedure against the CPU oracle's training at a reduced size: |dGAP| < 0.001.)"""
    import yt8m_amd.eval_util as eval_util
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.variables import reset_default_graph
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
    g.finalize()
    before = evaluate(min(heldout, 4 * B))["gap"]
    pos = 0.0
    for _ in range(train_steps):
        x, logit = batch()
        y = logit > tau
        pos += float(y.float().sum(1).mean())
        tg.step(x, y)
    m = evaluate(heldout)
    return {"value": m["gap"], "hit_at_one": m["avg_hit_at_one"], "untrained": before, "train_steps": train_steps, "batch": B,
            "heldout_videos": (heldout // B) * B, "positives_per_video": pos / train_steps, "data": "synthetic teacher shard"}


def cpu_baseline(B, seconds):
    """Times the torch-CPU fp32 restatement of the same step on the host cores (a reported baseline, not a target).
    Thread count: the best of {all usable cores, 64, 32, 16} on a cheap B=128 probe (oversubscribed MKL is far slower)."""
    from oracle import torch_ref
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand((B, D_IN), generator=gen) * 4.0 - 2.0
    y = torch.rand((B, VOCAB), generator=gen) < (3.4 / VOCAB)
    # pick the thread count on a cheap B=128 probe (MKL/OpenMP with every hardware thread of a 256-thread host is an
    # order of magnitude SLOWER than with 32-64 on this step); candidates: all, 64, 32, 16
    probe = torch_ref.MoeTrainStepCPU(D=D_IN, V=VOCAB, M=MIX, batch_size=128, dtype=torch.float32, seed=0)
    best, best_t = None, None
    for cores in sorted({c for c in (usable, 64, 32, 16) if 1 <= c <= usable}, reverse=True):
        torch.set_num_threads(cores)
        probe.step(x[:128], y[:128])                 # warm-up at this thread count
        t0 = time.perf_counter()
        probe.step(x[:128], y[:128])
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best, best_t = cores, dt
    st = torch_ref.MoeTrainStepCPU(D=D_IN, V=VOCAB, M=MIX, batch_size=B, dtype=torch.float32, seed=0)
    torch.set_num_threads(best)
    st.step(x, y)                                    # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        st.step(x, y)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": n * B / el, "unit": "videos/s", "cores": best, "kind": "port",
            "sample": "%d steps of the same B=%d fp32 MoeModel training step on torch-CPU (oracle/torch_ref.py, TF1 itself "
                      "is not runnable here), %.1f s, %d usable cores" % (n, B, el, usable)}


def main():
    a = parse()
    __graft_entry__.load_package()
    import yt8m_amd._lib as L
    import yt8m_amd.parallel as parallel
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    from yt8m_amd.variables import reset_default_graph
    import torch.distributed as dist

    rank, world, local = parallel.init_from_env()
    if world != a.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (a.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback for the measured path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    L.lib()

    B = a.batch
    if os.environ.get("YT8M_FUSED_HEAD_LOSS") == "0":     # A/B aid
        from yt8m_amd.flags import FLAGS
        FLAGS.fused_head_loss = False
    bf16 = a.dtype == "bf16"
    if bf16:
        from yt8m_amd.flags import FLAGS
        FLAGS.compute_dtype = "bfloat16"
    g = reset_default_graph(device=dev, seed=0)
    reducer = parallel.GradReducer() if (world > 1 or a.force_reducer) else None
    tg = train.TrainGraph(vlm.MoeModel(), batch_size=B * world, graph=g, reducer=reducer)
    xs, ys = make_pool(a.pool, B, dev, seed=1234 + rank)

    def run(k, base):
        for i in range(k):
            j = (base + i) % a.pool
            tg.step(xs[j], ys[j])

    run(max(a.warmup, 1), 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps, a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    roof = None
    if rank == 0 and not a.no_roofline:
        lib = L.lib()
        import ctypes
        lib.yt8m_prof_reset()
        lib.yt8m_prof_enable(1)
    if not a.no_roofline:
        # every rank runs the same extra steps (the all-reduce is collective); only rank 0 records events
        run(min(a.steps, 20), 0)
        torch.cuda.synchronize()
    if rank == 0 and not a.no_roofline:
        lib.yt8m_prof_enable(0)
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        lib.yt8m_prof_get(0, ctypes.byref(n), ctypes.byref(ms))
        steps_p = min(a.steps, 20)
        # algorithmic FLOPs of the GEMM launches of one step: fwd x.Wg, x.We; bwd x^T.dZg, x^T.dZe
        flops_step = 2.0 * 2.0 * B * D_IN * VOCAB * (2 * MIX + 1)
        launches_step = n.value / float(steps_p) if steps_p else 0
        avg_ms = ms.value / max(n.value, 1)
        flops_launch = flops_step / max(launches_step, 1)
        ach = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        peak = PEAK_BF16_MATRIX_TFLOPS if bf16 else PEAK_F32_MATRIX_TFLOPS
        roof = {"bound": "mfma", "kernel": "gemm_grouped_kernel (%s)" % ("v_mfma_f32_32x32x16_bf16" if bf16 else "v_mfma_f32_32x32x2_f32"),
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": None, "launches_per_step": launches_step, "avg_launch_ms": avg_ms,
                "algorithmic_flops_per_launch": flops_launch}
        fam = {}
        for fid, name in [(2, "elementwise"), (3, "optimizer")]:
            lib.yt8m_prof_get(fid, ctypes.byref(n), ctypes.byref(ms))
            fam[name] = {"launches_per_step": n.value / float(steps_p), "ms_per_step": ms.value / steps_p}
        # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/r1_pmc_traffic.md:
        # separate FETCH_SIZE / WRITE_SIZE passes, calibrated on the copy probe); rocprofv3 cannot run inside bench.py.
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            ks = [v for k, v in pm["kernels"].items() if k.startswith("gemm_grouped_kernel")]
            if ks and B == 1024 and not bf16:
                roof["traffic"] = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in ks) / len(ks)
                roof["traffic_unit"] = "bytes/launch (L2-miss side, PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r1_pmc_traffic.md)"
                roof["algorithmic_bytes_per_launch"] = 4.0 * (B * D_IN + D_IN * VOCAB * (2 * MIX + 1) + B * VOCAB * (2 * MIX + 1))
        except Exception:
            pass
        roof["other_families"] = fam
        roof["gemm_ms_per_step"] = avg_ms * launches_step

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(B, a.cpu_seconds)

    gap = None
    if rank == 0 and world == 1 and not a.no_gap and not bf16:
        try:
            gap = gap_leg(dev, B)
        except Exception as e:                                    # never let the secondary metric break the bench line
            gap = {"value": None, "error": repr(e)}

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {"metric": "training videos/sec", "value": a.steps * B * world / el, "unit": "videos/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: MoeModel (2 mixtures) on video-level features, D=1152, V=4716, "
                                      "%s training step (fwd+bwd+clip+Adam%s)"
                                      % ("bf16-operand VARIANT (not the fp32 headline) of the" if bf16 else "fp32",
                                         "+RCCL all-reduce" if world > 1 else ""),
                          "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                          "params": 27173592},
               "roofline": roof, "cpu_baseline": cpu, "gap_at_20": gap}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
