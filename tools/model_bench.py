"""Step time of the other BASELINE configs through the plugin surface (not the headline bench; for tuning).
usage: python tools/model_bench.py [config ...]   configs: logistic moe chain lstm lstm_attn netvlad dbof"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
import yt8m_amd.frame_level_models as flm  # noqa: E402
import yt8m_amd.losses as losses  # noqa: E402
import yt8m_amd.train as train  # noqa: E402
import yt8m_amd.video_level_models as vlm  # noqa: E402
from yt8m_amd.flags import FLAGS  # noqa: E402
from yt8m_amd.variables import reset_default_graph  # noqa: E402
import ctypes  # noqa: E402

dev = torch.device("cuda:0")
V = 4716
CONFIGS = {
    "logistic": dict(model=vlm.LogisticModel, B=128, frame=False),
    "moe": dict(model=vlm.MoeModel, B=1024, frame=False),
    "chain": dict(model=vlm.DeepCombineChainModel, B=512, frame=False, multitask=True,
                  flags=dict(deep_chain_layers=8, deep_chain_relu_cells=128, support_type=",".join(["label"] * 8))),
    "lstm": dict(model=flm.LstmModel, B=128, frame=True),
    "lstm_b512": dict(model=flm.LstmModel, B=512, frame=True),
    "lstm_bf16": dict(model=flm.LstmModel, B=128, frame=True, flags=dict(compute_dtype="bfloat16")),
    "lstm_attn": dict(model=flm.LstmAttentionMaxPoolingModel, B=128, frame=True),
    "netvlad": dict(model=flm.NetVLADModel, B=128, frame=True),
    "config5": dict(model=flm.GatedNetVLADAttentionChainModel, B=128, frame=True, multitask=True,
                    flags=dict(deep_chain_layers=3, deep_chain_relu_cells=128, support_type=",".join(["label"] * 3))),
    "config5_bf16": dict(model=flm.GatedNetVLADAttentionChainModel, B=128, frame=True, multitask=True,
                         flags=dict(deep_chain_layers=3, deep_chain_relu_cells=128, support_type=",".join(["label"] * 3),
                                    compute_dtype="bfloat16")),
    "netvlad_bf16": dict(model=flm.NetVLADModel, B=128, frame=True, flags=dict(compute_dtype="bfloat16")),
    "config5_b1024": dict(model=flm.GatedNetVLADAttentionChainModel, B=1024, frame=True, multitask=True,
                          flags=dict(deep_chain_layers=3, deep_chain_relu_cells=128, support_type=",".join(["label"] * 3))),
    "config5_bf16_b1024": dict(model=flm.GatedNetVLADAttentionChainModel, B=1024, frame=True, multitask=True,
                               flags=dict(deep_chain_layers=3, deep_chain_relu_cells=128, support_type=",".join(["label"] * 3),
                                          compute_dtype="bfloat16")),
    "lstm_parallel": dict(model=flm.LstmParallelFinaloutputModel, B=128, frame=True,
                          flags=dict(feature_sizes="1024,128", lstm_cells="1024,128")),
    "lstm_posattn": dict(model=flm.LstmPositionalAttentionMaxPoolingModel, B=128, frame=True),
    "cnn_chain": dict(model=flm.CnnDeepCombineChainModel, B=128, frame=True, multitask=True,
                      flags=dict(deep_chain_layers=3, deep_chain_relu_cells=128, support_type=",".join(["label"] * 3))),
    # (lr: with the reference's default 0.01 this stack leaves the finite range within seven steps on the uniform-noise frames of this
    #  harness -- the recurrence turns chaotic and dz reaches 1e38 in BOTH forms of the forward kernel, tools/gru_nan_debug.py; which
    #  step overflows depends on rounding.  The timing does not care, a printed "loss nan" invites the wrong conclusion.)
    "gru_pool": dict(model=flm.GruPoolingModel, B=128, frame=True, lr=0.001),
    "ln_lstm": dict(model=flm.LayerNormLstmMemoryModel, B=128, frame=True),
    "lstm_mem_dropout": dict(model=flm.LstmMemoryModel, B=128, frame=True, flags=dict(dropout=True, keep_prob=0.8)),
    "chain_dropout": dict(model=vlm.DeepCombineChainModel, B=512, frame=False, multitask=True,
                          flags=dict(deep_chain_layers=8, deep_chain_relu_cells=128, support_type=",".join(["label"] * 8),
                                     dropout=True, keep_prob=0.8)),
    "dbof": dict(model=flm.DbofModel, B=128, frame=True, flags=dict(dbof_add_batch_norm=False)),
}


def run(name, steps=5):
    cfg = CONFIGS[name]
    FLAGS.reset()
    for k, v in cfg.get("flags", {}).items():
        setattr(FLAGS, k, v)
    for kv in os.environ.get("YT8M_SET", "").split(","):           # e.g. YT8M_SET=lstm_pipeline_chunks=8
        if "=" in kv:
            k, v = kv.split("=", 1)
            setattr(FLAGS, k, type(getattr(FLAGS, k))(v) if not isinstance(getattr(FLAGS, k), bool) else v == "1")
    B = cfg["B"]
    g = reset_default_graph(device=dev, seed=0)
    mt = cfg.get("multitask", False)
    tg = train.TrainGraph(cfg["model"](), batch_size=B, graph=g, multitask=mt, base_learning_rate=cfg.get("lr", 0.01),
                          label_loss_fn=losses.MultiTaskCrossEntropyLoss() if mt else None)
    gen = torch.Generator(device=dev).manual_seed(1)
    if cfg["frame"]:
        x = torch.randint(0, 256, (B, 300, 1152), device=dev, generator=gen, dtype=torch.uint8)
        nf = torch.full((B,), 300, device=dev, dtype=torch.int32)
    else:
        x = torch.rand((B, 1152), device=dev, generator=gen) * 4 - 2
        nf = None
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    lib = L.lib()
    for _ in range(2):
        tg.step(x, y, nf)
    torch.cuda.synchronize()
    lib.yt8m_prof_reset()
    prof = os.environ.get("YT8M_NO_PROF") is None      # the per-family hipEvent profiler turns hipGraph replay off
    lib.yt8m_prof_enable(1 if prof else 0)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = tg.step(x, y, nf)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    lib.yt8m_prof_enable(0)
    fam = []
    for fid, fname in enumerate(["gemm", "fused", "elementwise", "optim", "lstm", "netvlad", "lstm_bwd", "gemm_x3"]):
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        lib.yt8m_prof_get(fid, ctypes.byref(n), ctypes.byref(ms))
        if n.value:
            fam.append("%s %d launches %.2f ms" % (fname, n.value // steps, ms.value / steps))
    params = sum(v.numel() for v in g.trainable_variables())
    print("%-10s B=%4d  %9.2f ms/step  %9.0f videos/s  loss %.3f  params %.1fM | %s"
          % (name, B, el * 1e3, B / el, float(out["loss"]), params / 1e6, "; ".join(fam)), flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or [c for c in CONFIGS if c not in ("lstm_b512", "lstm_bf16", "config5_bf16", "netvlad_bf16", "config5_b1024", "config5_bf16_b1024")]):
        run(n)
