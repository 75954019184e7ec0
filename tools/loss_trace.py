"""Per-step losses of a plugin config on the synthetic bench input (debug: which knob makes a run leave the finite range).
usage: python tools/loss_trace.py <config> [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_bench as mb  # noqa: E402


def main():
    name = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    cfg = mb.CONFIGS[name]
    mb.FLAGS.reset()
    for k, v in cfg.get("flags", {}).items():
        setattr(mb.FLAGS, k, v)
    for kv in os.environ.get("YT8M_SET", "").split(","):           # e.g. YT8M_SET=fold_dequant=0: the float-frames path of a plugin
        if "=" in kv:
            k, v = kv.split("=", 1)
            cur = getattr(mb.FLAGS, k)
            setattr(mb.FLAGS, k, (v == "1") if isinstance(cur, bool) else type(cur)(v))
    B = cfg["B"]
    g = mb.reset_default_graph(device=mb.dev, seed=0)
    mt = cfg.get("multitask", False)
    tg = mb.train.TrainGraph(cfg["model"](), batch_size=B, graph=g, multitask=mt,
                             base_learning_rate=float(os.environ.get("LR", cfg.get("lr", 0.01))),
                             label_loss_fn=mb.losses.MultiTaskCrossEntropyLoss() if mt else None)
    gen = torch.Generator(device=mb.dev).manual_seed(1)
    x = torch.randint(0, 256, (B, 300, 1152), device=mb.dev, generator=gen, dtype=torch.uint8)
    nf = torch.full((B,), 300, device=mb.dev, dtype=torch.int32)
    y = torch.rand((B, mb.V), device=mb.dev, generator=gen) < 3.4 / mb.V
    out = []
    for _ in range(steps):
        o = tg.step(x, y, nf)
        out.append(float(o["loss"]))
    mx = max(float(v.data.abs().max()) for v in g.trainable_variables())
    print(name, " ".join("%.4g" % v for v in out), "| max |param| %.4g" % mx, flush=True)


main()
