"""gru_pool training steps, persistent vs per-step recurrences: loss per step and which gradients go non-finite."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_bench as mb
import yt8m_amd.seq_ops as seq_ops
from yt8m_amd.variables import reset_default_graph
import yt8m_amd.train as train
import yt8m_amd.losses as losses
dev = mb.dev

def go(fwd, bwd, steps=4):
    seq_ops.GRU_PERSIST_FWD, seq_ops.GRU_PERSIST_BWD = fwd, bwd
    cfg = mb.CONFIGS["gru_pool"]
    mb.FLAGS.reset()
    for k, v in cfg.get("flags", {}).items():
        setattr(mb.FLAGS, k, v)
    B = cfg["B"]
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(cfg["model"](), batch_size=B, graph=g)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.randint(0, 256, (B, 300, 1152), device=dev, generator=gen, dtype=torch.uint8)
    nf = torch.full((B,), 300, device=dev, dtype=torch.int32)
    y = torch.rand((B, mb.V), device=dev, generator=gen) < 3.4 / mb.V
    out = []
    for s in range(steps):
        o = tg.step(x, y, nf)
        torch.cuda.synchronize()
        bad = [v.name for v in g.trainable_variables() if not bool(torch.isfinite(v.data).all())]
        out.append("%.4f%s" % (float(o["loss"]), (" bad params: " + ",".join(bad[:4])) if bad else ""))
    print("fwd=%d bwd=%d: %s" % (fwd, bwd, " | ".join(out)), flush=True)

for fb in ((False, False), (True, False), (False, True), (True, True)):
    go(*fb)
