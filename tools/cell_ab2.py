"""Per-step losses and per-variable gradient checksums of the GRU pooling model (A/B with YT8M_NO_PACKED_CELLS=1)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.frame_level_models as flm
import yt8m_amd.train as train
from yt8m_amd.flags import FLAGS
from yt8m_amd.variables import reset_default_graph

dev = torch.device("cuda:0")
FLAGS.reset()
B, V = 128, 4716
g = reset_default_graph(device=dev, seed=0)
tg = train.TrainGraph(flm.GruPoolingModel(), batch_size=B, graph=g)
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (B, 300, 1152), device=dev, generator=gen, dtype=torch.uint8)
nf = torch.full((B,), 300, device=dev, dtype=torch.int32)
y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
for i in range(4):
    out = tg.step(x, y, nf)
    torch.cuda.synchronize()
    print("step", i, "loss", float(out["loss"]))
    if i < 2:
        for k, v in g.vars.items():
            print("   ", k, tuple(v.shape), "%.6e" % float(v.grad.double().abs().sum()), "w %.6e" % float(v.data.double().abs().sum()))
