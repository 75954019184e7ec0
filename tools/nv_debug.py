"""Debug aid: NetVLAD fused-vs-generic at bench scale (B=128, F=300, D=1152)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.frame_level_models as flm
import yt8m_amd.train as train
from yt8m_amd.flags import FLAGS
from yt8m_amd.variables import reset_default_graph

dev = torch.device("cuda:0")
B, V = 128, 4716
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (B, 300, 1152), device=dev, generator=gen, dtype=torch.uint8)
nf = torch.full((B,), 300, device=dev, dtype=torch.int32)
y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
res = {}
for fold in (True, False):
    FLAGS.reset()
    FLAGS.fold_dequant = fold
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.NetVLADModel(), batch_size=B, graph=g)
    r = tg.forward(x, y, nf)
    g.finalize()
    r = tg.forward(x, y, nf)
    loss = tg.loss(r, y)
    loss.backward()
    res[fold] = {k: v.grad.clone() for k, v in g.vars.items()}
    print("fold", fold, "loss", float(loss), {k: (float(v.abs().max()), bool(torch.isfinite(v).all())) for k, v in res[fold].items()})
for k in res[True]:
    d = (res[True][k] - res[False][k]).abs().max()
    print(k, "max diff", float(d), "ref max", float(res[False][k].abs().max()))
# now train a few steps with the fused path
FLAGS.reset()
g = reset_default_graph(device=dev, seed=0)
tg = train.TrainGraph(flm.NetVLADModel(), batch_size=B, graph=g)
for i in range(8):
    out = tg.step(x, y, nf)
    print("step", i, "loss", float(out["loss"]), "Wc finite", bool(torch.isfinite(g.vars["netvlad/cluster_weights"].data).all()),
          "grad Wc max", float(g.vars["netvlad/cluster_weights"].grad.abs().max()))
