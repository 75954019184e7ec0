"""Host-side enqueue time of the cfg[1] training step vs its GPU time (how far the Python host runs ahead of the device)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd.train as train
import yt8m_amd.video_level_models as vlm
from yt8m_amd.variables import reset_default_graph

dev = torch.device("cuda:0")
B, D, V = 1024, 1152, 4716
g = reset_default_graph(device=dev, seed=0)
tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.rand((B, D), device=dev, generator=gen) * 4 - 2
y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
for _ in range(10):
    tg.step(x, y)
torch.cuda.synchronize()
for n in (20, 50):
    t0 = time.perf_counter()
    for _ in range(n):
        tg.step(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %3d: host enqueue %.3f ms/step, total %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
