"""Workload for the round-2 PMC passes: the HBM copy probe (known byte count: calibrates FETCH_SIZE / WRITE_SIZE on this box as
MI355X_MICROARCH.md prescribes) followed by a few training steps of the headline workload (BASELINE configs[3], LstmModel B=128)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
import yt8m_amd.frame_level_models as flm  # noqa: E402
import yt8m_amd.train as train  # noqa: E402
from yt8m_amd.variables import reset_default_graph  # noqa: E402

dev = torch.device("cuda:0")
lib = L.lib()
n = 64 * 1024 * 1024                      # 256 MiB read + 256 MiB written per launch
a = torch.empty(n, device=dev).normal_()
b = torch.empty_like(a)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.yt8m_probe_copy_f32(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, st)
torch.cuda.synchronize()
del a, b
B, F, D, V = 128, 300, 1152, 4716
g = reset_default_graph(device=dev, seed=0)
tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
nf = torch.full((B,), F, device=dev, dtype=torch.int32)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    tg.step(x, y, nf)
torch.cuda.synchronize()
print("pmc workload done")
