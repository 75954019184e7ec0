"""How long does the host take to ENQUEUE one headline training step (Python + ctypes + torch allocator + launches) against how
long the device takes to run it?  If the two are close, a slow or busy host core makes the step launch-bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

bench.__graft_entry__.load_package()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
g, tg, pool = bench.build("lstm", 128, 1, 0, dev, None, False)
x, y, nf = pool[0]
for _ in range(3):
    tg.step(x, y, nf)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for i in range(N):
    x, y, nf = pool[i % len(pool)]
    tg.step(x, y, nf)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, device-bound total %.2f ms/step (%d steps)" % ((t1 - t0) * 1e3 / N, (t2 - t0) * 1e3 / N, N))
# one step alone, fully synchronised before: the enqueue time of a step the device is not holding back
torch.cuda.synchronize()
ts = []
for i in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    tg.step(x, y, nf)
    b = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((b - a) * 1e3)
print("host enqueue of a single step on an idle device: %s ms" % ", ".join("%.2f" % t for t in ts))
