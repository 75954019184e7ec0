"""Do two forward persistent recurrences (128 workgroups each at B = 128, H = 1024) run side by side on two high-priority streams?
Wall time of one launch against two independent launches enqueued back to back on two streams (and the same for the backward kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
L = importlib.import_module("youtube-8m_amd._lib")
ops = importlib.import_module("youtube-8m_amd.ops")
_p = ops._p
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = L.lib()
B, F, H = 128, 100, 1024


def layer():
    z = torch.randn((F, B, 4 * H), device=dev) * 0.3
    Wh = (torch.rand((H, 4 * H), device=dev) - 0.5) * 0.06
    cs = torch.zeros((F + 1, B, H), device=dev)
    hs = torch.zeros((F + 1, B, H), device=dev)
    out = torch.empty((F, B, H), device=dev)
    pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F), dtype=torch.uint8, device=dev)
    return z, Wh, cs, hs, out, pws


def fwd(t, s):
    z, Wh, cs, hs, out, pws = t
    L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(pws), pws.numel(), s.cuda_stream))


a, b = layer(), layer()
s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
for name, both in (("one launch", False), ("two launches, two streams", True), ("one launch", False), ("two launches, two streams", True)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fwd(a, s1)
    if both:
        fwd(b, s2)
    torch.cuda.synchronize()
    print("forward recurrence, %-28s %.3f ms for %d steps" % (name, (time.perf_counter() - t0) * 1e3, F), flush=True)
