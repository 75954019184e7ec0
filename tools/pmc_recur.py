"""Workload for the round-6 counter study of the persistent recurrences: the f16 ("h2") backward recurrence and the
f16 forward recurrence of the headline shape (B = 128, H = 1024, 300 steps, one exchange image per step), stand-alone, a few launches
each.  Run under `rocprofv3 --kernel-trace --pmc <counters>` (one counter group per run: tools/r6_pmc_recur.sh); without a profiler it
prints the hipEvent time per step.  usage: python tools/pmc_recur.py [launches]   (env PCHECK_B / PCHECK_F: batch / steps)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
from yt8m_amd.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
lib = L.lib()
B, F, H = int(os.environ.get("PCHECK_B", "128")), int(os.environ.get("PCHECK_F", "300")), 1024
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gen = torch.Generator(device=dev).manual_seed(0)
gates = torch.rand((F, B, 4 * H), device=dev, generator=gen)
Wh = (torch.rand((H, 4 * H), device=dev, generator=gen) - 0.5) * 0.06
cs = torch.randn((F + 1, B, H), device=dev, generator=gen) * 0.5
dz = torch.empty((F, B, 4 * H), device=dev)
dout = torch.randn((F, B, H), device=dev, generator=gen) * 0.01
wword = torch.zeros(64, dtype=torch.int32, device=dev)
L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
nbytes = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)
for it in range(N):
    pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    work = torch.zeros((4, B, H), device=dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.yt8m_lstm_persist_bwd_h2(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H,
                                         _p(wword), _p(pws), pws.numel(), _stream()))
    e1.record()
    torch.cuda.synchronize()
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
    print("bwd h2: %.3f ms for %d steps = %.2f us/step" % (e0.elapsed_time(e1), F, e0.elapsed_time(e1) * 1e3 / F), flush=True)
z0 = torch.randn((F, B, 4 * H), device=dev, generator=gen) * 0.3
hs = torch.zeros((F + 1, B, H), device=dev)
cs0 = torch.zeros((F + 1, B, H), device=dev)
out = torch.empty((F, B, H), device=dev)
for it in range(N):
    pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    z = z0.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.yt8m_lstm_persist_fwd_h2(_p(z), _p(Wh), 4 * H, _p(cs0), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(wword), _p(pws),
                                         pws.numel(), _stream()))
    e1.record()
    torch.cuda.synchronize()
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
    print("fwd h2: %.3f ms for %d steps = %.2f us/step" % (e0.elapsed_time(e1), F, e0.elapsed_time(e1) * 1e3 / F), flush=True)
