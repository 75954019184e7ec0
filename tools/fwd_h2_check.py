"""Forward persistent recurrence: six bf16 products (yt8m_lstm_persist_fwd) against three f16 products (yt8m_lstm_persist_fwd_h2) --
results, stand-alone time, and the same launches (a) with a num_frames vector, (b) right behind 60 ms of dense GEMM work (a hot,
clocked-down chip, as inside the training step)."""
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
B, F, H = 128, 300, 1024
z0 = torch.randn((F, B, 4 * H), device=dev) * 0.3
Wh = (torch.rand((H, 4 * H), device=dev) - 0.5) * 0.06
wword = torch.zeros(64, dtype=torch.int32, device=dev)
L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
nfv = torch.full((B,), F, dtype=torch.int32, device=dev)
ga = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
gb = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
res = {}
for case in ("plain", "num_frames", "hot"):
    for mode in ("x3", "h2", "x3", "h2"):
        cs = torch.zeros((F + 1, B, H), device=dev)
        hs = torch.zeros((F + 1, B, H), device=dev)
        out = torch.empty((F, B, H), device=dev)
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F), dtype=torch.uint8, device=dev)
        z = z0.clone()
        nf = _p(nfv) if case != "plain" else None
        if case == "hot":
            for _ in range(60):
                torch.mm(ga, gb)
        torch.cuda.synchronize() if case != "hot" else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if mode == "h2":
            L.check(lib.yt8m_lstm_persist_fwd_h2(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), nf, 0, F, B, H, 1.0, _p(wword), _p(pws),
                                                 pws.numel(), _stream()))
        else:
            L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), nf, 0, F, B, H, 1.0, _p(pws), pws.numel(),
                                              _stream()))
        e1.record()
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        res[mode] = (out.clone(), cs.clone(), z.clone())
        print("%-10s %s %.3f ms = %.2f us/step" % (case, mode, e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / F), flush=True)
a, b = res["x3"], res["h2"]
print("max |out diff| %.3g  max |c diff| / max |c| %.3g  max |gates diff| %.3g"
      % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max() / a[1].abs().max()), float((a[2] - b[2]).abs().max())))
