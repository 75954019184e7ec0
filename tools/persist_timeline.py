"""Per-item timeline of workgroup 0 of the persistent LSTM forward kernel (timing variant of the library:
tools/build_variant.sh ptiming -DYT8M_PERSIST_TIMING; run with YT8M_LIB=tools/variants/lib_ptiming.so)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
from yt8m_amd.ops import _p, _stream  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
B, F, H = int(os.environ.get("PB", 128)), 300, 1024
lib = L.lib()
z0 = torch.randn((F, B, 4 * H), device=dev) * 0.3
Wh = (torch.rand((H, 4 * H), device=dev) - 0.5) * 0.06
cs = torch.zeros((F + 1, B, H), device=dev)
hs = torch.zeros((F + 1, B, H), device=dev)
out = torch.empty((F, B, H), device=dev)
nb = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F) if os.environ.get("PSTEPS") else lib.yt8m_lstm_persist_workspace_bytes(B, H)
pws = torch.zeros(nb, dtype=torch.uint8, device=dev)
BWD = len(sys.argv) > 1 and sys.argv[1] in ("bwd", "bwd_h2", "bwd_bf16")
MODE = sys.argv[1] if len(sys.argv) > 1 else "fwd"         # bwd_h2 / bwd_bf16: the f16 / one-plane bf16 forms of the recurrent product
gates = torch.rand((F, B, 4 * H), device=dev)
csr = torch.randn((F + 1, B, H), device=dev) * 0.5
dz = torch.empty((F, B, 4 * H), device=dev)
dout = torch.randn((F, B, H), device=dev) * 0.01
nfv = torch.full((B,), F, dtype=torch.int32, device=dev)
for it in range(2):
    if BWD:
        work = torch.zeros((4, B, H), device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if MODE == "bwd_h2":
            wword = torch.zeros(64, dtype=torch.int32, device=dev)
            L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
            L.check(lib.yt8m_lstm_persist_bwd_h2(_p(gates), _p(Wh), 4 * H, _p(csr), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H,
                                                 _p(wword), _p(pws), nb, _stream()))
        else:
            L.check((lib.yt8m_lstm_persist_bwd_bf16 if MODE == "bwd_bf16" else lib.yt8m_lstm_persist_bwd)(
                _p(gates), _p(Wh), 4 * H, _p(csr), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H, _p(pws), nb, _stream()))
        e1.record()
        torch.cuda.synchronize()
        print("bwd kernel %.3f ms = %.2f us/step" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / F))
        continue
    z = z0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if MODE == "fwd_h2":
        wword = torch.zeros(64, dtype=torch.int32, device=dev)
        L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
        L.check(lib.yt8m_lstm_persist_fwd_h2(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nfv), 0, F, B, H, 1.0, _p(wword), _p(pws), nb,
                                             _stream()))
    else:
        L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nfv), 0, F, B, H, 1.0, _p(pws), nb, _stream()))
    e1.record()
    torch.cuda.synchronize()
    print("kernel %.3f ms = %.2f us/step" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / F))
dbg = pws[nb - 65536:].view(torch.int64).cpu().numpy()[:4096].reshape(2, 256, 8)
for wsel, wname, names in ((0, "matrix wave 0", ["start", "polled + next loads issued", "mfma done", "partials written"]),
                           (1, "epilogue wave 8 (every 4th item)", ["start (operand loads issued)", "partials arrived", "reduced",
                                                                    "stores issued", "drained"] + ([] if BWD else ["gate math done"]))):
    d = dbg[wsel]
    ks = np.arange(40, 104) if wsel == 0 else np.arange(40, 104, 4)      # (an epilogue wave finishes every 4th item: rotation, both passes)
    print(wname, "-- cycles (mean / min / max)")
    period = np.diff(d[ks, 0])
    print("  %-34s %8.0f %8.0f %8.0f" % ("period", period.mean(), period.min(), period.max()))
    for i in range(min(len(names), 5) - 1):
        seg = d[ks, i + 1] - d[ks, i]
        print("  %-34s %8.0f %8.0f %8.0f" % (names[i] + " -> " + names[i + 1][:12], seg.mean(), seg.min(), seg.max()))
    if len(names) > 5:
        seg = d[ks, 5] - d[ks, 2]
        print("  %-34s %8.0f %8.0f %8.0f" % ("reduced -> gate math done", seg.mean(), seg.min(), seg.max()))
m, e = dbg[0], dbg[1]
ks = np.arange(40, 104, 4)
lag = e[ks, 4] - m[ks, 3]
print("matrix wave 0 partials written -> epilogue drained (publish latency) %8.0f %8.0f %8.0f" % (lag.mean(), lag.min(), lag.max()))
chain = m[43:104, 1] - m[40:101, 3]
print("matrix wave 0: partials written (item k) -> poll passed for item k+4 (the same chain's next step) %8.0f %8.0f %8.0f"
      % (chain.mean(), chain.min(), chain.max()))
