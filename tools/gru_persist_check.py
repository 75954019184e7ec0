"""Persistent GRU (csrc/gru_persist.inl) against the per-step kernels at the plugin shape: max differences per array, first bad time step,
and the stand-alone time per step of both forms.  usage: python tools/gru_persist_check.py [F] [B] [H]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.load_package()
import yt8m_amd._lib as L
import yt8m_amd.ops as ops
dev = torch.device("cuda:0")
lib = L.lib()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
_p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
_st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(1)
zg0 = torch.randn((F, B, 2 * H), device=dev, generator=g) * 0.8 + 0.5
zc0 = torch.randn((F, B, H), device=dev, generator=g) * 0.8
Wg = (torch.rand((H, 2 * H), device=dev, generator=g) - 0.5) * 0.08
Wc = (torch.rand((H, H), device=dev, generator=g) - 0.5) * 0.08
h0 = torch.randn((B, H), device=dev, generator=g) * 0.3
nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
nf[0], nf[1] = F, 0
dout = torch.randn((F, B, H), device=dev, generator=g) * 0.1
dhF = torch.randn((B, H), device=dev, generator=g) * 0.1
ws = ops._workspace(dev)
pws = torch.zeros(lib.yt8m_gru_persist_workspace_bytes(B, H, F), dtype=torch.uint8, device=dev)

def run(persist, timing=None):
    zg, zc = zg0.clone(), zc0.clone()
    hs = torch.zeros((F + 1, B, H), device=dev); hs[0] = h0
    rh, out = torch.zeros((F, B, H), device=dev), torch.zeros((F, B, H), device=dev)
    dzg, dzc = torch.zeros((F, B, 2 * H), device=dev), torch.zeros((F, B, H), device=dev)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    if persist:
        L.check(lib.yt8m_gru_persist_fwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(rh), _p(out), _p(nf), 0, F, B, H, _p(pws), pws.numel(), _st()))
        e[1].record()
        work = dhF.clone()
        L.check(lib.yt8m_gru_persist_bwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(dout), _p(dzg), _p(dzc), _p(work), _p(nf), 0, F, B, H, _p(pws), pws.numel(), _st()))
        dh0 = work
    else:
        L.check(lib.yt8m_gru_layer_fwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(rh), _p(out), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _st()))
        e[1].record()
        work = torch.zeros((3, B, H), device=dev)
        L.check(lib.yt8m_gru_layer_bwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(dout), _p(dhF), _p(dzg), _p(dzc), _p(work), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _st()))
        dh0 = work[F % 2].clone()
    e[2].record()
    torch.cuda.synchronize()
    if timing is not None:
        timing.append((e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
    return dict(zg=zg, zc=zc, hs=hs, rh=rh, out=out, dzg=dzg, dzc=dzc, dh0=dh0)

a = run(False)
b = run(True)
try:
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _st()))
except Exception as ex:
    print("status:", ex)
for k in a:
    x, y = a[k], b[k]
    bad = ~torch.isfinite(y)
    d = (x - y).abs()
    d[bad] = 0
    line = "%-4s max|diff| %.3e  max|ref| %.3e  non-finite %d" % (k, float(d.max()), float(x.abs().max()), int(bad.sum()))
    if x.dim() == 3:
        per_t = d.amax(dim=(1, 2))
        worst = int(per_t.argmax())
        line += "  worst t %d" % worst
        if bad.any():
            line += "  first non-finite t %d" % int(bad.flatten(1).any(dim=1).nonzero()[0])
        big = (per_t > 1e-4 * float(x.abs().max())).nonzero().flatten().tolist()
        if big:
            line += "  steps off: %s" % (big[:8],)
    print(line)
# workspace reuse: several more persistent runs on the same pws, the last one compared again; then large pre-activations
for _ in range(3):
    c = run(True)
print("after reuse: " + "  ".join("%s %.2e" % (k, float((a[k] - c[k]).abs().max())) for k in a))
zg0 *= 40.0; zc0 *= 40.0
a2 = run(False); c2 = run(True)
print("x40 inputs:  " + "  ".join("%s %.2e%s" % (k, float((a2[k] - c2[k]).abs().nan_to_num(1e9).max()), "" if bool(torch.isfinite(c2[k]).all()) else " NONFINITE") for k in a))
zg0 /= 40.0; zc0 /= 40.0
for name, p in (("per-step", False), ("persistent", True)):
    tm = []
    for _ in range(4):
        run(p, tm)
    f = min(t[0] for t in tm[1:]); bw = min(t[1] for t in tm[1:])
    print("%-10s forward %.3f ms = %.2f us/step   backward %.3f ms = %.2f us/step" % (name, f, f * 1e3 / F, bw, bw * 1e3 / F))
