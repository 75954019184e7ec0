"""Debug: where does the gru_pool run with the persistent forward leave the finite range?  Every gru_layer call of a training run is
replayed through the library in both forms (persistent launch / per-step kernels): saved tensors compared, then the per-step
backward on each form's saved tensors."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import model_bench as mb  # noqa: E402
import yt8m_amd.seq_ops as so  # noqa: E402
from yt8m_amd import _lib, ops  # noqa: E402
from yt8m_amd.ops import _p, _stream  # noqa: E402

orig = so.gru_layer
STEP = [0]
ONLY = int(os.environ.get("DBG_FROM_STEP", "3"))


def nonfin(t):
    return int((~torch.isfinite(t)).sum())


def fwd(x_tm, Wg, bg, Wc, bc, nf, persist):
    F, B, Din = x_tm.shape
    H = Wc.data.shape[1]
    dev = x_tm.device
    L = _lib.lib()
    x2 = x_tm.view(F * B, Din)
    zg = torch.empty((F, B, 2 * H), dtype=torch.float32, device=dev)
    zc = torch.empty((F, B, H), dtype=torch.float32, device=dev)
    ops.gemm_any(x2, Wg.data[:Din], out=zg.view(F * B, 2 * H), bias=bg.data, bf16=False)
    ops.gemm_any(x2, Wc.data[:Din], out=zc.view(F * B, H), bias=bc.data, bf16=False)
    pre = (zg.clone(), zc.clone())
    hs = torch.empty((F + 1, B, H), dtype=torch.float32, device=dev)
    hs[0].zero_()
    rh = torch.empty((F, B, H), dtype=torch.float32, device=dev)
    out = torch.empty((F, B, H), dtype=torch.float32, device=dev)
    if persist:
        main = torch.cuda.current_stream(dev)
        pws = so._persist_ws(dev, main, "gru", L.yt8m_gru_persist_workspace_bytes(B, H, F))
        _lib.check(L.yt8m_gru_persist_fwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(rh), _p(out),
                                          _p(nf), 0, F, B, H, _p(pws), pws.numel(), _stream()))
    else:
        ws = ops._workspace(dev)
        _lib.check(L.yt8m_gru_layer_fwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(rh),
                                        _p(out), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _stream()))
    torch.cuda.synchronize()
    return dict(zg=zg, zc=zc, hs=hs, rh=rh, out=out, pre=pre)


def bwd(st, Wg, Wc, nf, dout, Din):
    F, B, H = st["out"].shape
    dev = dout.device
    dzg = torch.empty((F, B, 2 * H), dtype=torch.float32, device=dev)
    dzc = torch.empty((F, B, H), dtype=torch.float32, device=dev)
    work = torch.empty((3, B, H), dtype=torch.float32, device=dev)
    ws = ops._workspace(dev)
    _lib.check(_lib.lib().yt8m_gru_layer_bwd(_p(st["zg"]), _p(st["zc"]), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(st["hs"]),
                                             _p(dout), None, _p(dzg), _p(dzc), _p(work), _p(nf), F, B, H, _p(ws), ws.numel() * 4,
                                             _stream()))
    torch.cuda.synchronize()
    return dzg, dzc


def both(x_tm, Wg, bg, Wc, bc, num_frames):
    if STEP[0] >= ONLY:
        with torch.no_grad():
            nf = so._nf(num_frames)
            xd = x_tm.detach().contiguous()
            a = fwd(xd, Wg, bg, Wc, bc, nf, False)
            b = fwd(xd, Wg, bg, Wc, bc, nf, True)
            print("  step %d layer in=%d  max|Wg| %.3g max|pre zg| %.3g max|pre zc| %.3g" % (
                STEP[0], xd.shape[2], float(Wg.data.abs().max()), float(a["pre"][0].abs().max()), float(a["pre"][1].abs().max())), flush=True)
            for k in ("zg", "zc", "hs", "rh", "out"):
                d = (a[k] - b[k]).abs()
                print("    %-3s nonfinite per-step %d persist %d  max diff %.3g  (elements > 1e-3: %d)" % (
                    k, nonfin(a[k]), nonfin(b[k]), float(d[torch.isfinite(d)].max()), int((d > 1e-3).sum())), flush=True)
            gen = torch.Generator(device=xd.device).manual_seed(5)
            dout = torch.randn(a["out"].shape, device=xd.device, generator=gen) * 1e-3
            for name, st in (("per-step", a), ("persist", b)):
                dzg, dzc = bwd(st, Wg, Wc, nf, dout, xd.shape[2])
                print("    bwd on %-8s saved tensors: dzg nonfinite %d max %.3g | dzc nonfinite %d max %.3g" % (
                    name, nonfin(dzg), float(dzg[torch.isfinite(dzg)].abs().max()), nonfin(dzc), float(dzc[torch.isfinite(dzc)].abs().max())),
                    flush=True)
    return orig(x_tm, Wg, bg, Wc, bc, num_frames)


so.gru_layer = both


def main():
    cfg = mb.CONFIGS["gru_pool"]
    mb.FLAGS.reset()
    B = cfg["B"]
    g = mb.reset_default_graph(device=mb.dev, seed=0)
    tg = mb.train.TrainGraph(cfg["model"](), batch_size=B, graph=g)
    gen = torch.Generator(device=mb.dev).manual_seed(1)
    x = torch.randint(0, 256, (B, 300, 1152), device=mb.dev, generator=gen, dtype=torch.uint8)
    nf = torch.full((B,), 300, device=mb.dev, dtype=torch.int32)
    y = torch.rand((B, mb.V), device=mb.dev, generator=gen) < 3.4 / mb.V
    for s in range(6):
        STEP[0] = s
        o = tg.step(x, y, nf)
        bad = [(v.name, nonfin(v.data)) for v in g.trainable_variables() if not bool(torch.isfinite(v.data).all())]
        print("step %d loss %.5g nonfinite params: %s" % (s, float(o["loss"]), bad), flush=True)


main()
