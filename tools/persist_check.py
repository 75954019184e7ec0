"""Persistent LSTM recurrence vs the per-step kernels: max differences + timings (tuning aid; the pytest form lives in
tests/test_gpu_round2.py).  usage: python tools/persist_check.py [time]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
import yt8m_amd.seq_ops as seq_ops  # noqa: E402
from yt8m_amd.variables import reset_default_graph, xavier_uniform, zeros  # noqa: E402

dev = torch.device("cuda:0")
seq_ops.PERSIST_CHECK = True
seq_ops.PERSIST_FWD_CHUNKS = seq_ops.PERSIST_BWD_CHUNKS = 0       # this tool times / checks the partition it is given
MODE = 1 if os.environ.get("PCHECK_FWD_ONLY") else 2      # 1: persistent forward only, 2: forward + backward


def run(B, F, D, H, L_, chunks, nf, persist, seed=0, backward=True):
    seq_ops.PERSIST = bool(persist)
    seq_ops.PERSIST_BWD = persist == 2 or persist is True
    g = reset_default_graph(device=dev, seed=seed)
    g.begin_step()
    gen = torch.Generator(device=dev).manual_seed(seed)
    x = (torch.rand((F, B, D), device=dev, generator=gen) - 0.5)
    wb, d_in = [], D
    for l in range(L_):
        W = g.get_variable("l%d/w" % l, (d_in + H, 4 * H), xavier_uniform)
        b = g.get_variable("l%d/b" % l, (4 * H,), zeros)
        wb.append((W, b))
        d_in = H
    g.finalize()
    x.requires_grad_(True)
    out, finals = seq_ops.lstm_stack(x, nf, wb, chunks=chunks)
    res = [out] + [t for p in finals for t in p]
    grads = None
    if backward:
        gen2 = torch.Generator(device=dev).manual_seed(7)
        loss = sum((r * torch.rand(r.shape, device=dev, generator=gen2)).sum() for r in res)
        loss.backward()
        grads = [x.grad.clone(), g.grads.clone()]
    torch.cuda.synchronize()
    return [r.detach().clone() for r in res], grads


def compare(tag, B, F, D, H, L_, chunks, ragged):
    nf = None
    if ragged:
        nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(3), dtype=torch.int32)
        nf[0] = F
        if B > 1:
            nf[1] = 0
    a, ga = run(B, F, D, H, L_, chunks, nf, MODE)
    b, gb = run(B, F, D, H, L_, chunks, nf, False)
    md = max(float((u - v).abs().max()) for u, v in zip(a, b))
    mg = max(float((u - v).abs().max() / (v.abs().max() + 1e-30)) for u, v in zip(ga, gb))
    print("%-28s B=%4d F=%3d H=%4d L=%d chunks=%d ragged=%d  max|fwd diff| %.3g  max rel grad diff %.3g  %s"
          % (tag, B, F, H, L_, chunks, ragged, md, mg, "OK" if md < 2e-5 and mg < 2e-4 else "MISMATCH"), flush=True)


def timing(B=128, F=300, D=1152, H=1024, L_=2):
    for persist in (True, False):
        for chunks in (1, 4):
            for it in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(B, F, D, H, L_, chunks, None, persist, backward=True)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            print("persist=%d chunks=%d: fwd+bwd wall %.2f ms (includes setup)" % (persist, chunks, el * 1e3), flush=True)
    # kernel-only timing of the forward recurrence, one layer
    lib = L.lib()
    import ctypes
    from yt8m_amd.ops import _p, _stream
    z0 = torch.randn((F, B, 4 * H), device=dev) * 0.3
    Wh = (torch.rand((H, 4 * H), device=dev) - 0.5) * 0.06
    cs = torch.zeros((F + 1, B, H), device=dev)
    hs = torch.zeros((F + 1, B, H), device=dev)
    out = torch.empty((F, B, H), device=dev)
    wword = torch.zeros(64, dtype=torch.int32, device=dev)
    L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
    ref = None
    for it in range(9):
        steps = it >= 3                                 # one exchange image per step (XCD-L2-shared fetch) vs two alternating ones
        h2 = it >= 6                                    # the recurrent product as three f16 products (yt8m_lstm_persist_fwd_h2)
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F) if steps else lib.yt8m_lstm_persist_workspace_bytes(B, H),
                          dtype=torch.uint8, device=dev)
        z = z0.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if h2:
            L.check(lib.yt8m_lstm_persist_fwd_h2(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(wword), _p(pws),
                                                 pws.numel(), _stream()))
        else:
            L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(pws), pws.numel(),
                                              _stream()))
        e1.record()
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        note = ""
        if steps and not h2:
            ref = (out.clone(), cs.clone())
        if h2 and ref is not None:
            note = "  max |out - out_x3| %.3g, max |c - c_x3| / max |c| %.3g" % (
                float((out - ref[0]).abs().max()), float((cs - ref[1]).abs().max() / ref[1].abs().max()))
        print("persistent fwd kernel (%s): %.3f ms for %d steps = %.2f us/step%s"
              % ("image per step, recurrent product as three f16 products" if h2 else
                 "image per step, recurrent product as six bf16 products" if steps and lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, H)
                 else ("image per step" if steps else "two images, fp32 MFMA"), e0.elapsed_time(e1), F, e0.elapsed_time(e1) * 1e3 / F, note),
              flush=True)


def timing_bwd(B=int(os.environ.get("PCHECK_B", "128")), F=int(os.environ.get("PCHECK_F", "300")), H=1024):
    lib = L.lib()
    from yt8m_amd.ops import _p, _stream
    gates = torch.rand((F, B, 4 * H), device=dev)
    Wh = (torch.rand((H, 4 * H), device=dev) - 0.5) * 0.06
    cs = torch.randn((F + 1, B, H), device=dev) * 0.5
    dz = torch.empty((F, B, 4 * H), device=dev)
    dout = torch.randn((F, B, H), device=dev) * 0.01
    wword = torch.zeros(64, dtype=torch.int32, device=dev)
    L.check(lib.yt8m_h2_absmax(_p(Wh), H, 4 * H, 4 * H, _p(wword), _stream()))
    ref = None
    for it in range(12):
        steps = it >= 3
        bf16 = 6 <= it < 9                                # the recurrent product on one bf16 plane (yt8m_lstm_persist_bwd_bf16)
        h2 = it >= 9                                      # ... as three f16 products of two-half-plane splits (yt8m_lstm_persist_bwd_h2)
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F) if steps else lib.yt8m_lstm_persist_workspace_bytes(B, H),
                          dtype=torch.uint8, device=dev)
        work = torch.zeros((4, B, H), device=dev)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if h2:
            L.check(lib.yt8m_lstm_persist_bwd_h2(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H,
                                                 _p(wword), _p(pws), pws.numel(), _stream()))
        else:
            L.check((lib.yt8m_lstm_persist_bwd_bf16 if bf16 else lib.yt8m_lstm_persist_bwd)(
                _p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H, _p(pws), pws.numel(), _stream()))
        e1.record()
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        note = ""
        if steps and not bf16 and not h2:
            ref = (dz.clone(), work.clone())
        if h2 and ref is not None:                        # against the fp32-pipe form of the same launch, on each time step's own scale
            d = (dz - ref[0]).abs().amax(dim=(1, 2)) / (ref[0].abs().amax(dim=(1, 2)) + 1e-30)
            note = "  max over steps of |dz - dz_fp32| / max|dz_t| = %.3g (step %d), final dh %.3g" % (
                float(d.max()), int(d.argmax()), float((work - ref[1]).abs().max() / (ref[1].abs().max() + 1e-30)))
        print("persistent bwd kernel (%s): %.3f ms for %d steps = %.2f us/step%s"
              % (("image per step, three f16 products" if h2 else "image per step, bf16 recurrent product" if bf16 else "image per step")
                 if steps else "two images", e0.elapsed_time(e1), F, e0.elapsed_time(e1) * 1e3 / F, note), flush=True)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":           # stand-alone backward kernel only (A/B of kernel variants)
        timing_bwd()
        sys.exit(0)
    compare("audio stack H=128", 128, 12, 128, 128, 2, 2, 1)
    compare("audio stack H=128 b=40", 40, 7, 128, 128, 2, 1, 1)
    compare("small cold H=256", 8, 9, 64, 256, 2, 1, 1)
    compare("pad rows H=512", 50, 12, 96, 512, 2, 2, 1)
    compare("headline shape short", 128, 16, 1152, 1024, 2, 1, 0)
    compare("headline ragged chunks=4", 128, 24, 1152, 1024, 2, 4, 1)
    compare("big batch", 512, 6, 128, 1024, 1, 1, 1)
    if len(sys.argv) > 1:
        timing_bwd()
        timing()
