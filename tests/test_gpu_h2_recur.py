"""-m gpu, round 5: the backward recurrence's product dz_t . W_h^T as three f16 products of two-half-plane splits
(csrc/lstm_persist.hip lstm_persist_bwd_kernel<.., H2>, yt8m_lstm_persist_bwd_h2) against the fp32-pipe form of the same launch
(yt8m_lstm_persist_bwd, itself pinned to fp64 autograd by tests/test_gpu_round3.py) and against an fp64 restatement of the recurrence
(W/all_frame_models/lstm_model.py:34-47 through tf.gradients)."""
import ctypes

import pytest
import torch

import yt8m_amd._lib as _lib
from yt8m_amd.ops import _p, _stream

pytestmark = pytest.mark.gpu


def _inputs(dev, B, F, H, seed, decades):
    g = torch.Generator(device=dev).manual_seed(seed)
    gates = torch.rand((F, B, 4 * H), device=dev, generator=g)
    gates[:, :, H:2 * H] = gates[:, :, H:2 * H] * 2 - 1                     # the tanh gate
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.08
    cs = torch.randn((F + 1, B, H), device=dev, generator=g) * 0.7
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.01
    if decades:                                                             # rows whose gradients differ by decades; a few all-zero ones
        row = 10.0 ** (-decades * torch.rand((B,), device=dev, generator=g))
        row[::7] = 0.0
        dout = dout * row[None, :, None]
    nf = torch.randint(1, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    nf[0], nf[1] = F, 1
    return gates, Wh, cs, dout, nf


def _run(lib, which, gates, Wh, cs, dout, nf, t0, T, wword=None):
    F, B, H4 = gates.shape
    H = H4 // 4
    dev = gates.device
    dz = torch.zeros((F, B, 4 * H), device=dev)
    work = torch.zeros((4, B, H), device=dev)
    g = torch.Generator(device=dev).manual_seed(99)
    work[0] = torch.randn((B, H), device=dev, generator=g) * 0.02          # running (dh, dc) handed in by the caller
    work[1] = torch.randn((B, H), device=dev, generator=g) * 0.02
    pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, T), dtype=torch.uint8, device=dev)
    args = [_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, _p(nf), t0, T, B, H]
    if which == "h2":
        _lib.check(lib.yt8m_lstm_persist_bwd_h2(*args, _p(wword), _p(pws), pws.numel(), _stream()))
    else:
        _lib.check(lib.yt8m_lstm_persist_bwd(*args, _p(pws), pws.numel(), _stream()))
    torch.cuda.synchronize()
    _lib.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
    return dz, work


def _absmax_word(lib, Wh):
    H, H4 = Wh.shape
    word = torch.zeros(64, dtype=torch.int32, device=Wh.device)
    _lib.check(lib.yt8m_h2_absmax(_p(Wh), H, H4, H4, _p(word), _stream()))
    return word


@pytest.mark.parametrize("B,F,H,decades", [(128, 24, 1024, 0), (128, 40, 1024, 6), (256, 12, 512, 3), (256, 9, 1024, 0)])
def test_f16_form_of_the_backward_recurrence_equals_the_fp32_form_to_fp32_rounding(dev, B, F, H, decades):
    lib = _lib.lib()
    assert lib.yt8m_lstm_persist_bwd_on_f16_pipe(B, H) == 1
    gates, Wh, cs, dout, nf = _inputs(dev, B, F, H, B + F + H, decades)
    word = _absmax_word(lib, Wh)
    t0, T = 2, F - 2                                                       # a part that does not start at frame 0
    dz32, w32 = _run(lib, "f32", gates, Wh, cs, dout, nf, t0, T)
    dzh, wh = _run(lib, "h2", gates, Wh, cs, dout, nf, t0, T, word)
    assert torch.equal(dzh[:t0], dz32[:t0])                                # untouched frames
    # every (frame, row) on its OWN scale: a row whose gradient is six decades below its neighbours' keeps fp32-grade digits
    num = (dzh - dz32).abs().amax(dim=2)
    den = dz32.abs().amax(dim=2)
    live = den > 0
    assert float((num[live] / den[live]).max()) < 3e-6
    assert torch.equal(dzh[~live[:, :, None].expand_as(dzh)], dz32[~live[:, :, None].expand_as(dz32)])   # ended videos / zero rows: exact zeros
    half = (0 + T) & 1                                                    # the running (dh, dc) handed back
    for k in (2 * half, 2 * half + 1):
        rown = (wh[k] - w32[k]).abs().amax(dim=1)
        rowd = w32[k].abs().amax(dim=1)
        ok = rowd > 0
        assert float((rown[ok] / rowd[ok]).max()) < 3e-6


def test_f16_form_against_fp64_restatement(dev):
    """The recurrence restated in fp64 on the host arithmetic of torch (same saved activations): the f16 form's error is of the size of
    the fp32 form's, both far inside the tolerance the fp32 stack is tested to."""
    lib = _lib.lib()
    B, F, H = 128, 16, 1024
    gates, Wh, cs, dout, nf = _inputs(dev, B, F, H, 5, 0)
    word = _absmax_word(lib, Wh)
    dz32, w32 = _run(lib, "f32", gates, Wh, cs, dout, nf, 0, F)
    dzh, wh = _run(lib, "h2", gates, Wh, cs, dout, nf, 0, F, word)
    g64, W64, c64, d64 = gates.double(), Wh.double(), cs.double(), dout.double()
    gen = torch.Generator(device=dev).manual_seed(99)
    dh = (torch.randn((B, H), device=dev, generator=gen) * 0.02).double()
    dc = (torch.randn((B, H), device=dev, generator=gen) * 0.02).double()
    ref = torch.zeros_like(g64)
    for t in range(F - 1, -1, -1):
        live = (t < nf.long())[:, None]
        gi, gj, gf, go = g64[t, :, :H], g64[t, :, H:2 * H], g64[t, :, 2 * H:3 * H], g64[t, :, 3 * H:]
        tc = torch.tanh(c64[t + 1])
        dht = dh + d64[t]
        dct = dc + dht * go * (1 - tc * tc)
        dz = torch.cat([dct * gj * gi * (1 - gi), dct * gi * (1 - gj * gj), dct * c64[t] * gf * (1 - gf), dht * tc * go * (1 - go)], dim=1)
        dz = torch.where(live, dz, torch.zeros_like(dz))
        ref[t] = dz
        dc = torch.where(live, dct * gf, dc)
        dh = torch.where(live, dz @ W64.t(), dh)
    scale = ref.abs().amax(dim=2) + 1e-300
    e32 = float(((dz32.double() - ref).abs().amax(dim=2) / scale).max())
    eh = float(((dzh.double() - ref).abs().amax(dim=2) / scale).max())
    assert e32 < 2e-5 and eh < 2e-5 and eh < 4 * e32 + 1e-6, (e32, eh)


def test_f16_form_is_a_permission(dev, monkeypatch):
    """Shapes / workspaces that cannot take the f16 form run the fp32 form: bit-identical results."""
    lib = _lib.lib()
    B, F, H = 48, 6, 256                                                   # H = 256: never the f16 form
    assert lib.yt8m_lstm_persist_bwd_on_f16_pipe(B, H) == 0
    gates, Wh, cs, dout, nf = _inputs(dev, B, F, H, 3, 0)
    word = _absmax_word(lib, Wh)
    a = _run(lib, "f32", gates, Wh, cs, dout, nf, 0, F)
    b = _run(lib, "h2", gates, Wh, cs, dout, nf, 0, F, word)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("f16", [False, True])
def test_recurrence_measures_the_row_and_launch_maxima_of_dz(dev, f16):
    """yt8m_lstm_persist_bwd_ex: rowmax[t B + b] and the launch's word equal max |dz| of what the launch stored -- bit for bit, so the
    scales derived from them are the ones yt8m_h2_rowscales / yt8m_h2_absmax would derive in their passes over dz -- and dz itself is
    what the plain entry point produces."""
    lib = _lib.lib()
    B, F, H = 128, 20, 1024
    assert lib.yt8m_lstm_persist_bwd_images_rows(B, H) > 0
    gates, Wh, cs, dout, nf = _inputs(dev, B, F, H, 17, 4)
    word = _absmax_word(lib, Wh)
    t0, T = 4, F - 4
    ref = _run(lib, "h2" if f16 else "f32", gates, Wh, cs, dout, nf, t0, T, word)
    dz = torch.zeros((F, B, 4 * H), device=dev)
    work = torch.zeros((4, B, H), device=dev)
    g = torch.Generator(device=dev).manual_seed(99)
    work[0] = torch.randn((B, H), device=dev, generator=g) * 0.02
    work[1] = torch.randn((B, H), device=dev, generator=g) * 0.02
    pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, T), dtype=torch.uint8, device=dev)
    rowmax = torch.zeros(F * B, dtype=torch.int32, device=dev)
    part = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.check(lib.yt8m_lstm_persist_bwd_ex(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, _p(nf), t0, T, B, H,
                                            _p(word) if f16 else None, _p(rowmax), _p(part), _p(pws), pws.numel(), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(dz, ref[0]) and torch.equal(work, ref[1])
    want = dz.abs().amax(dim=2).reshape(-1)
    assert torch.equal(rowmax.view(torch.float32), want)
    assert float(part.view(torch.float32)[0]) == float(want.max())
    assert int((want == 0).sum()) > 0                                       # ended videos / zero rows were part of the case


def test_split_rowmax_equals_rowscales_plus_split_rows(dev):
    lib = _lib.lib()
    R, C = 640, 4096
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn((R, C), device=dev, generator=g) * (10.0 ** (-6 * torch.rand((R, 1), device=dev, generator=g)))
    x[5] = 0.0
    nbytes = lib.yt8m_x3_image_bytes(R, C) // 3 * 2
    S, inv = torch.empty(R, device=dev), torch.empty(R, device=dev)
    a = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.yt8m_h2_rowscales(_p(x), R, C, C, _p(S), _p(inv), _stream()))
    _lib.check(lib.yt8m_h2_split_rows(_p(x), R, C, C, _p(S), _p(a), _stream()))
    rowmax = x.abs().amax(dim=1).contiguous()
    b = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    inv2 = torch.empty(R, device=dev)
    _lib.check(lib.yt8m_h2_split_rowmax(_p(x), R, C, C, _p(rowmax), _p(inv2), _p(b), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(inv, inv2)


def test_f16_form_of_the_forward_recurrence_equals_the_six_product_form_to_fp32_rounding(dev):
    """yt8m_lstm_persist_fwd_h2 (opt-in in the native stack): h_t and W_h as two half planes, three products -- against the six-product
    bf16 split of the same launch (pinned to fp64 autograd by tests/test_gpu_round3.py), ragged rows included."""
    lib = _lib.lib()
    B, F, H = 128, 40, 1024
    assert lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, H) == 1
    g = torch.Generator(device=dev).manual_seed(8)
    z0 = torch.randn((F, B, 4 * H), device=dev, generator=g) * 0.4
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.08
    nf = torch.randint(1, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    word = _absmax_word(lib, Wh)
    res = []
    for f16 in (False, True):
        z = z0.clone()
        cs, hs = torch.zeros((F + 1, B, H), device=dev), torch.zeros((F + 1, B, H), device=dev)
        out = torch.empty((F, B, H), device=dev)
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F), dtype=torch.uint8, device=dev)
        head = [_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nf), 0, F, B, H, 1.0]
        if f16:
            _lib.check(lib.yt8m_lstm_persist_fwd_h2(*head, _p(word), _p(pws), pws.numel(), _stream()))
        else:
            _lib.check(lib.yt8m_lstm_persist_fwd(*head, _p(pws), pws.numel(), _stream()))
        torch.cuda.synchronize()
        _lib.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        res.append((z, cs, hs, out))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(a.abs().max()))
