"""-m gpu, round 6: DropoutWrapper(input_keep_prob) inside the native recurrent stack (csrc/lstm_stack.hip), the
persistent recurrences' register-carried state (csrc/lstm_persist.hip), the declared weight-gradient role of the generic GEMM entry."""
import ctypes

import numpy as np
import pytest
import torch

import yt8m_amd._lib as L
import yt8m_amd.ops as ops
import yt8m_amd.seq_ops as seq_ops
from yt8m_amd.variables import reset_default_graph, xavier_uniform, zeros

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("R,C,keep,offset,dyn", [(256, 128, 0.6, 0, False), (96, 1152, 0.5, 4 * 777, True), (1000, 64, 0.9, 64 * 1000, False)])
def test_h2_split_dropout_is_dropout_then_split_bit_for_bit(dev, R, C, keep, offset, dyn):
    """The mask inside the split pass is the stream yt8m_dropout_f32 draws (W/all_frame_models/lstm_memory_model.py:36-44: tf.nn.dropout on
    a layer's input): images of the dropped tensor, plain and transposed, byte for byte -- without the dropped tensor."""
    lib = L.lib()
    g = torch.Generator(device=dev).manual_seed(R + C)
    x = torch.randn((R, C), device=dev, generator=g)
    seed = 0x1234567887654321
    dropped = torch.empty_like(x)
    L.check(lib.yt8m_dropout_f32(_p(x), _p(dropped), x.numel(), keep, seed, offset, _st()))
    assert 0.3 < float((dropped == 0).float().mean()) / (1 - keep) < 3.0           # a mask really was applied
    word = ops.h2_absmax(x) if dyn else None
    scale = 1.0 if dyn else 1024.0
    nb = max(lib.yt8m_x3_image_bytes(R, C) // 3 * 2, 16), max(lib.yt8m_x3_image_bytes(C, R) // 3 * 2, 16)
    want = [torch.zeros(n, dtype=torch.uint8, device=dev) for n in nb]
    got = [torch.zeros(n, dtype=torch.uint8, device=dev) for n in nb]
    L.check(lib.yt8m_h2_split(_p(dropped), R, C, C, scale, _p(word), _p(want[0]), _p(want[1]), None, _st()))
    L.check(lib.yt8m_h2_split_dropout(_p(x), R, C, scale, _p(word), _p(got[0]), _p(got[1]), keep, seed, offset, _st()))
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def _run_stack(dev, B, F, D, H, nf, keep, seeds, native_dropout, seed=0):
    old = seq_ops.NATIVE_DROPOUT
    seq_ops.NATIVE_DROPOUT = native_dropout
    try:
        g = reset_default_graph(device=dev, seed=seed)
        g.begin_step()
        gen = torch.Generator(device=dev).manual_seed(seed)
        x = (torch.rand((F, B, D), device=dev, generator=gen) - 0.5)
        wb, d_in = [], D
        for l in range(2):
            wb.append((g.get_variable("l%d/w" % l, (d_in + H, 4 * H), xavier_uniform), g.get_variable("l%d/b" % l, (4 * H,), zeros)))
            d_in = H
        g.finalize()
        x.requires_grad_(True)
        n0 = dict(seq_ops.NATIVE_CALLS)
        out, finals = seq_ops.lstm_stack(x, nf, wb, chunks=2, input_keep_prob=keep, seeds=seeds)
        res = [out] + [t for p in finals for t in p]
        gen2 = torch.Generator(device=dev).manual_seed(7)
        sum((r * torch.rand(r.shape, device=dev, generator=gen2)).sum() for r in res).backward()
        torch.cuda.synchronize()
        went_native = seq_ops.NATIVE_CALLS["fwd"] == n0["fwd"] + 1 and seq_ops.NATIVE_CALLS["bwd"] == n0["bwd"] + 1
        P = [(w.data.detach().cpu().double(), b.data.detach().cpu().double()) for w, b in wb]
        return [r.detach().cpu() for r in res], [x.grad.detach().cpu(), g.grads.detach().cpu().clone()], x.detach().cpu().double(), P, went_native
    finally:
        seq_ops.NATIVE_DROPOUT = old


def test_dropout_wrapper_runs_on_the_native_stack_and_matches_the_oracle(dev, flags):
    """LstmMemoryModel's DropoutWrapper(BasicLSTMCell, input_keep_prob) on both layers (W/all_frame_models/lstm_memory_model.py:36-44)
    inside yt8m_lstm_stack_fwd / _bwd: the same function as the Python orchestration (dropout_f32 passes + per-call entry points) and
    as the fp64 restatement with the same Philox masks -- outputs, final states, dx, every weight and bias gradient."""
    from oracle import torch_ref
    B, F, D, H = 32, 40, 64, 128
    if not L.lib().yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available for this shape / device")
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(3), dtype=torch.int32)
    nf[0], nf[1] = F, 0
    keep, seeds = 0.6, (0xABCDEF0123456789, 0x0F1E2D3C4B5A6978)
    a, ga, x64, P, native = _run_stack(dev, B, F, D, H, nf, keep, seeds, True)
    assert native, "dropout did not stay on the native stack"
    b, gb, _, _, native_b = _run_stack(dev, B, F, D, H, nf, keep, seeds, False)
    assert not native_b
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 5e-6
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-9
    xs = x64.transpose(0, 1).clone().requires_grad_(True)
    layers = [(w.clone().requires_grad_(True), bb.clone().requires_grad_(True)) for w, bb in P]
    out, c, h = torch_ref.lstm_stack(xs, nf.cpu(), layers, dropout_spec=(keep, list(seeds)))
    ref = [out.transpose(0, 1)] + [t for pair in zip(c, h) for t in pair]
    for u, v in zip(a, ref):
        assert float((u.double() - v).abs().max()) < 2e-5
    gen2 = torch.Generator(device=dev).manual_seed(7)
    sum((r * torch.rand(r.shape, device=dev, generator=gen2).cpu().double()).sum() for r in ref).backward()
    assert float((ga[0].double() - xs.grad.transpose(0, 1)).abs().max()) <= 2e-4 * float(xs.grad.abs().max())
    flat = torch.cat([t.grad.flatten() for pair in layers for t in pair])
    assert float((ga[1].double() - flat).abs().max()) <= 2e-4 * float(flat.abs().max())
    # the masks really were applied, and per layer
    c0 = torch_ref.lstm_stack(x64.transpose(0, 1), nf.cpu(), [(w, bb) for w, bb in P])[1]
    assert float((torch.cat(c0, 1) - torch.cat(c, 1).detach()).abs().max()) > 1e-3


def test_backward_recurrence_carries_its_running_state_in_registers(dev):
    """Round 6: with one 16-row tile per epilogue wave (B = 128: the headline) the running (dh, dc) and c_t+1 travel in registers instead
    of through `work` / a second read of cs.  Same numbers, and `work` still hands (dh, dc) to the next launch: a backward pass cut into
    three launches equals the single launch bit for bit (the chunk boundaries go through memory, the steps inside through registers)."""
    lib = L.lib()
    B, F, H = 128, 24, 1024
    if not lib.yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available on this device")
    g = torch.Generator(device=dev).manual_seed(11)
    gates = torch.rand((F, B, 4 * H), device=dev, generator=g)
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.06
    cs = torch.randn((F + 1, B, H), device=dev, generator=g) * 0.5
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.01
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    nf[0], nf[1] = F, 0
    wword = ops.h2_absmax(Wh)
    nbytes = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)

    def run(parts):
        pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        work = torch.zeros((4, B, H), device=dev)
        work[0].normal_(generator=torch.Generator(device=dev).manual_seed(5))
        work[1].normal_(generator=torch.Generator(device=dev).manual_seed(6))
        dz = torch.zeros((F, B, 4 * H), device=dev)
        phase = 0
        for t0, T in parts:
            L.check(lib.yt8m_lstm_persist_bwd_h2(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), phase, None, _p(nf), t0, T, B, H,
                                                 _p(wword), _p(pws), pws.numel(), _st()))
            phase = (phase + T) % 2
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _st()))
        return dz, work[2 * phase:2 * phase + 2].clone()

    dz1, w1 = run([(0, F)])
    dz3, w3 = run([(16, 8), (5, 11), (0, 5)])
    assert torch.equal(dz1, dz3) and torch.equal(w1, w3)
    assert float(dz1.abs().max()) > 0 and bool(torch.isfinite(dz1).all())


@pytest.mark.parametrize("ragged", [False, True])
def test_k_split_pair_form_of_the_backward_recurrence_equals_the_unpaired_form(dev, ragged):
    """Round 6: lstm_persist_bwd_kernel<.., P2> -- two workgroups per 32 units, each over half of K, partial tiles handed over through
    tagged granules -- against the unpaired f16 form on the same inputs: dz per (step) on its own scale to fp32 rounding, the final
    (dh, dc) likewise, ended videos bit-exact zero; and bitwise reproducible run to run (fixed summation order across the pair)."""
    lib = L.lib()
    B, F, H = 128, 40, 1024
    if not lib.yt8m_lstm_persist_bwd_on_f16_pipe(B, H):
        pytest.skip("f16 backward recurrence not available on this device")
    g = torch.Generator(device=dev).manual_seed(21)
    gates = torch.rand((F, B, 4 * H), device=dev, generator=g)
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.06
    cs = torch.randn((F + 1, B, H), device=dev, generator=g) * 0.5
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.01
    nf = None
    if ragged:
        nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
        nf[0], nf[1] = F, 0
    wword = ops.h2_absmax(Wh)
    nbytes = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)

    def run(pair):
        L.check(lib.yt8m_lstm_persist_set_pair(pair))
        try:
            pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            work = torch.zeros((4, B, H), device=dev)
            work[0].normal_(generator=torch.Generator(device=dev).manual_seed(5))
            dz = torch.zeros((F, B, 4 * H), device=dev)
            phase = 0
            for t0, T in [(25, 15), (0, 25)]:                      # two launches: the hand-off slots and tags are re-used across launches
                L.check(lib.yt8m_lstm_persist_bwd_h2(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), phase, None, _p(nf), t0, T, B,
                                                     H, _p(wword), _p(pws), pws.numel(), _st()))
                phase = (phase + T) % 2
            torch.cuda.synchronize()
            L.check(lib.yt8m_lstm_persist_status(_p(pws), _st()))
            return dz, work[2 * phase:2 * phase + 2].clone()
        finally:
            L.check(lib.yt8m_lstm_persist_set_pair(-1))

    dz1, w1 = run(1)
    dz1b, w1b = run(1)
    dz0, w0 = run(0)
    assert torch.equal(dz1, dz1b) and torch.equal(w1, w1b)
    assert bool(torch.isfinite(dz1).all()) and float(dz1.abs().max()) > 0
    d = (dz1 - dz0).abs().amax(dim=(1, 2)) / (dz0.abs().amax(dim=(1, 2)) + 1e-30)
    assert float(d.max()) < 2e-6, float(d.max())
    assert float((w1 - w0).abs().max()) <= 2e-6 * float(w0.abs().max())
    if ragged:
        dead = torch.arange(F, device=dev)[:, None] >= nf[None, :].long()
        assert float(dz1[dead].abs().max()) == 0.0
