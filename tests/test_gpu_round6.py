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


@pytest.mark.parametrize("B,F,H,parts", [(128, 12, 1024, [(0, 12)]), (128, 9, 1024, [(0, 4), (4, 5)]), (37, 6, 256, [(0, 6)]), (70, 5, 512, [(0, 2), (2, 3)])])
def test_persistent_gru_equals_the_per_step_kernels(dev, B, F, H, parts):
    """Round 6: GRUCell as one launch per direction on the persistent recurrences' exchange protocol
    (csrc/gru_persist.inl; W/all_frame_models/gru_pooling_model.py:34-47) against the per-step kernels of csrc/cells.hip on the same
    inputs: activations r | u, c, the states, r * h and the outputs to 2e-6; dzg / dzc / the final dL/dh to 1e-5 of their maxima; rows past
    num_frames bit-exact (copied state, zero output, zero dz); chained launches over time ranges; bitwise reproducible run to run."""
    lib = L.lib()
    if not lib.yt8m_gru_persist_supported(B, H):
        pytest.skip("persistent GRU not available for this shape on this device")
    g = torch.Generator(device=dev).manual_seed(B + F + H)
    zg0 = torch.randn((F, B, 2 * H), device=dev, generator=g) * 0.8 + 0.5
    zc0 = torch.randn((F, B, H), device=dev, generator=g) * 0.8
    Wg = (torch.rand((H, 2 * H), device=dev, generator=g) - 0.5) * 0.08
    Wc = (torch.rand((H, H), device=dev, generator=g) - 0.5) * 0.08
    h0 = torch.randn((B, H), device=dev, generator=g) * 0.3
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    nf[0], nf[1], nf[2] = F, 0, 1
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.1
    dhF = torch.randn((B, H), device=dev, generator=g) * 0.1
    ws = ops._workspace(dev)

    def per_step():
        zg, zc = zg0.clone(), zc0.clone()
        hs = torch.zeros((F + 1, B, H), device=dev)
        hs[0] = h0
        rh, out = torch.zeros((F, B, H), device=dev), torch.zeros((F, B, H), device=dev)
        L.check(lib.yt8m_gru_layer_fwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(rh), _p(out), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _st()))
        dzg, dzc = torch.zeros((F, B, 2 * H), device=dev), torch.zeros((F, B, H), device=dev)
        work = torch.zeros((3, B, H), device=dev)
        L.check(lib.yt8m_gru_layer_bwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(dout), _p(dhF), _p(dzg), _p(dzc), _p(work), _p(nf),
                                       F, B, H, _p(ws), ws.numel() * 4, _st()))
        torch.cuda.synchronize()
        return zg, zc, hs, rh, out, dzg, dzc, work[F % 2].clone()

    def persistent():
        zg, zc = zg0.clone(), zc0.clone()
        hs = torch.zeros((F + 1, B, H), device=dev)
        hs[0] = h0
        rh, out = torch.zeros((F, B, H), device=dev), torch.zeros((F, B, H), device=dev)
        pws = torch.zeros(lib.yt8m_gru_persist_workspace_bytes(B, H, F), dtype=torch.uint8, device=dev)
        for t0, T in parts:
            L.check(lib.yt8m_gru_persist_fwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(rh), _p(out), _p(nf), t0, T, B, H, _p(pws),
                                             pws.numel(), _st()))
        dzg, dzc = torch.zeros((F, B, 2 * H), device=dev), torch.zeros((F, B, H), device=dev)
        work = dhF.clone()
        for t0, T in reversed(parts):
            L.check(lib.yt8m_gru_persist_bwd(_p(zg), _p(zc), _p(Wg), 2 * H, _p(Wc), H, _p(hs), _p(dout), _p(dzg), _p(dzc), _p(work), _p(nf), t0, T,
                                             B, H, _p(pws), pws.numel(), _st()))
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _st()))
        return zg, zc, hs, rh, out, dzg, dzc, work

    a = per_step()
    b = persistent()
    b2 = persistent()
    for x, y in zip(b, b2):
        assert torch.equal(x, y)
    names = ["zg", "zc", "hs", "rh", "out"]
    for n, x, y in zip(names, a[:5], b[:5]):
        assert bool(torch.isfinite(y).all()), n
        assert float((x - y).abs().max()) < 2e-6, (n, float((x - y).abs().max()))
    for n, x, y in zip(["dzg", "dzc", "dh0"], a[5:], b[5:]):
        assert float(x.abs().max()) > 0
        assert float((x - y).abs().max()) < 1e-5 * float(x.abs().max()), (n, float((x - y).abs().max()), float(x.abs().max()))
    t = torch.arange(F, device=dev)[:, None]
    dead = t >= nf[None, :].long()
    assert float(b[4][dead].abs().max()) == 0.0 and float(b[5][dead].abs().max()) == 0.0 and float(b[6][dead].abs().max()) == 0.0
    hs = b[2]
    assert torch.equal(hs[1:][dead], hs[:-1][dead])


def test_gru_layer_op_with_persistent_recurrences_matches_the_per_step_op(dev):
    """seq_ops.gru_layer with the persistent forward (default) and the opt-in persistent backward against the op on the per-step
    kernels: two stacked layers sharing one exchange workspace, outputs and every gradient."""
    from yt8m_amd.variables import ones
    F, B, D, H = 10, 128, 64, 1024
    if not L.lib().yt8m_gru_persist_supported(B, H):
        pytest.skip("persistent GRU not available on this device")
    old = seq_ops.GRU_PERSIST_FWD, seq_ops.GRU_PERSIST_BWD

    def run(fwd, bwd):
        seq_ops.GRU_PERSIST_FWD, seq_ops.GRU_PERSIST_BWD = fwd, bwd
        g = reset_default_graph(device=dev, seed=0)
        g.begin_step()
        gen = torch.Generator(device=dev).manual_seed(3)
        x = (torch.rand((F, B, D), device=dev, generator=gen) * 2 - 1).requires_grad_(True)
        nf = torch.randint(0, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)
        vs, d_in = [], D
        for l in range(2):
            vs.append((g.get_variable("l%d/wg" % l, (d_in + H, 2 * H), xavier_uniform), g.get_variable("l%d/bg" % l, (2 * H,), ones),
                       g.get_variable("l%d/wc" % l, (d_in + H, H), xavier_uniform), g.get_variable("l%d/bc" % l, (H,), zeros)))
            d_in = H
        g.finalize()
        h, finals = x, []
        for l in range(2):
            h, hf = seq_ops.gru_layer(h, *vs[l], nf)
            finals.append(hf)
        w = torch.randn((F, B, H), device=dev, generator=gen) * 0.01
        ((h * w).sum() + (finals[0] * 0.01).sum()).backward()
        torch.cuda.synchronize()
        return [h.detach().clone(), x.grad.clone()] + [v.grad.clone() for lay in vs for v in lay]

    try:
        ref = run(False, False)
        for fb in ((True, False), (True, True)):
            got = run(*fb)
            for a, b in zip(ref, got):
                assert float((a - b).abs().max()) <= 1e-5 * max(float(a.abs().max()), 1e-3), fb
    finally:
        seq_ops.GRU_PERSIST_FWD, seq_ops.GRU_PERSIST_BWD = old
    seq_ops.check_persist_errors()


def test_bf16_logits_of_the_moe_head_product_and_the_mixing_passes_that_read_them(dev):
    """Round 6 (W/all_video_models/moe_model.py:54-64 under --compute_dtype=bfloat16): the b1 product writes the logits as
    bf16 (fp32 accumulation, ONE rounding after the bias) -- bit for bit the bf16 rounding of its fp32 output; the mixing pass and the
    fused mixing backward that read bf16 logits equal the fp32-logit passes fed with those rounded logits bit for bit."""
    lib = L.lib()
    B, D, V, M = 512, 512, 1036, 2
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn((B, D), device=dev, generator=g)
    Wg = torch.randn((D, 3 * V), device=dev, generator=g) * 0.05
    We = torch.randn((D, 2 * V), device=dev, generator=g) * 0.05
    be = torch.randn((2 * V,), device=dev, generator=g) * 0.1
    xi = ops.bf16_image(x)
    WgT, WeT = ops.bf16_image(Wg, transpose=True), ops.bf16_image(We, transpose=True)
    Zg32, Ze32 = ops.gemm_b1_grouped([dict(A=xi, B=WgT), dict(A=xi, B=WeT, bias=be)])
    Zg16, Ze16 = ops.gemm_b1_grouped([dict(A=xi, B=WgT, out_dtype=torch.bfloat16), dict(A=xi, B=WeT, bias=be, out_dtype=torch.bfloat16)])
    assert Zg16.dtype == torch.bfloat16 and torch.equal(Zg16, Zg32.to(torch.bfloat16)) and torch.equal(Ze16, Ze32.to(torch.bfloat16))
    # mixing forward
    p16 = ops.moe_mix_fwd(Zg16, Ze16, V, M)
    p32 = ops.moe_mix_fwd(Zg16.float(), Ze16.float(), V, M)
    assert float((p16 - p32).abs().max()) <= 2e-7 and float(p16.min()) >= 0 and float(p16.max()) <= 1   # (same arithmetic; fma placement differs)
    # fused mixing backward on images, from dp and from labels
    kb = lambda K: (K + 15) // 16
    mk = lambda rows, K: torch.zeros(max(lib.yt8m_x3_image_bytes(rows, K) // 3, 16), dtype=torch.uint8, device=dev)
    dp = torch.randn((B, V), device=dev, generator=g) * 0.01
    y = (torch.rand((B, V), device=dev, generator=g) < 0.01).to(torch.uint8)
    for dpv, lab in ((dp, None), (None, y)):
        outs = []
        for fn, zg, ze in ((lib.yt8m_moe_mix_bwd_bf16_images_z16, Zg16, Ze16), (lib.yt8m_moe_mix_bwd_bf16_images, Zg16.float(), Ze16.float())):
            imgs = [mk(B, 3 * V), mk(3 * V, B), mk(B, 2 * V), mk(2 * V, B)]
            part = torch.zeros((lib.yt8m_moe_mix_bwd_bf16_partial_rows(B), 2 * V), device=dev)
            L.check(fn(_p(zg), _p(ze), _p(dpv), _p(lab), 0, B, V, M, 1e-6, 1.0 / B, None, _p(imgs[0]), kb(3 * V), _p(imgs[1]), kb(B), _p(imgs[2]),
                       kb(2 * V), _p(imgs[3]), kb(B), _p(part), _st()))
            outs.append(imgs + [part])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        assert int(outs[0][0].long().sum()) != 0


def test_declared_h2_role_of_the_moe_logits_product(dev):
    """Round 6: yt8m_gemm_auto_grouped with YT8M_GEMM_ROLE_H2 (the MoE head's x . [W_g | W_e], W/all_video_models/moe_model.py:43-52) runs
    three f16 products under one scale per operand matrix: against fp64 within the h2 contract (<= 4e-6 of the output scale at K = 1152,
    the six-product form's own error being ~1e-6), bias added, and bitwise reproducible; without the flag the result is the six-product form's."""
    g = torch.Generator(device=dev).manual_seed(12)
    B, D, N1, N2 = 1024, 1152, 3 * 4716, 2 * 4716
    x = torch.nn.functional.normalize(torch.rand((B, D), device=dev, generator=g) * 4 - 2, dim=1)
    Wg = (torch.rand((D, N1), device=dev, generator=g) - 0.5) * 0.09
    We = (torch.rand((D, N2), device=dev, generator=g) - 0.5) * 0.09
    be = torch.randn((N2,), device=dev, generator=g) * 0.1
    h2 = ops.gemm_grouped([dict(A=x, B=Wg), dict(A=x, B=We, bias=be)], role="h2")
    h2b = ops.gemm_grouped([dict(A=x, B=Wg), dict(A=x, B=We, bias=be)], role="h2")
    x3 = ops.gemm_grouped([dict(A=x, B=Wg), dict(A=x, B=We, bias=be)])
    ref = [x.double() @ Wg.double(), x.double() @ We.double() + be.double()]
    for a, b, c, r in zip(h2, h2b, x3, ref):
        assert torch.equal(a, b)
        scale = float(r.abs().max())
        assert float((a.double() - r).abs().max()) <= 4e-6 * scale, float((a.double() - r).abs().max()) / scale
        assert float((c.double() - r).abs().max()) <= 4e-6 * scale


def test_mixing_backward_measures_the_maxima_its_weight_gradient_products_split_under(dev, flags):
    """Round 6: yt8m_moe_mix_xent_bwd_absmax leaves max |dZg| / max |dZe| as float bits while it writes the gradients;
    yt8m_gemm_auto_grouped_ex takes them instead of two memset + absmax passes.  Same words, so the same images and the same products:
    a MoeModel training step (B = 1024, BASELINE configs[1]) bit for bit with and without."""
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm

    def run(on):
        old = ops.MIX_BWD_ABSMAX
        ops.MIX_BWD_ABSMAX = on
        try:
            flags.reset()
            B, D, V = 1024, 1152, 4716
            g = reset_default_graph(device=dev, seed=0)
            tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
            gen = torch.Generator(device=dev).manual_seed(5)
            losses = []
            for _ in range(3):
                x = torch.rand((B, D), device=dev, generator=gen) * 4.0 - 2.0
                y = torch.rand((B, V), device=dev, generator=gen) < (3.4 / V)
                losses.append(float(tg.step(x, y)["loss"]))
            torch.cuda.synchronize()
            return losses, g.params.clone(), g.adam_m.clone()
        finally:
            ops.MIX_BWD_ABSMAX = old

    a, b = run(True), run(False)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # the words themselves against the stand-alone absmax pass
    lib = L.lib()
    gen = torch.Generator(device=dev).manual_seed(6)
    Bv, V, M = 256, 1000, 2
    Zg, Ze = torch.randn((Bv, 3 * V), device=dev, generator=gen), torch.randn((Bv, 2 * V), device=dev, generator=gen)
    y = (torch.rand((Bv, V), device=dev, generator=gen) < 0.01).to(torch.uint8)
    up = torch.ones(1, device=dev)
    words = torch.empty(2, device=dev)
    L.check(lib.yt8m_moe_mix_xent_bwd_absmax(_p(Zg), _p(Ze), _p(y), 0, _p(up), Bv, V, M, 1e-6, 1.0, _p(words), _st()))
    assert float(words[0]) == float(Zg.abs().max()) and float(words[1]) == float(Ze.abs().max())


def test_resident_half_plane_images_of_the_moe_weights(dev, flags):
    """Round 6: the weights of a product that declared the h2 role get RESIDENT half-plane images (wimg.py planes = 2): the optimiser's
    tile pass rewrites them under the maximum the previous pass measured (a word in front of the image) instead of an absmax + split
    per step.  MoeModel B = 1024: the images exist from the third step on, and the run equals the one that re-splits per step -- the
    scale is the same power of two unless max |w| crossed one between two steps, so: losses to 1e-6, weights to 1e-7 absolute."""
    import yt8m_amd.train as train
    import yt8m_amd.video_level_models as vlm
    import yt8m_amd.wimg as wimg

    def run(resident):
        old = wimg.H2
        wimg.H2 = resident
        try:
            flags.reset()
            B, D, V = 1024, 1152, 4716
            g = reset_default_graph(device=dev, seed=0)
            tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
            gen = torch.Generator(device=dev).manual_seed(5)
            losses = []
            for _ in range(6):
                x = torch.rand((B, D), device=dev, generator=gen) * 4.0 - 2.0
                y = torch.rand((B, V), device=dev, generator=gen) < (3.4 / V)
                losses.append(float(tg.step(x, y)["loss"]))
            torch.cuda.synchronize()
            nh2 = sum(1 for k in (g.wimg.keys if g.wimg is not None else {}) if k[4] == 2)
            words = []
            if g.wimg is not None:
                for k, whole in g.wimg.h2_whole.items():
                    w = whole[:8].view(torch.float32).cpu()
                    v = g.trainable_variables()[k[0]]
                    words.append((float(w[0]), float(w[1]), float(v.data.abs().max())))
            return losses, g.params.clone(), nh2, words
        finally:
            wimg.H2 = old

    a = run(True)
    b = run(False)
    assert a[2] >= 2 and b[2] == 0, (a[2], b[2])
    for w0, w1, cur in a[3]:
        assert w1 == cur and 0.25 * cur < w0 < 4.0 * cur, (w0, w1, cur)     # next maximum = the weights as they are; the scale word is last step's
    assert all(abs(x - y) <= 1e-6 * abs(y) for x, y in zip(a[0], b[0])), (a[0], b[0])
    assert float((a[1] - b[1]).abs().max()) <= 1e-7


def test_large_fully_connected_forward_in_the_declared_h2_role(dev, flags):
    """Round 6: ops.linear over >= 1 024 rows declares the h2 role for its forward product (the NetVLAD hidden FC at B = 1024:
    [1024, 73728] . [73728, 1024], Appendix B of SURVEY.md): against fp64 within 4e-6 of the output scale at K = 73 728 (the six-product
    form: ~1e-6), gradients are those of the six-product run bit for bit (dx / dW keep their forms)."""
    g = reset_default_graph(device=dev, seed=0)
    g.begin_step()
    K, N, M = 73728, 1024, 1024
    W = g.get_variable("w", (K, N), xavier_uniform)
    b = g.get_variable("b", (N,), zeros)
    g.finalize()
    gen = torch.Generator(device=dev).manual_seed(3)
    x = torch.nn.functional.normalize(torch.randn((M, K), device=dev, generator=gen), dim=1)
    dy = torch.randn((M, N), device=dev, generator=gen) * 1e-3
    outs = {}
    old = ops.LINEAR_FWD_H2
    try:
        for on in (True, False):
            ops.LINEAR_FWD_H2 = on
            xr = x.clone().requires_grad_(True)
            g.grads.zero_()
            y = ops.linear(xr, W, b)
            y.backward(dy)
            torch.cuda.synchronize()
            outs[on] = (y.detach().clone(), xr.grad.clone(), W.grad.clone())
    finally:
        ops.LINEAR_FWD_H2 = old
    ref = (x[:64].double() @ W.data.double() + b.data.double())
    scale = float(ref.abs().max())
    assert float((outs[True][0][:64].double() - ref).abs().max()) <= 4e-6 * scale
    assert float((outs[False][0][:64].double() - ref).abs().max()) <= 4e-6 * scale
    assert torch.equal(outs[True][1], outs[False][1]) and torch.equal(outs[True][2], outs[False][2])
    # dx with dy split row by row (ops._linear_dx_h2_rows): rows of dy 2^-20 apart each keep fp32-grade RELATIVE precision
    dyw = dy * torch.pow(2.0, -torch.arange(M, device=dev).float() % 21)[:, None]
    olddx = ops.LINEAR_DX_H2
    try:
        got = {}
        for on in (True, False):
            ops.LINEAR_DX_H2 = on
            xr = x.clone().requires_grad_(True)
            g.grads.zero_()
            ops.linear(xr, W, b).backward(dyw)
            torch.cuda.synchronize()
            got[on] = xr.grad.clone()
    finally:
        ops.LINEAR_DX_H2 = olddx
    rows = [0, 5, 20, 41, 1023]
    refdx = dyw[rows].double() @ W.data.double().t()
    for on in (True, False):
        rel = (got[on][rows].double() - refdx).abs().amax(dim=1) / refdx.abs().amax(dim=1)
        assert float(rel.max()) <= 4e-6, (on, rel.tolist())


def test_moe_head_dx_as_row_scaled_h2_products(dev):
    """Round 6: dx = dZg . Wg^T + dZe . We^T of the MoE head (W/all_video_models/moe_model.py:40-64 through tf.gradients) from 1 024 rows on:
    two launches of ops._linear_dx_h2_rows, the second accumulating (beta = 1) -- against fp64 per row, rows 2^-16 apart."""
    g = reset_default_graph(device=dev, seed=0)
    g.begin_step()
    D, V, B = 1024, 1036, 1024
    Wg = g.get_variable("wg", (D, 3 * V), xavier_uniform)
    We = g.get_variable("we", (D, 2 * V), xavier_uniform)
    g.finalize()
    gen = torch.Generator(device=dev).manual_seed(8)
    rowmag = torch.pow(2.0, -(torch.arange(B, device=dev).float() % 17))[:, None]
    Zg = torch.randn((B, 3 * V), device=dev, generator=gen) * 1e-3 * rowmag
    Ze = torch.randn((B, 2 * V), device=dev, generator=gen) * 1e-3 * rowmag
    dx = ops._linear_dx_h2_rows(Zg, Wg)
    ops._linear_dx_h2_rows(Ze, We, out=dx, beta=1.0)
    torch.cuda.synchronize()
    rows = [0, 7, 16, 33, 1023]
    ref = Zg[rows].double() @ Wg.data.double().t() + Ze[rows].double() @ We.data.double().t()
    rel = (dx[rows].double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)
    assert float(rel.max()) <= 4e-6, rel.tolist()


def _run_frames_plugin(model, q, y, nf, dev, P=None, rs=None):
    """A frame-level plugin on RAW uint8 frames through TrainGraph with the reference's DefaultTransformer: forward (creates the
    variables), inject weights, forward + loss + backward."""
    import yt8m_amd.train as train
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(model, batch_size=q.shape[0], graph=g)
    qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    tg.forward(qd, yd, nfd)
    g.finalize()
    if P is None:
        P = {k: (rs.randn(*v.shape) * 0.3).astype(np.float32) for k, v in g.vars.items()}
    for k, v in P.items():
        g.vars[k].data.copy_(torch.from_numpy(v).to(dev).view(g.vars[k].data.shape))
    res = tg.forward(qd, yd, nfd)
    loss = tg.loss(res, yd)
    loss.backward()
    grads = {k: v.grad.detach().cpu().numpy().astype(np.float64) for k, v in g.vars.items() if v.trainable}
    return res["predictions"].detach().cpu().numpy().astype(np.float64), float(loss.detach()), grads, P


@pytest.mark.parametrize("positional", [False, True])
def test_attention_lstm_plugins_take_the_raw_uint8_frames(dev, flags, positional, monkeypatch):
    """Round 6 (last session): LstmAttentionMaxPoolingModel / LstmPositionalAttentionMaxPoolingModel declare accepts_quantized_input --
    the stack's layer-0 projection AND the attention FC on concat([x, ...]) read the reader's bytes (seq_ops.attention_logits_u8 with
    per-frame and per-video parts), no fp32 [B,F,D] tensor.  Against (a) the same plugin on the dequantised float frames (the path the
    reference's transformer feeds) with the same weights and (b) the fp64 restatement: predictions, loss, every gradient; ragged
    num_frames >= 1 (an empty video's attention weights are 0 / 0 in the reference as well)."""
    from oracle import np_ref, torch_ref
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.ops as ops_
    monkeypatch.setattr(ops_, "SKINNY_MIN_ROWS", 1)                      # the attention FC's float parts on the streaming kernels too
    rs = np.random.RandomState(11 + positional)
    B, F, D, Hh, V, A, E = 8, 12, 64, 128, 13, 8, 8
    flags.lstm_cells, flags.lstm_layers, flags.lstm_attentions, flags.positional_embedding_size = str(Hh), 2, A, E
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    y = rs.rand(B, V) < 0.2
    cls = flm.LstmPositionalAttentionMaxPoolingModel if positional else flm.LstmAttentionMaxPoolingModel
    qt = torch.from_numpy(q).to(dev)
    assert seq_ops.u8_attention_supported(qt, A) and flm._lib_u8_ok(D), "the shape of this test must take the uint8 path"
    pa, la, ga, P = _run_frames_plugin(cls(), q, y, nf, dev, rs=rs)
    monkeypatch.setattr(cls, "accepts_quantized_input", False)          # the transformer dequantises: float frames into the plugin
    pb, lb, gb, _ = _run_frames_plugin(cls(), q, y, nf, dev, P=P)
    assert set(ga) == set(gb)
    assert np.abs(pa - pb).max() < 2e-5 and abs(la - lb) < 1e-4 * max(1.0, abs(lb))
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() <= 2e-4 * max(1.0, np.abs(gb[k]).max()), k
    # fp64 restatement on the dequantised, masked, l2-normalised frames
    x64 = torch.from_numpy(np_ref.dequant_l2norm_folded(q, nf))
    tp = {k: torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for k, v in P.items()}
    layers = [(tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    head = (tp["attention-/weights"], tp["attention-/biases"], tp["gates-sub-moe/weights"], tp["experts-sub-moe/weights"],
            tp["experts-sub-moe/biases"], 2)
    if positional:
        pr = torch_ref.lstm_positional_attention_max_pooling(x64, torch.from_numpy(nf), layers, tp["positional_embedding"], *head)
    else:
        pr = torch_ref.lstm_attention_max_pooling(x64, torch.from_numpy(nf), layers, *head)
    lr = torch_ref.cross_entropy(pr, torch.from_numpy(y.astype(np.float64)))
    lr.backward()
    assert np.abs(pa - pr.detach().numpy()).max() < 1e-4
    for k, t in tp.items():
        if t.grad is not None:
            assert np.abs(ga[k] - t.grad.numpy()).max() <= 5e-4 * max(1.0, np.abs(t.grad.numpy()).max()), k


def test_parallel_lstm_plugin_takes_the_raw_uint8_frames(dev, flags, monkeypatch):
    """LstmParallelFinaloutputModel on the reader's bytes: every stack reads ITS slice of the uint8 frames (row norms of the slice in the
    projection's epilogue) -- l2_normalize(slice of l2_normalize(x)) = l2_normalize(slice of x).  A slice the uint8 projection does not
    cover (width % 8 != 0) is dequantised; both against the float path with the same weights and against the fp64 restatement
    (W/all_frame_models/lstm_parallel_finaloutput_model.py:13-73)."""
    from oracle import np_ref, torch_ref
    import yt8m_amd.frame_level_models as flm
    rs = np.random.RandomState(21)
    B, F, V = 8, 10, 13
    fsz, hsz = [64, 32, 12], [128, 128, 128]
    flags.feature_sizes, flags.lstm_cells, flags.lstm_layers = "64,32,12", "128,128,128", 2
    q = rs.randint(0, 256, size=(B, F, sum(fsz))).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1], nf[2] = F, 1, 0
    y = rs.rand(B, V) < 0.2
    cls = flm.LstmParallelFinaloutputModel
    pa, la, ga, P = _run_frames_plugin(cls(), q, y, nf, dev, rs=rs)
    monkeypatch.setattr(cls, "accepts_quantized_input", False)
    pb, lb, gb, _ = _run_frames_plugin(cls(), q, y, nf, dev, P=P)
    assert np.abs(pa - pb).max() < 2e-5 and abs(la - lb) < 1e-4 * max(1.0, abs(lb))
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() <= 2e-4 * max(1.0, np.abs(gb[k]).max()), k
    x64 = torch.from_numpy(np_ref.dequant_l2norm_folded(q, nf))
    tp = {k: torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for k, v in P.items()}
    sets = [[(tp["RNN%d/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % (i, l)],
              tp["RNN%d/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % (i, l)]) for l in range(2)] for i in range(3)]
    st = torch_ref.lstm_parallel_finaloutput(x64, torch.from_numpy(nf), sets, fsz)
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, torch.from_numpy(y.astype(np.float64)))
    lr.backward()
    assert np.abs(pa - pr.detach().numpy()).max() < 1e-4
    for k, t in tp.items():
        if t.grad is not None:
            assert np.abs(ga[k] - t.grad.numpy()).max() <= 5e-4 * max(1.0, np.abs(t.grad.numpy()).max()), k


def test_cnn_chain_plugin_takes_the_raw_uint8_frames(dev, flags, monkeypatch):
    """CnnDeepCombineChainModel on the reader's bytes (seq_ops.u8_cnn): every (filter, shift) product of the einsum CNN reads the same
    half image of the frames at a row offset, the weight gradients come from the transposed byte image -- against the float path (the
    concatenated shifted inputs through ops.linear) with the same weights, and against the fp64 restatement
    (W/all_frame_models/cnn_deep_combine_chain_model.py:10-140) under the multitask loss; ragged num_frames >= 1, F > the filter lengths."""
    from oracle import np_ref, torch_ref
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.losses as losses
    import yt8m_amd.train as train
    rs = np.random.RandomState(31)
    B, F, D, V, Lc, cells = 16, 7, 32, 12, 2, 8
    flags.deep_chain_layers, flags.deep_chain_relu_cells = Lc, cells
    flags.support_type = ",".join(["label"] * Lc)
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    y = rs.rand(B, V) < 0.2
    qt = torch.from_numpy(q).to(dev)
    assert seq_ops.u8_cnn_supported(qt) and seq_ops.u8_attention_supported(qt, 1), "the shape of this test must take the uint8 path"

    def run(model, P=None):
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(model, batch_size=B, graph=g, multitask=True, label_loss_fn=losses.MultiTaskCrossEntropyLoss())
        qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
        tg.forward(qd, yd, nfd)
        g.finalize()
        if P is None:
            P = {k: (rs.randn(*v.shape) * 0.3).astype(np.float32) for k, v in g.vars.items()}
        for k, v in P.items():
            g.vars[k].data.copy_(torch.from_numpy(v).to(dev).view(g.vars[k].data.shape))
        res = tg.forward(qd, yd, nfd)
        loss = tg.loss(res, yd)
        loss.backward()
        grads = {k: v.grad.detach().cpu().numpy().astype(np.float64) for k, v in g.vars.items() if v.trainable}
        return (res["predictions"].detach().cpu().numpy().astype(np.float64), res["support_predictions"].detach().cpu().numpy().astype(np.float64),
                float(loss.detach()), grads, P)

    cls = flm.CnnDeepCombineChainModel
    pa, sa, la, ga, P = run(cls())
    monkeypatch.setattr(cls, "accepts_quantized_input", False)
    pb, sb, lb, gb, _ = run(cls(), P)
    assert set(ga) == set(gb)
    assert np.abs(pa - pb).max() < 2e-5 and np.abs(sa - sb).max() < 2e-5 and abs(la - lb) < 1e-4 * max(1.0, abs(lb))
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() <= 2e-4 * max(1.0, np.abs(gb[k]).max()), k
    x64 = torch.from_numpy(np_ref.dequant_l2norm_folded(q, nf))
    tp = {k: torch.from_numpy(v.astype(np.float64)).requires_grad_(True) for k, v in P.items()}
    main, sup = torch_ref.cnn_deep_combine_chain(x64, torch.from_numpy(nf), tp, Lc, 2, cells)
    assert np.abs(pa - main.detach().numpy()).max() < 1e-4 and np.abs(sa - sup.detach().numpy()).max() < 1e-4
    sp = flags.support_loss_percent
    y64 = torch.from_numpy(y.astype(np.float64))
    lr = (1 - sp) * torch_ref.cross_entropy(main, y64) + sp * torch_ref.cross_entropy(sup, y64.repeat(1, Lc))
    lr.backward()
    assert abs(la - lr.item()) < 1e-4 * abs(lr.item())
    for k, t in tp.items():
        if t.grad is not None:
            assert np.abs(ga[k] - t.grad.numpy()).max() <= 5e-4 * max(1.0, np.abs(t.grad.numpy()).max()), k


@pytest.mark.parametrize("F,shapes", [(9, [(1, 8), (2, 8), (3, 12)]), (9, [(1, 32), (2, 64), (3, 32)]),   # (filter length, columns); the
                                      (2, [(1, 32), (2, 32), (4, 64)]), (1, [(1, 8), (3, 8)])])            # 32-multiples: one-product form;
def test_pooled_u8_cnn_equals_the_pooled_output_of_the_unpooled_op(dev, F, shapes):                        # F below the filter length
    """seq_ops.u8_cnn_maxpool (time-major pooling with the argmax kept, per-column gathered weight gradient: csrc/cnn_pool.hip) against
    seq_ops.u8_cnn followed by a max over the frames (dense weight-gradient products on the transposed byte image), and both against fp64:
    pooled values, and the filters' gradients of a random linear functional of them.  Ragged videos incl. an empty one and one frame."""
    from oracle import np_ref
    rs = np.random.RandomState(41)
    B, D = 32, 48
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1], nf[2] = F, 1, 0
    qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)
    assert seq_ops.u8_cnn_supported(qd)
    Ws = [(rs.randn(fs * D, n) * 0.1).astype(np.float32) for fs, n in shapes]
    coef = rs.randn(B, sum(n for _, n in shapes)).astype(np.float32)
    out = {}
    for mode in ("pooled", "dense"):
        g = reset_default_graph(device=dev, seed=0)
        fv = [g.get_variable("f%d" % k, W.shape, zeros) for k, W in enumerate(Ws)]
        g.finalize()
        for v, W in zip(fv, Ws):
            v.data.copy_(torch.from_numpy(W).to(dev))
        g.begin_step()
        frames = seq_ops.U8FrameImages(qd, nfd)
        if mode == "pooled":
            p = seq_ops.u8_cnn_maxpool(frames, fv)
        else:
            p = seq_ops.u8_cnn(frames, fv).max(dim=1).values
        (p * torch.from_numpy(coef).to(dev)).sum().backward()
        out[mode] = (p.detach().cpu().numpy().astype(np.float64), [v.grad.detach().cpu().numpy().astype(np.float64) for v in fv])
    assert np.abs(out["pooled"][0] - out["dense"][0]).max() < 1e-6
    for a, b in zip(out["pooled"][1], out["dense"][1]):
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max())
    # fp64: concat of the shifted inputs, einsum, max over ALL frames
    x = torch.from_numpy(np_ref.dequant_l2norm_folded(q, nf))
    tw = [torch.from_numpy(W.astype(np.float64)).requires_grad_(True) for W in Ws]
    cols = []
    for (fs, n), W in zip(shapes, tw):
        sh = [x] + [torch.cat([x.new_zeros(B, min(i, F), D), x[:, :max(F - i, 0)]], dim=1) for i in range(1, fs)]
        cols.append(torch.cat(sh, dim=2) @ W)
    pr = torch.cat(cols, dim=2).max(dim=1).values
    (pr * torch.from_numpy(coef.astype(np.float64))).sum().backward()
    assert np.abs(out["pooled"][0] - pr.detach().numpy()).max() < 2e-6 * max(1.0, float(pr.detach().abs().max()))
    for a, t in zip(out["pooled"][1], tw):
        assert np.abs(a - t.grad.numpy()).max() <= 5e-6 * max(1.0, float(t.grad.abs().max()))


@pytest.mark.parametrize("B", [64, 512])
def test_moe_head_skips_the_input_gradient_of_its_data_columns(dev, flags, B):
    """ops.moe_head(dx_from=k): the first k columns of the head's input are data (the chain models keep the model input in front of what
    they learned, W/all_video_models/deep_combine_chain_model.py:66-70); dx is computed for columns [k, K) only -- the row window [k, K)
    of the weights (whole 32-row groups of the resident half-plane image at 512 rows, a row slice on the fp32 kernel at 64).  The
    learned part's gradient and every parameter gradient equal the full computation's; the data columns of dx are zeros."""
    rs = np.random.RandomState(51 + B)
    K, k0, V, M = 1152 + 128, 1152, 24, 2
    x = torch.from_numpy(rs.randn(B, K).astype(np.float32)).to(dev)
    coef = torch.from_numpy(rs.randn(B, V).astype(np.float32)).to(dev)
    res = {}
    for dx_from in (0, k0):
        g = reset_default_graph(device=dev, seed=0)
        Wg = g.get_variable("g", (K, V * (M + 1)), xavier_uniform)
        We = g.get_variable("e", (K, V * M), xavier_uniform)
        be = g.get_variable("b", (V * M,), zeros)
        g.finalize()
        g.begin_step()
        data = x[:, :k0].clone()
        learned = x[:, k0:].clone().requires_grad_(True)
        p = ops.moe_head(torch.cat([data, learned], dim=1), Wg, We, be, V, M, dx_from=dx_from)
        (p * coef).sum().backward()
        res[dx_from] = [t.detach().cpu().numpy().astype(np.float64) for t in (p, learned.grad, Wg.grad, We.grad, be.grad)]
    for a, b in zip(res[0], res[k0]):
        assert np.abs(a - b).max() <= 1e-6 * max(1e-6, np.abs(a).max())
    # the op's own dx: zeros in the data columns
    g = reset_default_graph(device=dev, seed=0)
    Wg = g.get_variable("g", (K, V * (M + 1)), xavier_uniform)
    We = g.get_variable("e", (K, V * M), xavier_uniform)
    be = g.get_variable("b", (V * M,), zeros)
    g.finalize()
    g.begin_step()
    xx = x.clone().requires_grad_(True)
    (ops.moe_head(xx, Wg, We, be, V, M, dx_from=k0) * coef).sum().backward()
    assert float(xx.grad[:, :k0].abs().max()) == 0.0 and float(xx.grad[:, k0:].abs().max()) > 0.0


@pytest.mark.parametrize("which", ["gru_pool", "gru_with_pool", "ln_lstm", "frame_logistic"])
def test_gru_and_layernorm_lstm_plugins_take_the_raw_uint8_frames(dev, flags, which, monkeypatch):
    """GruPoolingModel / GruWithPoolingModel / LayerNormLstmMemoryModel on the reader's bytes: layer 0's hoisted input projection and
    its weight gradient read the byte images (seq_ops.u8_hoisted_fwd / _dw, the forms of the native LSTM stack's layer 0), no dx for
    the data.  Against the same plugin on the dequantised float frames with the same weights: predictions, loss, every gradient."""
    import yt8m_amd.frame_level_models as flm
    rs = np.random.RandomState(61)
    B, F, D, Hh, V = 16, 8, 32, 128, 13
    flags.gru_cells, flags.gru_layers, flags.lstm_cells, flags.lstm_layers = Hh, 2, str(Hh), 2
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    y = rs.rand(B, V) < 0.2
    cls = {"gru_pool": flm.GruPoolingModel, "gru_with_pool": flm.GruWithPoolingModel, "ln_lstm": flm.LayerNormLstmMemoryModel,
           "frame_logistic": flm.FrameLevelLogisticModel}[which]        # (the frame-level logistic model: its average over the frames)
    assert seq_ops.u8_hoisted_supported(torch.from_numpy(q).to(dev)) and seq_ops.u8_attention_supported(torch.from_numpy(q).to(dev), 1)
    P0 = None
    if which == "ln_lstm":                                              # gammas near 1 (a random gamma ~ N(0, 0.3) is a degenerate cell)
        pa0, _, _, P0 = _run_frames_plugin(cls(), q, y, nf, dev, rs=rs)
        P0 = {k: (np.abs(v) + 0.5).astype(np.float32) if k.endswith("gamma") else v for k, v in P0.items()}
    pa, la, ga, P = _run_frames_plugin(cls(), q, y, nf, dev, P=P0, rs=rs)
    monkeypatch.setattr(cls, "accepts_quantized_input", False)
    pb, lb, gb, _ = _run_frames_plugin(cls(), q, y, nf, dev, P=P)
    assert set(ga) == set(gb)
    assert np.abs(pa - pb).max() < 2e-5 and abs(la - lb) < 1e-4 * max(1.0, abs(lb))
    for k in ga:
        assert np.abs(ga[k] - gb[k]).max() <= 2e-4 * max(1.0, np.abs(gb[k]).max()), k
