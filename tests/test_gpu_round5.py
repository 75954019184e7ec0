"""-m gpu, round 5: operand images of the weight matrices kept current by the optimiser pass (csrc/optim.hip adam_tile_kernel,
csrc/wimg.hip, youtube-8m_amd/wimg.py; VERDICT r4 #3a).  The tile pass must (a) leave bitwise the weights / Adam slots the chunk pass
leaves (tf.train.AdamOptimizer after per-tensor clip, W/train.py:459-466, W/utils.py:164-174) and (b) leave byte for byte the images
the split pass makes of the updated weights -- so a training run with resident images is bitwise the run that re-splits."""
import ctypes

import numpy as np
import pytest
import torch

import yt8m_amd._lib as L
import yt8m_amd.frame_level_models as flm
import yt8m_amd.ops as ops
import yt8m_amd.train as train
import yt8m_amd.video_level_models as vlm
import yt8m_amd.wimg as wimg
from yt8m_amd.ops import _p, _stream
from yt8m_amd.variables import Graph, random_normal, reset_default_graph

pytestmark = pytest.mark.gpu


def _image_of(src, planes, trans, scale):
    """What yt8m_x3_split / yt8m_bf16_image make of fp32 src [R, C]: the reference bytes."""
    lib = L.lib()
    R, C = src.shape
    rows, K = (C, R) if trans else (R, C)
    buf = torch.zeros(lib.yt8m_x3_image_bytes(rows, K) // 3 * planes, dtype=torch.uint8, device=src.device)
    fn = lib.yt8m_x3_split if planes == 3 else lib.yt8m_bf16_image
    L.check(fn(_p(src), R, C, C, float(scale), None if trans else _p(buf), _p(buf) if trans else None, _stream()))
    return buf


def test_adam_tiles_equal_the_chunk_pass_and_the_split_pass_bit_for_bit(dev):
    alpha = float(np.float32(4.0 / 255.0))
    shapes = {"a": (200, 136), "vec": (1000,), "b": (130, 77), "c": (192, 64)}       # b: odd row pitch (scalar path), rows % 64 != 0

    def make():
        g = Graph(device=dev, seed=3)
        g.begin_step()
        for name, shp in shapes.items():
            g.get_variable(name, shp, random_normal(0.3), l2=1e-3 if name != "vec" else 0.0)
        g.finalize()
        gen = torch.Generator(device=dev).manual_seed(5)
        g.grads.copy_(torch.randn(g.grads.shape, device=dev, generator=gen) * 0.1)
        g.adam_m.copy_(torch.randn(g.grads.shape, device=dev, generator=gen) * 0.01)
        g.adam_v.copy_(torch.rand(g.grads.shape, device=dev, generator=gen) * 0.01)
        return g

    ref, got = make(), make()
    assert torch.equal(ref.params, got.params) and got.wimg is not None and not got.wimg.active
    ia, ib, ic = got.vars["a"].index, got.vars["b"].index, got.vars["c"].index
    keys = [(ia, 0, 200, 0, 3, 1.0), (ia, 0, 200, 1, 3, 1.0),              # both orientations of the whole matrix, x3
            (ia, 0, 128, 1, 3, alpha),                                       # a scaled row window (the LSTM layer-0 form), transposed
            (ia, 128, 72, 0, 1, 1.0),                                        # a window that ends with the matrix, one plane, plain
            (ib, 0, 130, 0, 1, 1.0), (ib, 0, 130, 1, 1, 1.0),
            (ic, 64, 64, 1, 3, 1.0), (ic, 64, 64, 0, 3, 1.0)]
    got.wimg.add(keys)
    assert got.wimg.active and L.lib().yt8m_wimg_count() >= len(keys)
    hyper = dict(gscale=0.5, clip=1.0, beta1=0.9, beta2=0.999, eps=1e-8)
    saved, ref.wimg = ref.wimg, None                                          # the reference run: chunk pass only
    for step in range(3):
        ops.sqnorm_and_adam(ref, 0.01 * (step + 1), **hyper)
        ops.sqnorm_and_adam(got, 0.01 * (step + 1), **hyper)
    ref.wimg = saved
    torch.cuda.synchronize()
    assert torch.equal(ref.params, got.params) and torch.equal(ref.adam_m, got.adam_m) and torch.equal(ref.adam_v, got.adam_v)
    assert float((ref.params - make().params).abs().max()) > 1e-3             # ... and the steps did move the weights
    for t, row0, rows, trans, planes, scale in keys:
        v = got.trainable_variables()[t]
        want = _image_of(v.data[row0:row0 + rows], planes, trans, scale)
        have = got.wimg.keys[(t, row0, rows, trans, planes, scale)]
        assert have.numel() == want.numel() and torch.equal(have, want), (t, row0, rows, trans, planes, scale)
        C = v.data.shape[1]
        r = wimg.resident_image(v.data[row0:row0 + rows], rows, C, C, trans, planes, scale)
        assert r is not None and r[0].data_ptr() == have.data_ptr() and (r[1], r[2]) == ((C, rows) if trans else (rows, C))
    # a sub-range of the tensors (the data-parallel buckets, the early pass of the recurrent stack): only those move
    before = got.params.clone()
    ops.sqnorm_and_adam(got, 0.01, tensors=(ib, ib + 1), **hyper)
    torch.cuda.synchronize()
    vb = got.vars["b"]
    moved = (got.params != before).nonzero().flatten()
    assert moved.numel() > 0 and int(moved.min()) >= vb.offset and int(moved.max()) < vb.offset + vb.numel()
    assert torch.equal(got.wimg.keys[(ib, 0, 130, 1, 1, 1.0)], _image_of(vb.data, 1, 1, 1.0))
    # a torch-side write to the arena is seen at the next begin_step: every image is rebuilt from the weights as they are
    got.vars["a"].data.mul_(1.5)
    n = wimg.STATS["refreshes"]
    got.begin_step()
    assert wimg.STATS["refreshes"] == n + 1
    assert torch.equal(got.wimg.keys[(ia, 0, 128, 1, 3, alpha)], _image_of(got.vars["a"].data[:128], 3, 1, alpha))
    # release: the lookup table forgets the arena
    lo = got.params.data_ptr()
    got.wimg.close()
    assert L.lib().yt8m_wimg_lookup(ctypes.c_void_p(lo + 4 * got.vars["a"].offset), 200, 136, 136, 0, 3, 1.0) is None


def _run_moe(dev, enabled, steps, bf16, flags, inject_at=None):
    import yt8m_amd.ops as _ops
    wimg.ENABLED = enabled
    h2_logits, _ops.MOE_LOGITS_H2 = _ops.MOE_LOGITS_H2, False        # (round 6: the fp32 head's logits default to per-call h2 images; the
    try:                                                             #  resident six-product images are what this test is about)
        flags.reset()
        if bf16:
            flags.compute_dtype = "bfloat16"
        B, D, V = 1024, 1152, 4716
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
        gen = torch.Generator(device=dev).manual_seed(21)
        losses = []
        for i in range(steps):
            x = torch.rand((B, D), device=dev, generator=gen) * 4.0 - 2.0
            y = torch.rand((B, V), device=dev, generator=gen) < (3.4 / V)
            if inject_at == i:
                g.vars["gates/weights"].data.mul_(0.5)                        # a torch-side write between two steps
            losses.append(float(tg.step(x, y)["loss"]))
        torch.cuda.synchronize()
        active = g.wimg is not None and g.wimg.active
        nimg = len(g.wimg.keys) if g.wimg is not None else 0
        return losses, g.params.clone(), g.adam_m.clone(), g.adam_v.clone(), active, nimg
    finally:
        wimg.ENABLED = True
        _ops.MOE_LOGITS_H2 = h2_logits


@pytest.mark.parametrize("bf16", [False, True])
def test_moe_training_with_resident_images_is_bitwise_the_run_that_resplits(dev, flags, bf16):
    """MoeModel (W/all_video_models/moe_model.py:12-65) at B = 1024 (BASELINE configs[1]): the products are large enough for the image kernels, so from
    the third step on the gate / expert weights are read from resident images (the forward's transposed ones; the input needs no
    dx) that the optimiser pass rewrote.  Same losses, same weights, same Adam slots, bit for bit -- also across a torch-side
    write to a weight between two steps."""
    t0 = wimg.STATS["tile_launches"]
    on = _run_moe(dev, True, 6, bf16, flags, inject_at=4)
    assert on[4] and on[5] >= 2 and wimg.STATS["tile_launches"] >= t0 + 4
    off = _run_moe(dev, False, 6, bf16, flags, inject_at=4)
    assert not off[4] and off[5] == 0
    assert on[0] == off[0], (on[0], off[0])
    for a, b in zip(on[1:4], off[1:4]):
        assert torch.equal(a, b)


def test_lstm_model_with_resident_images_is_bitwise_the_run_that_resplits(dev, flags, monkeypatch):
    """LstmModel on raw uint8 frames through the native recurrent stack (csrc/lstm_stack.hip): the input rows of every layer's
    weight [Din + H, 4H] own the transposed image of the forward projection (layer 0: times 4/255) and, above layer 0, the plain
    image of dx = dz . W_x^T; the MoE head owns both orientations of its weights.  Six steps with and without, bit for bit."""
    import yt8m_amd.seq_ops as seq_ops
    monkeypatch.setattr(wimg, "MIN_ELEMS", 0)

    def run(enabled):
        wimg.ENABLED = enabled
        try:
            flags.reset()
            flags.lstm_cells, flags.lstm_layers = "256", 2
            B, F, D, V = 32, 32, 64, 330
            g = reset_default_graph(device=dev, seed=0)
            tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
            gen = torch.Generator(device=dev).manual_seed(8)
            calls0 = dict(seq_ops.NATIVE_CALLS)
            losses = []
            for i in range(6):
                q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
                nf = torch.randint(1, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)
                y = torch.rand((B, V), device=dev, generator=gen) < 0.02
                losses.append(float(tg.step(q, y, nf)["loss"]))
            torch.cuda.synchronize()
            seq_ops.check_persist_errors()
            assert seq_ops.NATIVE_CALLS["fwd"] > calls0["fwd"], "the native stack did not run"
            keys = sorted(g.wimg.keys) if (g.wimg is not None and g.wimg.active) else []
            names = {v.index: v.name for v in g.trainable_variables()}
            return losses, g.params.clone(), g.adam_m.clone(), [(names[k[0]],) + k[1:] for k in keys]
        finally:
            wimg.ENABLED = True

    on, off = run(True), run(False)
    assert off[3] == [] and on[0] == off[0], (on[0], off[0])
    assert torch.equal(on[1], off[1]) and torch.equal(on[2], off[2])
    w0, w1 = ("RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l for l in range(2))
    # (the projections read h2 images under device-measured scales since the round-5 f16 products: the resident x3 image left is
    # the plain one of dx = dz . W_x^T above layer 0 -- while that product stays on the six-product form)
    assert on[3] == [] or all(k[0] == w1 and k[3] == 0 for k in on[3]), on[3]


# ---- single-pass NetVLAD forward (csrc/netvlad_fused.hip vlad_video_kernel) ---------------------------------------------
@pytest.mark.parametrize("shape", [(3, 300, 1152), (5, 50, 128), (2, 37, 1024), (4, 320, 1152), (6, 1, 256), (3, 33, 384)])
@pytest.mark.parametrize("nsplit", [2, 1])
def test_netvlad_single_pass_equals_the_rows_cols_pair(dev, shape, nsplit):
    """One workgroup per video (assignment GEMM, softmax, aggregation GEMM in ONE launch, the frames read from HBM once) against the
    two-kernel pair on the same inputs: cT is the same arithmetic bit for bit (same fragments, same summation order over the
    64-feature blocks, same epilogue), n and agg differ only by summation order / the per-video instead of per-batch f16 scale of c
    -- fp32-grade either way.  Shapes: the headline one, a partial last 32-frame step, D = 1024 (8 groups per wave), F = 320 (every
    LDS step in use), a single frame, D = 384.  (The oracle comparison of whichever form is the default: test_gpu_kernels.py::
    test_netvlad_fused_u8, which runs the single-pass kernel wherever it covers the shape.)"""
    import yt8m_amd.seq_ops as seq_ops
    lib = L.lib()
    B, F, Dm = shape
    K = 64
    rs = np.random.RandomState(F * 7 + Dm)
    q = rs.randint(0, 256, size=(B, F, Dm)).astype(np.uint8)
    q[0, 0] = 0
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0] = F
    if B > 2:
        nf[1], nf[2] = 1, 0
    Wc = torch.from_numpy((rs.randn(Dm, K) * 3.0 / np.sqrt(Dm)).astype(np.float32)).to(dev)
    bc = torch.from_numpy((rs.randn(K) * 0.5).astype(np.float32)).to(dev)
    qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)
    try:
        L.check(lib.yt8m_netvlad_set_single(1))
        assert lib.yt8m_netvlad_single_pass(B, F, Dm, K) == 1
        c1, n1, a1 = seq_ops.netvlad_fwd_u8(qd, nfd, Wc, bc, nsplit=nsplit)
        c1b, n1b, a1b = seq_ops.netvlad_fwd_u8(qd, nfd, Wc, bc, nsplit=nsplit)
        L.check(lib.yt8m_netvlad_set_single(0))
        assert lib.yt8m_netvlad_single_pass(B, F, Dm, K) == 0
        c0, n0, a0 = seq_ops.netvlad_fwd_u8(qd, nfd, Wc, bc, nsplit=nsplit)
    finally:
        L.check(lib.yt8m_netvlad_set_single(-1))
    torch.cuda.synchronize()
    assert torch.equal(c1, c1b) and torch.equal(n1, n1b) and torch.equal(a1, a1b)        # deterministic
    assert torch.equal(c1, c0)
    assert float((n1 - n0).abs().max()) <= 1e-5 * max(1.0, float(n0.abs().max()))
    tol = 2e-6 if nsplit == 2 else 3e-3
    assert float((a1 - a0).abs().max()) <= tol * max(1.0, float(a0.abs().max()))
    if B > 2:
        assert float(a1[2].abs().max()) == 0.0 and float(c1[2].abs().max()) == 0.0        # a video without frames


def test_head_weight_gradients_on_a_side_stream_are_the_same_step(dev, flags, monkeypatch):
    """LstmModel: when the native recurrent stack's backward follows, the MoE head leaves dW_g / dW_e / db_e to a side stream (the
    first backward recurrence waits for dx only) and the optimiser passes -- the early one inside yt8m_lstm_stack_bwd through
    yt8m_opt_ranges.after_stream, the end-of-step one through ops.join_side_work -- wait for it.  Same arithmetic, other stream:
    bitwise the step that computes them on the main stream; at the headline's head size the early pass engages in both runs."""
    import yt8m_amd.seq_ops as seq_ops

    def run(defer):
        monkeypatch.setattr(ops, "DEFER_HEAD_DW", defer)
        flags.reset()
        flags.lstm_cells, flags.lstm_layers = "256", 2
        B, F, D, V = 32, 32, 64, 4716                       # V = 4716: the head holds > 2^22 parameters, the early pass engages
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
        gen = torch.Generator(device=dev).manual_seed(3)
        early0 = seq_ops.EARLY_ADAM_RUNS[0]
        losses = []
        for i in range(4):
            q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
            nf = torch.randint(1, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)
            y = torch.rand((B, V), device=dev, generator=gen) < 0.001
            losses.append(float(tg.step(q, y, nf)["loss"]))
            assert g.side_pending == [] and not g.defer_head_dw
        torch.cuda.synchronize()
        seq_ops.check_persist_errors()
        return losses, g.params.clone(), g.adam_v.clone(), seq_ops.EARLY_ADAM_RUNS[0] - early0

    a, b = run(True), run(False)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[3] == 4 and b[3] == 4, (a[3], b[3])
