"""-m gpu, round 4: the backward recurrence that writes the operand images of its own dz (csrc/lstm_persist.hip, rotated epilogue)
against the separate split pass -- bit for bit -- and the in-kernel column sums; the split pass's per-tile column sums."""
import ctypes

import numpy as np
import pytest
import torch

import yt8m_amd._lib as L
import yt8m_amd.ops as ops
from yt8m_amd.ops import _p, _stream

pytestmark = pytest.mark.gpu


def _x3_bytes(lib, rows, K):
    return lib.yt8m_x3_image_bytes(rows, K)


@pytest.mark.parametrize("scaled", [False, True])
def test_recurrence_written_images_equal_the_split_pass_bit_for_bit(dev, scaled):
    lib = L.lib()
    B, F, H, t0, T = 128, 9, 1024, 2, 6
    rows = lib.yt8m_lstm_persist_bwd_images_rows(B, H)
    if rows <= 0:
        pytest.skip("this device / environment does not take the rotated epilogue at B = 128, H = 1024")
    gen = torch.Generator(device=dev).manual_seed(11)
    gates = torch.rand((F, B, 4 * H), device=dev, generator=gen)
    Wh = (torch.rand((H, 4 * H), device=dev, generator=gen) - 0.5) * 0.06
    cs = torch.randn((F + 1, B, H), device=dev, generator=gen) * 0.5
    dout = torch.randn((F, B, H), device=dev, generator=gen) * 0.01
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)
    nf[0], nf[1] = F, 0
    rsc = torch.rand((F * B,), device=dev, generator=gen) + 0.5
    work0 = torch.randn((4, B, H), device=dev, generator=gen) * 0.01
    M = T * B

    def run(images):
        dz = torch.zeros((F, B, 4 * H), device=dev)
        work = work0.clone()
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, T), dtype=torch.uint8, device=dev)
        if images is None:
            L.check(lib.yt8m_lstm_persist_bwd(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, _p(nf), t0, T, B, H,
                                              _p(pws), pws.numel(), _stream()))
        else:
            L.check(lib.yt8m_lstm_persist_bwd_images(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, _p(nf), t0, T, B, H,
                                                     _p(pws), pws.numel(), ctypes.byref(images), _stream()))
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        return dz, work

    plain = torch.zeros(_x3_bytes(lib, M, 4 * H), dtype=torch.uint8, device=dev)
    trans = torch.zeros(_x3_bytes(lib, 4 * H, M), dtype=torch.uint8, device=dev)
    trans_s = torch.zeros_like(trans)
    cp = torch.zeros((rows, 4 * H), device=dev)
    cps = torch.zeros((rows, 4 * H), device=dev)
    im = L.PersistBwdImages(plain.data_ptr(), trans.data_ptr(), trans_s.data_ptr() if scaled else None, rsc.data_ptr() if scaled else None,
                            cp.data_ptr(), cps.data_ptr() if scaled else None)
    dz_i, work_i = run(im)
    dz_r, work_r = run(None)
    # the images change nothing else: dz, and the (dh, dc) the launch hands to the next one -- the half of `work` its last step writes
    # (phase 0 + T steps).  (Round 6: the plain launch carries the running state in registers and no longer writes the OTHER half on the
    # way; the image-writing epilogue, short of registers, still does.)
    h = 2 * (T % 2)
    assert torch.equal(dz_i, dz_r) and torch.equal(work_i[h:h + 2], work_r[h:h + 2])
    part = dz_r[t0:t0 + T].reshape(M, 4 * H)
    assert float(part.abs().max()) > 0
    ref_p = torch.zeros_like(plain)
    ref_t = torch.zeros_like(trans)
    ref_ts = torch.zeros_like(trans)
    L.check(lib.yt8m_x3_split_ex(_p(part), M, 4 * H, 4 * H, 1.0, _p(rsc[t0 * B:]) if scaled else None, _p(ref_p), _p(ref_t),
                                 _p(ref_ts) if scaled else None, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(plain, ref_p), "plain image differs from yt8m_x3_split"
    assert torch.equal(trans, ref_t), "transposed image differs from yt8m_x3_split"
    if scaled:
        assert torch.equal(trans_s, ref_ts), "scaled transposed image differs from yt8m_x3_split_ex"
    col = part.double().sum(0)
    assert float((cp.double().sum(0) - col).abs().max()) <= 1e-5 * float(part.abs().sum(0).max())
    if scaled:
        cols = (part.double() * rsc[t0 * B:(t0 + T) * B, None].double()).sum(0)
        assert float((cps.double().sum(0) - cols).abs().max()) <= 1e-5 * float(part.abs().sum(0).max()) * 1.5


def test_split_pass_column_sums(dev):
    lib = L.lib()
    R, C = 200, 192
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((R, C), device=dev, generator=gen)
    rsc = torch.rand((R,), device=dev, generator=gen) + 0.5
    tr = torch.zeros(lib.yt8m_x3_image_bytes(C, R), dtype=torch.uint8, device=dev)
    trs = torch.zeros_like(tr)
    nt = (R + 63) // 64
    cp, cps = torch.zeros((nt, C), device=dev), torch.zeros((nt, C), device=dev)
    L.check(lib.yt8m_x3_split_colsum(_p(x), R, C, C, 1.0, _p(rsc), None, _p(tr), _p(trs), _p(cp), _p(cps), _stream()))
    ref, refs = torch.zeros_like(tr), torch.zeros_like(tr)
    L.check(lib.yt8m_x3_split_ex(_p(x), R, C, C, 1.0, _p(rsc), None, _p(ref), _p(refs), _stream()))
    torch.cuda.synchronize()
    assert torch.equal(tr, ref) and torch.equal(trs, refs)
    for t in range(nt):
        blk = x[64 * t:64 * (t + 1)].double()
        assert float((cp[t].double() - blk.sum(0)).abs().max()) < 1e-5
        assert float((cps[t].double() - (blk * rsc[64 * t:64 * (t + 1), None].double()).sum(0)).abs().max()) < 1e-5


def test_early_head_optimizer_pass_is_bitwise_the_end_of_step_pass(dev, flags, monkeypatch):
    """LstmModel on the native stack: clip + Adam of the MoE head inside the stack's backward pass (on the library's weight-gradient
    stream, seq_ops._early_optimizer_hook) leaves exactly the parameters, Adam slots and norms of the plain end-of-step pass."""
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.seq_ops as seq_ops
    import yt8m_amd.train as train
    from yt8m_amd.variables import reset_default_graph
    flags.lstm_cells = "256"
    B, F, D, V = 32, 32, 64, 4716
    gen = torch.Generator(device=dev).manual_seed(3)
    q = torch.randint(0, 256, (B, F, D), device=dev, generator=gen, dtype=torch.uint8)
    y = torch.rand((B, V), device=dev, generator=gen) < 3.4 / V
    nf = torch.randint(1, F + 1, (B,), device=dev, generator=gen, dtype=torch.int32)

    def run(early):
        monkeypatch.setattr(seq_ops, "EARLY_ADAM", early)
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
        for _ in range(3):
            out = tg.step(q, y, nf)
        torch.cuda.synchronize()
        return g.params.clone(), g.adam_m.clone(), g.adam_v.clone(), g.norms.clone(), float(out["loss"])

    n0, b0 = seq_ops.EARLY_ADAM_RUNS[0], seq_ops.NATIVE_CALLS["bwd"]
    a = run(True)
    if seq_ops.NATIVE_CALLS["bwd"] == b0:
        pytest.skip("the native stack did not take this shape on this device")
    assert seq_ops.EARLY_ADAM_RUNS[0] == n0 + 3, "the early pass did not engage"
    n1 = seq_ops.EARLY_ADAM_RUNS[0]
    b = run(False)
    assert seq_ops.EARLY_ADAM_RUNS[0] == n1
    for u, v in zip(a[:4], b[:4]):
        assert torch.equal(u, v)
    assert a[4] == b[4]


# ---- reduce-scatter + all-gather == all-reduce on real RCCL, two ranks (needs >= 2 GPUs, skipped on the 1-GPU box) ------
def _rsag_worker(rank, world, port, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import __graft_entry__
        __graft_entry__.load_package()
        import torch as th
        import torch.distributed as dist
        import yt8m_amd.parallel as parallel
        th.cuda.set_device(rank)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                                device_id=th.device("cuda", rank))
        n = 1000003                                                    # n % world != 0: the tail takes its own all-reduce
        gen = th.Generator(device="cuda").manual_seed(100 + rank)
        src = th.randn(n, device="cuda", generator=gen)
        a, b = src.clone(), src.clone()
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        parallel._rsag_torch(b, None).wait()
        th.cuda.synchronize()
        ok = bool(th.equal(a, b))
        # the library's own transport (yt8m_comm_allreduce_rsag_f32) against its all-reduce
        store = dist.TCPStore("127.0.0.1", port + 1, world, rank == 0)

        def exchange(uid):
            if rank == 0:
                store.set("uid", uid)
                return uid
            return store.get("uid")

        comm = parallel.CabiComm(rank, world, exchange=exchange, device=rank)
        c, d = src.clone(), src.clone()
        comm.all_reduce(c, algo="allreduce").wait()
        comm.all_reduce(d, algo="rs_ag").wait()
        th.cuda.synchronize()
        ok = ok and bool(th.equal(c, d)) and bool(th.allclose(a, c, rtol=0, atol=0))
        comm.close()
        dist.destroy_process_group()
        q.put((rank, "ok" if ok else "mismatch"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


def test_rsag_equals_allreduce_bitwise_on_two_ranks(dev):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rsag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---- --compute_dtype=bfloat16 on the native recurrent stack: hoisted products on one-plane bf16 images, fp32-grade recurrence ------
@pytest.mark.parametrize("u8", [True, False])
def test_native_stack_in_bf16_operand_mode(dev, flags, u8):
    """LstmModel with --compute_dtype=bfloat16 at a shape the library's stack takes (F B >= 1024, H = 256): the input projections,
    dx and the weight gradients run on the b1 kernel over bf16 roundings of their operands, the recurrence on the persistent
    fp32-grade kernels.  Predictions against the value emulation of exactly that (torch_ref.lstm_stack(bf16_operands="input"));
    gradients against the fp32 oracle within bf16 operand noise."""
    import yt8m_amd.frame_level_models as flm
    import yt8m_amd.seq_ops as seq_ops
    import yt8m_amd.train as train
    from oracle import np_ref, torch_ref
    from yt8m_amd.feature_transform import IdenticalTransformer
    from yt8m_amd.variables import reset_default_graph
    flags.lstm_cells, flags.lstm_layers, flags.compute_dtype = "256", 2, "bfloat16"
    rs = np.random.RandomState(61)
    B, F, D, V = 32, 32, 64, 19
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 1
    y = rs.rand(B, V) < 0.15
    if u8:
        q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
        x64 = np_ref.dequant_l2norm_folded(q, nf)
        xin = torch.from_numpy(q).to(dev)
        tcls = None
    else:
        x64 = rs.randn(B, F, D) * (np.arange(F)[None, :, None] < nf[:, None, None])
        xin = torch.from_numpy(x64.astype(np.float32)).to(dev)
        x64 = x64.astype(np.float32).astype(np.float64)
        tcls = IdenticalTransformer
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g, transformer_class=tcls)
    yd, nfd = torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    tg.forward(xin, yd, nfd)
    g.finalize()
    # small weights: with O(1) gains a 32-step recurrence amplifies the 2^-9 operand rounding chaotically and no emulation tracks it
    P = {k: (rs.randn(*v.shape) * (0.06 if "RNN" in k else 0.05)).astype(np.float32) for k, v in g.vars.items()}
    for k, v in P.items():
        g.vars[k].data.copy_(torch.from_numpy(v).to(dev).view(g.vars[k].data.shape))
    n0 = seq_ops.NATIVE_CALLS["fwd"]
    res = tg.forward(xin, yd, nfd)
    if seq_ops.NATIVE_CALLS["fwd"] == n0:
        pytest.skip("the native stack did not take this shape on this device")
    loss = tg.loss(res, yd)
    loss.backward()
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    tp = {k: T(v).requires_grad_(True) for k, v in P.items()}
    layers = [(tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    with torch.no_grad():
        ste = torch_ref.lstm_model_state(T(x64), torch.from_numpy(nf.astype(np.int64)), layers, bf16_operands="input")
        pre = torch_ref.moe(ste, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    got = res["predictions"].detach().cpu().double().numpy()
    assert np.abs(got - pre.numpy()).max() < 3e-3
    st = torch_ref.lstm_model_state(T(x64), torch.from_numpy(nf.astype(np.int64)), layers)
    pr = torch_ref.moe(st, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    lr = torch_ref.cross_entropy(pr, T(y))
    lr.backward()
    assert np.abs(got - pr.detach().numpy()).max() < 3e-2
    for k, t in tp.items():
        ref = t.grad.numpy()
        gk = g.vars[k].grad.detach().cpu().double().numpy().reshape(ref.shape)
        assert np.abs(gk - ref).max() <= 6e-2 * max(np.abs(ref).max(), 1e-3), k


# ---- the interleaved image-GEMM kernels against the round-3 kernels --------------------------------------------------------------------
def _with_schedule(mode, fn):
    lib = L.lib()
    L.check(lib.yt8m_x3_set_schedule(mode))
    try:
        return fn()
    finally:
        L.check(lib.yt8m_x3_set_schedule(0))


@pytest.mark.parametrize("M,N,K", [(300, 520, 16), (256, 256, 48), (257, 255, 80), (512, 300, 96), (300, 260, 112), (256, 512, 128),
                                   (1000, 777, 1000), (300, 5000, 72), (256, 256, 8192 + 16), (2304, 1024, 8192), (4096, 4096, 1152)])
def test_interleaved_image_gemms_equal_the_round3_kernels_bit_for_bit(dev, M, N, K):
    """gemm_b1q_kernel / gemm_x3q_kernel<3> / gemm_x3q_kernel<1> run the same products in the same order per accumulator as the
    round-3 kernels: identical bits, for K-block counts around the request ring's depth (1, 3, 5, 6, 7, 8), odd counts, ragged
    M / N, tiles whose K range is split into parts (few tiles, long K) and a many-tile shape."""
    g = torch.Generator(device=dev).manual_seed(M * 131 + N * 17 + K)
    A = torch.randn((M, K), device=dev, generator=g)
    B = torch.randn((N, K), device=dev, generator=g)
    bias = torch.randn((N,), device=dev, generator=g)
    # one-plane (bf16) images
    ia, ib = ops.bf16_image(A), ops.bf16_image(B)
    new = _with_schedule(1, lambda: ops.gemm_b1_grouped([dict(A=ia, B=ib, bias=bias)])[0].clone())
    old = _with_schedule(2, lambda: ops.gemm_b1_grouped([dict(A=ia, B=ib, bias=bias)])[0].clone())
    assert torch.equal(new, old)
    ref = A.bfloat16().double() @ B.bfloat16().double().t() + bias.double()
    assert float((new.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    # three-plane images (six products)
    xa, xb = ops.x3_split(A)[0], ops.x3_split(B)[0]
    new = _with_schedule(1, lambda: ops.gemm_x3_grouped([dict(A=xa, B=xb, bias=bias)])[0].clone())
    old = _with_schedule(2, lambda: ops.gemm_x3_grouped([dict(A=xa, B=xb, bias=bias)])[0].clone())
    assert torch.equal(new, old)
    ref = A.double() @ B.double().t() + bias.double()
    assert float((new.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))
    # one-plane x three-plane (the uint8 layer-0 form): A exact in bf16
    Aq = torch.randint(-128, 128, (M, K), device=dev, generator=g).float()
    i1 = ops.bf16_image(Aq)
    lib = L.lib()
    ws = ops._workspace(dev)
    r = torch.rand((M,), device=dev, generator=g) + 0.5
    cs = torch.randn((N,), device=dev, generator=g)

    affine = N % 4 == 0                                   # the affine epilogue needs N % 4 == 0; plain product + bias otherwise

    def x1x3():
        z = torch.full((M, N), float("nan"), device=dev)
        L.check(lib.yt8m_gemm_x1x3_nt(M, N, K, _p(i1.buf), _p(xb.buf), _p(z), N, _p(bias), _p(r) if affine else None,
                                      _p(cs) if affine else None, 0.25, _p(ws), ws.numel() * 4, _stream()))
        return z
    new, old = _with_schedule(1, x1x3), _with_schedule(2, x1x3)
    assert torch.equal(new, old)
    ref = Aq.double() @ B.double().t()
    ref = (r.double()[:, None] * (ref + 0.25 * cs.double()) if affine else ref) + bias.double()
    assert float((new.double() - ref).abs().max()) < 3e-6 * max(1.0, float(ref.abs().max()))


# ---- backward recurrence on one bf16 plane (--compute_dtype=bfloat16) -------------------------------------------------------------------
@pytest.mark.parametrize("B,F,H", [(128, 24, 1024), (256, 10, 512), (100, 9, 1024)])
def test_bf16_backward_recurrence_against_a_rounding_emulation(dev, B, F, H):
    """yt8m_lstm_persist_bwd_bf16 = the BasicLSTM backward recurrence with dh_{t-1} = bf16(dz_t) . bf16(W_h)^T (round to nearest
    even, fp32 accumulation), everything else as in the fp32 launch: against a torch restatement of exactly that (fp64 products of
    the rounded operands) to 2e-3 of max|dz| -- what is left is where a dz value sits within 1e-7 of a bf16 rounding boundary -- and
    within 5 % of the fp32 launch, from which it must differ (the bf16 form was taken) unless the shape cannot take it (the
    third case: B = 100 is not a multiple of four 16-row tiles per workgroup, so the request falls back to the fp32 form)."""
    lib = L.lib()
    if not lib.yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent backward recurrence not available for this shape / device")
    g = torch.Generator(device=dev).manual_seed(B + F + H)
    gates = torch.rand((F, B, 4 * H), device=dev, generator=g)
    gates[:, :, H:2 * H] = gates[:, :, H:2 * H] * 2.0 - 1.0           # j = tanh(.)
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.06
    cs = torch.randn((F + 1, B, H), device=dev, generator=g) * 0.5
    dout = torch.randn((F, B, H), device=dev, generator=g) * 0.01

    def run(fn):
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F), dtype=torch.uint8, device=dev)
        work = torch.zeros((4, B, H), device=dev)
        dz = torch.full((F, B, 4 * H), float("nan"), device=dev)
        L.check(fn(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dz), _p(work), 0, None, None, 0, F, B, H, _p(pws), pws.numel(), _stream()))
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        return dz
    dz32 = run(lib.yt8m_lstm_persist_bwd)
    dzb = run(lib.yt8m_lstm_persist_bwd_bf16)
    assert torch.isfinite(dzb).all()
    takes_bf16 = lib.yt8m_lstm_persist_bwd_images_rows(B, H) > 0 and H in (512, 1024)   # rotated, prefetching form
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float64)
    Wr = rnd(Wh) if takes_bf16 else Wh.double()
    dh = torch.zeros((B, H), dtype=torch.float64, device=dev)
    dc = torch.zeros((B, H), dtype=torch.float64, device=dev)
    ref = torch.empty((F, B, 4 * H), dtype=torch.float64, device=dev)
    for t in range(F - 1, -1, -1):
        gi, gj, gf, go = gates[t].double().chunk(4, 1)
        cp, cn = cs[t].double(), cs[t + 1].double()
        tc = torch.tanh(cn)
        dht = dh + dout[t].double()
        dct = dc + dht * go * (1.0 - tc * tc)
        ref[t] = torch.cat([dct * gj * gi * (1.0 - gi), dct * gi * (1.0 - gj * gj), dct * cp * gf * (1.0 - gf), dht * tc * go * (1.0 - go)], 1)
        dc = dct * gf
        dzt = ref[t].float()
        dh = (rnd(dzt) if takes_bf16 else dzt.double()) @ Wr.t()
    scale = float(ref.abs().max())
    assert float((dzb.double() - ref).abs().max()) < (2e-3 if takes_bf16 else 2e-5) * scale
    assert float((dzb - dz32).abs().max()) < 5e-2 * scale
    assert torch.equal(dzb, dz32) != takes_bf16


@pytest.mark.parametrize("B,F,H", [(128, 12, 1024), (64, 9, 512), (8, 5, 1024)])
def test_bf16_forward_recurrence_against_a_rounding_emulation(dev, B, F, H):
    """yt8m_lstm_persist_fwd_bf16 = BasicLSTMCell under dynamic_rnn (W/all_frame_models/lstm_model.py:34-47) with the recurrent
    product bf16(h_{t-1}) . bf16(W_h), fp32 accumulation, everything else as the fp32 launch: states and gates against a torch
    restatement of exactly that (fp64 products of the rounded operands) to 2e-3, within 2e-2 of the fp32 launch, and different from it
    where the shape takes the bf16-pipe kernel (the third case, one 16-row tile, does not: the request falls back)."""
    lib = L.lib()
    if not lib.yt8m_lstm_persist_supported(B, H):
        pytest.skip("persistent recurrence not available for this shape / device")
    g = torch.Generator(device=dev).manual_seed(B * 3 + F + H)
    z_in = torch.randn((F, B, 4 * H), device=dev, generator=g) * 0.7
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.08
    nf = torch.randint(0, F + 1, (B,), device=dev, generator=g, dtype=torch.int32)
    nf[0], nf[1] = F, 0
    h0 = torch.randn((B, H), device=dev, generator=g) * 0.3
    c0 = torch.randn((B, H), device=dev, generator=g) * 0.3

    def run(fn):
        pws = torch.zeros(lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F), dtype=torch.uint8, device=dev)
        z = z_in.clone()
        cs = torch.zeros((F + 1, B, H), device=dev)
        hs = torch.zeros((F + 1, B, H), device=dev)
        cs[0], hs[0] = c0, h0
        out = torch.full((F, B, H), float("nan"), device=dev)
        L.check(fn(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nf), 0, F, B, H, 1.0, _p(pws), pws.numel(), _stream()))
        torch.cuda.synchronize()
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
        return hs, cs, out
    hs32, cs32, out32 = run(lib.yt8m_lstm_persist_fwd)
    hsb, csb, outb = run(lib.yt8m_lstm_persist_fwd_bf16)
    takes_bf16 = H in (512, 1024) and B > 16
    rnd = (lambda t: t.to(torch.bfloat16).to(torch.float64)) if takes_bf16 else (lambda t: t.double())
    Wr = rnd(Wh)
    h, c = h0.double(), c0.double()
    for t in range(F):
        live = (t < nf).unsqueeze(1)
        i, j, f, o = (z_in[t].double() + rnd(h.float()) @ Wr).chunk(4, 1)
        cn = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
        hn = torch.tanh(cn) * torch.sigmoid(o)
        c, h = torch.where(live, cn, c), torch.where(live, hn, h)
        tol = 2e-3 if takes_bf16 else 2e-5
        assert float((hsb[t + 1].double() - h).abs().max()) < tol and float((csb[t + 1].double() - c).abs().max()) < tol
        assert float((outb[t].double() - torch.where(live, h, torch.zeros_like(h))).abs().max()) < tol
    assert float((hsb - hs32).abs().max()) < 2e-2
    assert torch.equal(hsb, hs32) != takes_bf16
