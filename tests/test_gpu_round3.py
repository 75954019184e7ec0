"""-m gpu: round-3 parity and robustness cases (VERDICT r2 "Next round" items 1, 6, 7, 9 and ADVICE r2).

* the headline recurrence at headline LENGTH: H = 1024, two layers, F = 300, ragged, forward + backward, on the product's own
  time partition, against fp64 autograd of the oracle restatement (error growth of dh through 300 steps x 2 layers x 3 parts);
* model-level LstmModel on the persistent kernels under both time partitions (product default / caller's chunks);
* the sticky time-out word of the persistent recurrence;
* data-parallel readiness that one GPU can prove (1-rank RCCL group, CU-hogging neighbour on a side stream)."""
import ctypes

import numpy as np
import pytest
import torch

import yt8m_amd._lib as L
import yt8m_amd.frame_level_models as flm
import yt8m_amd.seq_ops as seq_ops
import yt8m_amd.train as train
from yt8m_amd.ops import _p, _stream
from yt8m_amd.variables import reset_default_graph

pytestmark = pytest.mark.gpu


def T64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


# ---- a time-out in ANY launch is reported, whatever ran on the workspace afterwards -------------------------
def test_persist_timeout_word_is_sticky_across_launches(dev):
    lib = L.lib()
    B, F, H = 32, 4, 256
    if not lib.yt8m_lstm_persist_supported(B, H):
        pytest.skip("persistent recurrence not available on this device")
    nbytes = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, F)
    pws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    z0 = torch.randn((F, B, 4 * H), device=dev, generator=g) * 0.3
    Wh = (torch.rand((H, 4 * H), device=dev, generator=g) - 0.5) * 0.1

    def fwd():
        z = z0.clone()
        cs = torch.zeros((F + 1, B, H), device=dev)
        hs = torch.zeros((F + 1, B, H), device=dev)
        out = torch.empty((F, B, H), device=dev)
        L.check(lib.yt8m_lstm_persist_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), None, 0, F, B, H, 1.0, _p(pws), nbytes, _stream()))
        return out

    ref = fwd()
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))                 # clean
    L.check(lib.yt8m_lstm_persist_debug_fault(_p(pws), _stream()))            # "launch k timed out"
    again = fwd()                                                              # launch k + 1 zeroes its control block ...
    again2 = fwd()
    with pytest.raises(L.Yt8mHipError):                                       # ... and the time-out is still reported
        L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))
    L.check(lib.yt8m_lstm_persist_status(_p(pws), _stream()))                 # reported once, then clear
    assert torch.equal(ref, again) and torch.equal(ref, again2)               # the per-launch flag did not leak into later launches


def _recurrence_workspaces():
    """(tensor whose head holds a sticky word, stream) of every resident recurrence workspace: the Python orchestration's per-layer
    buffers and the native stack's scratch (its first bytes are layer 0's workspace)."""
    return [(ws, st) for ws, st in seq_ops._PERSIST_WS.values()] + [(e[0], e[1]) for e in seq_ops._STACK_SCRATCH.values()]


@pytest.mark.parametrize("native", [True, False])
def test_check_persist_errors_sees_an_early_launch(dev, flags, monkeypatch, native):
    """seq_ops.check_persist_errors() (called by checkpoint.save and bench.py) raises when an EARLIER launch of a step's many
    launches on the resident workspace timed out -- not just the last one -- for the native stack and the Python orchestration."""
    monkeypatch.setattr(seq_ops, "NATIVE_STACK", native)
    seq_ops._PERSIST_WS.clear()
    seq_ops._STACK_SCRATCH.clear()
    flags.lstm_cells, flags.lstm_layers = "256", 2
    rs = np.random.RandomState(3)
    B, F, D, V = 32, 40, 64, 17
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.15).to(dev)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    tg.step(q, y)
    seq_ops.check_persist_errors()
    wss = _recurrence_workspaces()
    assert wss and bool(seq_ops._STACK_SCRATCH) == native, "the persistent recurrence did not engage as expected"
    L.check(L.lib().yt8m_lstm_persist_debug_fault(_p(wss[0][0]), _stream()))
    tg.step(q, y)                                                              # dozens of launches on the same workspace
    with pytest.raises(L.Yt8mHipError):
        seq_ops.check_persist_errors()
    seq_ops.check_persist_errors()


# ---- headline backward at headline length ------------------------------------------------------------------
@pytest.mark.parametrize("B,F", [(32, 300), (256, 24)])
def test_headline_recurrence_full_length_vs_fp64(dev, B, F):
    """(B = 256, F = 24: VERDICT r4 #8 -- the persistent-recurrence variants bench.py's per-GPU batch sweep runs, eight 32-row groups per
    workgroup column, against fp64 on their own instead of only against the per-step kernels.)
    H = 1024, two layers, F = 300, B = 32 (two 16-row tiles per row group), ragged lengths incl. 0 and F, the product's own time
    partition (one forward launch per layer, three backward parts -- NOT the test fixture's override): outputs, final states, dx
    and every weight / bias gradient against fp64 autograd of torch_ref.lstm_stack.  300 steps x 2 layers of dh feed-through:
    the tolerances are the F = 10 test's (test_gpu_round2.py::test_persistent_lstm_vs_fp64_oracle_full_width), i.e. no error
    growth with sequence length is tolerated beyond them."""
    from oracle import torch_ref
    from test_gpu_round2 import _stack_run
    D, H = 128, 1024
    if not L.lib().yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available on this device")
    assert seq_ops.PERSIST_FWD_CHUNKS == 1 and seq_ops.PERSIST_BWD_CHUNKS == 3, "this test must run the product's partition"
    rs = np.random.RandomState(12)
    nfh = rs.randint(1, F + 1, size=B).astype(np.int32)
    nfh[:6] = [F, 0, 1, F, F - 1, F // 2]
    nf = torch.from_numpy(nfh).to(dev)
    res, grads, x64, P = _stack_run(dev, B, F, D, H, 2, 4, nf, True)
    xs = x64.transpose(0, 1).clone().requires_grad_(True)          # oracle takes [B,F,D]
    layers = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in P]
    out, c, h = torch_ref.lstm_stack(xs, nf.cpu(), layers)
    ref = [out.transpose(0, 1)] + [t for pair in zip(c, h) for t in pair]
    for u, v in zip(res, ref):
        assert float((u.double() - v).abs().max()) < 2e-5
    gen2 = torch.Generator(device=dev).manual_seed(7)
    loss = sum((r * torch.rand(r.shape, device=dev, generator=gen2).cpu().double()).sum() for r in ref)
    loss.backward()
    dx_ref = xs.grad.transpose(0, 1)
    assert float(dx_ref.abs().max()) > 0
    assert float((grads[0].double() - dx_ref).abs().max()) <= 1e-4 * float(dx_ref.abs().max())
    gW = torch.cat([t.grad.reshape(-1) for pair in layers for t in pair])
    assert float((grads[1].double() - gW).abs().max()) <= 1e-4 * float(gW.abs().max())
    # early time steps carry gradient that travelled through the whole sequence: check them on their own scale too
    early = slice(0, 20)
    scale = float(dx_ref[early].abs().max())
    assert scale > 0 and float((grads[0][early].double() - dx_ref[early]).abs().max()) <= 2e-4 * scale


def test_lstm_model_on_persistent_kernels_vs_oracle(dev, flags, lstm_partition):
    """LstmModel (2 x 256 cells: persistent recurrence, x3 products, uint8 projection) on raw uint8 frames against the fp64 oracle
    for predictions and all gradients, under BOTH time partitions (the product's default and the caller's chunks = 4)."""
    from oracle import np_ref, torch_ref
    flags.lstm_cells, flags.lstm_layers, flags.lstm_pipeline_chunks = "256", 2, 4
    rs = np.random.RandomState(41)
    B, F, D, V = 32, 40, 64, 17
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = np.resize(np.array([40, 0, 1, 4, 40, 7, 2, 39, 15, 23], dtype=np.int32), B)
    y = rs.rand(B, V) < 0.15
    qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    calls0 = dict(seq_ops.NATIVE_CALLS)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    tg.forward(qd, yd, nfd)
    g.finalize()
    # (a contractive recurrence: with large weights 40 steps amplify fp32 rounding chaotically on ANY implementation)
    P = {k: (rs.randn(*v.shape) * 0.06).astype(np.float32) for k, v in g.vars.items()}
    for k, v in g.vars.items():
        v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))
    res = tg.forward(qd, yd, nfd, fuse_loss=False)
    loss = tg.loss(res, yd)
    loss.backward()
    seq_ops.check_persist_errors()
    assert _recurrence_workspaces(), "the persistent recurrence did not engage"
    if lstm_partition == "product-partition":
        assert calls0["fwd"] + 2 == seq_ops.NATIVE_CALLS["fwd"] and calls0["bwd"] + 1 == seq_ops.NATIVE_CALLS["bwd"], "native stack did not run"
    x64 = np_ref.dequant_l2norm_folded(q, nf)
    tp = {k: T64(v).requires_grad_(True) for k, v in P.items()}
    layers = [(tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/weights" % l], tp["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/biases" % l])
              for l in range(2)]
    state = torch_ref.lstm_model_state(T64(x64), torch.from_numpy(nf), layers)
    pr = torch_ref.moe(state, tp["gates/weights"], tp["experts/weights"], tp["experts/biases"], 2)
    assert float((res["predictions"].detach().cpu().double() - pr.detach()).abs().max()) < 1e-5
    torch_ref.cross_entropy(pr, T64(y)).backward()
    for k, t in tp.items():
        got = g.vars[k].grad.detach().cpu().double().view(t.shape)
        assert float((got - t.grad).abs().max()) <= 2e-4 * max(1.0, float(t.grad.abs().max())), k


# ---- the layer-0 weight gradient straight from the uint8 frames ---------------------------------------------
def _tm(a, B, F):
    """[B,F,...] batch-major -> time-major rows m = f * B + b."""
    return np.ascontiguousarray(np.swapaxes(a, 0, 1)).reshape((F * B,) + a.shape[2:])


def test_u8_frames_image_t_equals_oracle_bit_for_bit(dev):
    """(q - 128)^T as a one-plane operand image (rows = features, K = time-major frame rows; zeros for padding frames and the K
    tail) against oracle/x3_ref.image: q - 128 is exact in bf16, so plane 0 of the three-plane image of the fp32 matrix is the
    one-plane image and planes 1, 2 are zero."""
    from oracle import x3_ref
    lib = L.lib()
    rs = np.random.RandomState(5)
    for B, F, D in ((16, 5, 64), (10, 3, 48), (32, 2, 1152)):
        q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
        nf = rs.randint(0, F + 1, size=B).astype(np.int32)
        nf[0], nf[1] = F, 0
        live = (np.arange(F)[None, :] < nf[:, None])[:, :, None]
        xm = _tm(np.where(live, q.astype(np.float32) - 128.0, 0.0).astype(np.float32), B, F)          # [F*B, D]
        want = x3_ref.image(np.ascontiguousarray(xm.T))                                                # [RG, KB, 3, 32, 2, 8]
        assert not want[:, :, 1:].any()
        img = torch.full((want[:, :, 0].size * 2,), 0x55, dtype=torch.uint8, device=dev)
        qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)      # (named: a temporary would be freed before the launch)
        L.check(lib.yt8m_u8_frames_image_t(_p(qd), _p(nfd), B, F, D, _p(img), _stream()))
        got = img.cpu().numpy().view(np.uint16).reshape(want[:, :, 0].shape)
        rows = np.arange(want.shape[0] * 32).reshape(-1, 32) < D                                       # rows beyond D are never read
        assert np.array_equal(got[rows[:, None, :].repeat(want.shape[1], 1)], want[:, :, 0][rows[:, None, :].repeat(want.shape[1], 1)])


def test_x3_split_ex_row_scaled_image(dev):
    """yt8m_x3_split_ex: plain, transposed and row-scaled transposed images from one pass, each bit for bit against the oracle."""
    from oracle import x3_ref
    lib = L.lib()
    rs = np.random.RandomState(8)
    R, C = 150, 70
    x = (rs.randn(R, C) * np.exp(rs.randn(R, C) * 3)).astype(np.float32)
    r = (rs.rand(R) * 0.1).astype(np.float32)
    r[3] = 0.0
    nb = lambda rows, K: lib.yt8m_x3_image_bytes(rows, K)
    ip = torch.empty(nb(R, C), dtype=torch.uint8, device=dev)
    it = torch.empty(nb(C, R), dtype=torch.uint8, device=dev)
    its = torch.empty(nb(C, R), dtype=torch.uint8, device=dev)
    xd, rd = torch.from_numpy(x).to(dev), torch.from_numpy(r).to(dev)
    L.check(lib.yt8m_x3_split_ex(_p(xd), R, C, C, 1.0, _p(rd), _p(ip), _p(it), _p(its), _stream()))
    wp, wt = x3_ref.image(x), x3_ref.image(np.ascontiguousarray(x.T))
    wts = x3_ref.image(np.ascontiguousarray((x * r[:, None]).astype(np.float32).T))
    assert np.array_equal(ip.cpu().numpy().view(np.uint16).reshape(wp.shape), wp)
    assert np.array_equal(it.cpu().numpy().view(np.uint16).reshape(wt.shape), wt)
    assert np.array_equal(its.cpu().numpy().view(np.uint16).reshape(wts.shape), wts)
    assert lib.yt8m_x3_split_ex(_p(xd), R, C, C, 1.0, _p(rd), _p(ip), None, None, _stream()) == -1     # rowscale without its image


def test_colsum_weighted(dev):
    lib = L.lib()
    g = torch.Generator(device=dev).manual_seed(2)
    for rows, cols in ((12800, 4096), (100, 70), (5000, 33)):
        X = torch.randn((rows, cols), device=dev, generator=g)
        w = torch.rand((rows,), device=dev, generator=g)
        out = torch.full((cols,), 2.0, device=dev)
        outw = torch.full((cols,), 7.0, device=dev)
        ws = torch.empty(2 * lib.yt8m_colsum_workspace_bytes(rows, cols), dtype=torch.uint8, device=dev)
        L.check(lib.yt8m_colsum_weighted_f32(_p(X), rows, cols, cols, _p(w), _p(out), 1.0, _p(outw), _p(ws), ws.numel(), _stream()))
        ref, refw = X.double().sum(0), (X.double() * w.double()[:, None]).sum(0)
        assert float((out.double() - 2.0 - ref).abs().max()) <= 1e-5 * float(X.abs().sum(0).max())
        assert float((outw.double() - refw).abs().max()) <= 1e-5 * float(X.abs().sum(0).max())


def test_layer0_weight_gradient_from_uint8_frames(dev):
    """dW_x = x^T dz with x = l2_normalize(dequantise(q)) never materialised: alpha ((q - 128)^T (r (.) dz) + (beta / alpha)
    colsum(r (.) dz)) on the one-plane product, for a time part in the MIDDLE of whole-sequence images (K range + K-block strides),
    accumulating (beta = 1), against the fp64 product at D = 1152."""
    from oracle import np_ref
    lib = L.lib()
    rs = np.random.RandomState(17)
    B, F, D, N = 16, 12, 1152, 512
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(1, F + 1, size=B).astype(np.int32)
    nf[0], nf[1] = F, 0
    t0, T = 4, 5
    dz = (rs.randn(F * B, N) * 0.1).astype(np.float32)
    live_m = _tm((np.arange(F)[None, :] < nf[:, None])[:, :, None].astype(np.float32), B, F)[:, 0]
    dz *= live_m[:, None]                                                   # the recurrence writes zeros beyond num_frames
    x64 = _tm(np_ref.dequant_l2norm_folded(q, nf), B, F)                    # [F*B, D] fp64
    qd, nfd, dzd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev), torch.from_numpy(dz).to(dev)
    FB, M = F * B, T * B
    qimg = torch.empty(((FB + 31) // 32) * (D // 16) * 1024, dtype=torch.uint8, device=dev)
    rrow = torch.empty((FB,), device=dev)
    L.check(lib.yt8m_u8_frames_image(_p(qd), _p(nfd), B, F, D, 1e-12, _p(qimg), None, _p(rrow), _stream()))
    qT = torch.empty(((D + 31) // 32) * ((FB + 15) // 16) * 1024, dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_u8_frames_image_t(_p(qd), _p(nfd), B, F, D, _p(qT), _stream()))
    dzc = dzd[t0 * B:(t0 + T) * B]
    dzTs = torch.empty(lib.yt8m_x3_image_bytes(N, M), dtype=torch.uint8, device=dev)
    rr = rrow[t0 * B:]
    L.check(lib.yt8m_x3_split_ex(_p(dzc), M, N, N, 1.0, _p(rr), None, None, _p(dzTs), _stream()))
    db = torch.zeros((N,), device=dev)
    csr = torch.empty((N,), device=dev)
    L.check(lib.yt8m_colsum_weighted_f32(_p(dzc), M, N, N, _p(rr), _p(db), 0.0, _p(csr), None, 0, _stream()))
    dW = torch.full((D, N), 0.25, device=dev)
    ws = torch.empty(lib.yt8m_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
    alpha = 4.0 / 255.0
    beta = 128.0 * alpha + (4.0 / 512.0 - 2.0)
    qTk = ctypes.c_void_p(qT.data_ptr() + (t0 * B // 16) * 1024)
    L.check(lib.yt8m_gemm_x1x3_nt_ex(D, N, M, qTk, FB // 16, _p(dzTs), 0, _p(dW), N, None, alpha, None, _p(csr), beta / alpha, 1.0, _p(ws),
                                     ws.numel(), _stream()))
    ref = 0.25 + x64[t0 * B:(t0 + T) * B].T @ dz[t0 * B:(t0 + T) * B].astype(np.float64)
    err = np.abs(dW.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 2e-6 * np.abs(ref - 0.25).max() + 1e-7, err
    assert float((db.double() - torch.from_numpy(dz[t0 * B:(t0 + T) * B].astype(np.float64).sum(0)).to(dev)).abs().max()) < 1e-5


# ---- the whole stack behind the C ABI ----------------------------------------------------------------------------
@pytest.mark.parametrize("h2", [0, 1])
@pytest.mark.parametrize("B,F,D,H,L_,ragged", [(32, 40, 96, 256, 2, True), (128, 24, 1152, 1024, 2, True), (64, 32, 128, 512, 1, False)])
def test_native_stack_equals_the_python_orchestration(dev, monkeypatch, B, F, D, H, L_, ragged, h2):
    """yt8m_lstm_stack_fwd / _bwd (float input, dx requested) against the Python orchestration of the same per-call entry points on
    the same partition.  h2 = 0 (YT8M_STACK_H2=0: every hoisted product on the six-product bf16 split, as the orchestration):
    forward results are bit-identical (same kernels, same operands); gradients agree to fp32 rounding (the weight-gradient products
    read K ranges of whole-sequence images instead of per-part images: same values, same summation).  h2 = 1 (the default since
    round 5: every layer takes its projection and weight gradients as three f16 products -- a float bottom-layer input under a
    device-measured scale): the same function to the products' 2^-21 -- outputs to 5e-6, gradients to 2e-5 of their scale."""
    from test_gpu_round2 import _stack_run
    monkeypatch.setenv("YT8M_STACK_H2", str(h2))
    if not L.lib().yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available for this shape / device")
    nf = None
    if ragged:
        nf = torch.randint(0, F + 1, (B,), device=dev, generator=torch.Generator(device=dev).manual_seed(3), dtype=torch.int32)
        nf[0], nf[1] = F, 0
    n0 = dict(seq_ops.NATIVE_CALLS)
    a, ga, _, _ = _stack_run(dev, B, F, D, H, L_, 2, nf, True)
    assert seq_ops.NATIVE_CALLS["fwd"] == n0["fwd"] + 1 and seq_ops.NATIVE_CALLS["bwd"] == n0["bwd"] + 1, "native stack did not engage"
    monkeypatch.setattr(seq_ops, "NATIVE_STACK", False)
    b, gb, _, _ = _stack_run(dev, B, F, D, H, L_, 2, nf, True)
    assert seq_ops.NATIVE_CALLS["fwd"] == n0["fwd"] + 1
    for u, v in zip(a, b):
        assert torch.equal(u, v) if h2 == 0 else float((u - v).abs().max()) <= 5e-6
    tol = 2e-6 if h2 == 0 else 2e-5
    for u, v in zip(ga, gb):
        assert float((u - v).abs().max()) <= tol * float(v.abs().max()) + 1e-9


def test_full_lstm_model_step_through_the_c_abi_only(dev, flags):
    """One complete training step of LstmModel (uint8 frames -> recurrent stack -> MoE head + CrossEntropyLoss -> backward -> clip
    + Adam) driven through ctypes calls into libyt8m_hip.so ONLY: yt8m_lstm_stack_fwd, yt8m_moe_fwd, yt8m_moe_bwd,
    yt8m_lstm_stack_bwd (with the head's early optimiser pass handed over as a descriptor: yt8m_lstm_stack_set_early_optimizer) and
    yt8m_optimizer_ranges.  torch is the device allocator (plus one concatenation / split of the
    [c0|h0|c1|h1] state, pure data movement); the Graph object only lends its flat parameter / gradient / Adam arenas and chunk table.
    Result: the same updated parameters as TrainGraph.step of the Python host from the same initial weights."""
    import math
    lib = L.lib()
    flags.lstm_cells, flags.lstm_layers = "256", 2
    rs = np.random.RandomState(77)
    B, F, D, V, H, M, NL = 32, 40, 64, 33, 256, 2, 2
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    nfh = rs.randint(1, F + 1, size=B).astype(np.int32)
    nfh[:3] = [F, 0, 1]
    nf = torch.from_numpy(nfh).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.1).to(dev)

    def fresh():
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
        tg.forward(q, y, nf)
        tg.ensure_finalized()
        return g, tg

    g0, tg0 = fresh()
    P = {k: (rs.randn(*v.shape) * 0.2).astype(np.float32) for k, v in g0.vars.items()}

    def load(g):
        for k, v in g.vars.items():
            v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))

    load(g0)
    ref = tg0.step(q, y, nf)                                   # the Python host's step
    want = g0.params.detach().clone()
    ref_p, ref_loss = ref["predictions"].clone(), float(ref["loss"])

    def cabi_step(early):
        g, tg = fresh()                                        # arenas + chunk table only from here on
        load(g)
        g.begin_step()
        cell = lambda l, n: g.vars["RNN/multi_rnn_cell/cell_%d/basic_lstm_cell/%s" % (l, n)]
        Wg, We, be = g.vars["gates/weights"], g.vars["experts/weights"], g.vars["experts/biases"]
        desc = L.LstmStackDesc(B, F, D, H, NL, 1, 1.0, 0, 0, 0)
        assert lib.yt8m_lstm_stack_supported(ctypes.byref(desc)), lib.yt8m_last_error()
        tape = torch.empty(lib.yt8m_lstm_stack_tape_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=dev)
        scratch = torch.zeros(lib.yt8m_lstm_stack_scratch_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=dev)
        ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])
        Wp, bp = ptrs([cell(l, "weights").data for l in range(NL)]), ptrs([cell(l, "biases").data for l in range(NL)])
        L.check(lib.yt8m_lstm_stack_fwd(ctypes.byref(desc), _p(q), _p(nf), Wp, bp, _p(tape), tape.numel(), _p(scratch), scratch.numel(), _stream()))
        finals = []
        for l in range(NL):
            for which in (1, 2):                               # c_l, h_l
                finals.append(seq_ops._tape_view(lib, desc, tape, l, which, (B, H)))
        state = torch.cat(finals, dim=1).contiguous()          # [c0|h0|c1|h1]  (W/all_frame_models/lstm_model.py:52-57)
        S = state.shape[1]
        Zg = torch.empty((B, V * (M + 1)), device=dev)
        Ze = torch.empty((B, V * M), device=dev)
        p = torch.empty((B, V), device=dev)
        loss = torch.zeros((1,), device=dev)
        yu8 = y.to(torch.uint8).contiguous()
        nws = lib.yt8m_moe_workspace_bytes_ex(B, S, V, M)
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        L.check(lib.yt8m_moe_fwd(_p(state), _p(Wg.data), _p(We.data), _p(be.data), _p(yu8), 0, B, S, V, M, 1e-5, _p(Zg), _p(Ze), _p(p), _p(loss),
                                 _p(ws), nws, _stream()))
        assert float((p - ref_p).abs().max()) < 1e-6 and abs(float(loss) - ref_loss) <= 1e-5 * abs(ref_loss)
        dstate = torch.empty_like(state)
        L.check(lib.yt8m_moe_bwd(_p(state), _p(Wg.data), _p(We.data), _p(Zg), _p(Ze), _p(yu8), 0, B, S, V, M, 1e-5, 1.0, _p(Wg.grad), _p(We.grad),
                                 _p(be.grad), 0.0, _p(dstate), _p(ws), nws, _stream()))
        parts = [dstate[:, k * H:(k + 1) * H].contiguous() for k in range(2 * NL)]
        dcs, dhs = ptrs(parts[0::2]), ptrs(parts[1::2])
        dW, db = ptrs([cell(l, "weights").grad for l in range(NL)]), ptrs([cell(l, "biases").grad for l in range(NL)])
        zero = (ctypes.c_float * NL)(*([0.0] * NL))
        lr = 0.01                                              # --base_learning_rate, first step: no decay yet
        lr_t = lr * math.sqrt(1.0 - 0.999) / (1.0 - 0.9)
        assert abs(float(ref["learning_rate"]) - lr) < 1e-12
        nT = len(g.trainable_variables())
        # the optimiser pass as ONE descriptor (yt8m_opt_ranges): the head's variables -- final before the recurrent stack's backward
        # pass starts -- are updated by yt8m_lstm_stack_bwd itself in its idle window (yt8m_lstm_stack_set_early_optimizer), the rest
        # by yt8m_optimizer_ranges at the end; early = False: everything at the end.  Same kernels, same numbers: bitwise equal.
        tcs = (ctypes.c_int32 * len(g.chunk_start))(*g.chunk_start)

        def ranges(rs_):
            o = L.OptRanges()
            o.w, o.m, o.v, o.g = g.params.data_ptr(), g.adam_m.data_ptr(), g.adam_v.data_ptr(), g.grads.data_ptr()
            o.chunks, o.tensor_chunk_start = g.chunks.data_ptr(), g.chunk_start_dev.data_ptr()
            o.tensor_chunk_start_host = ctypes.cast(tcs, ctypes.c_void_p)
            o.l2, o.partial, o.norms = g.l2.data_ptr(), g.partial.data_ptr(), g.norms.data_ptr()
            o.nranges = len(rs_)
            for i, (lo, hi) in enumerate(rs_):
                o.range_lo[i], o.range_hi[i] = lo, hi
            o.gscale, o.clip, o.lr_t, o.beta1, o.beta2, o.eps = 1.0, float(tg.clip), lr_t, 0.9, 0.999, 1e-8
            return o

        head = sorted(v.index for v in (Wg, We, be))
        assert head == list(range(head[0], head[0] + 3))
        rest = [(lo, hi) for lo, hi in ((0, head[0]), (head[-1] + 1, nT)) if hi > lo]
        if early:
            L.check(lib.yt8m_lstm_stack_set_early_optimizer(ctypes.byref(ranges([(head[0], head[-1] + 1)]))))
        L.check(lib.yt8m_lstm_stack_bwd(ctypes.byref(desc), _p(q), _p(nf), Wp, _p(tape), tape.numel(), _p(scratch), scratch.numel(), None, dcs, dhs,
                                        dW, db, zero, zero, None, _stream()))
        L.check(lib.yt8m_lstm_stack_status(ctypes.byref(desc), _p(scratch), _stream()))
        L.check(lib.yt8m_optimizer_ranges(ctypes.byref(ranges(rest if early else [(0, nT)])), _stream()))
        torch.cuda.synchronize()
        return g.params.detach().clone(), g.adam_m.detach().clone()

    got_e, m_e = cabi_step(True)
    got_l, m_l = cabi_step(False)
    assert torch.equal(got_e, got_l) and torch.equal(m_e, m_l)  # the early pass is the same arithmetic, moved
    err = float((got_e - want).abs().max())
    assert err <= 1e-6 * max(1.0, float(want.abs().max())), err


# ---- data-parallel readiness one GPU can prove --------------------------------------------------------------------
@pytest.fixture()
def one_rank_rccl(dev):
    """A 1-rank RCCL (torch.distributed "nccl") group in this process: the whole reducer machinery runs, collectives are real RCCL
    kernels on this GPU."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def _lstm_steps(dev, reducer, q, y, nf, steps=2, hog=None):
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=q.shape[0], graph=g, reducer=reducer)
    outs = []
    for _ in range(steps):
        if hog is not None:
            hog()
        outs.append(tg.step(q, y, nf)["predictions"].clone())
    torch.cuda.synchronize()
    seq_ops.check_persist_errors()
    if reducer is not None:
        reducer.detach()
    return g.params.detach().clone(), outs


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_lstm_step_under_a_one_rank_reducer_is_bitwise_the_plain_step(dev, flags, one_rank_rccl, algo, monkeypatch):
    """The headline model (persistent kernels, native stack) under parallel.GradReducer on a 1-rank RCCL group -- bucketed async
    all-reduce (or reduce-scatter + all-gather) of every gradient bucket, per-bucket clip + Adam, the CU headroom policy active --
    ends two steps with bit-identical parameters and predictions to the plain single-GPU step."""
    import yt8m_amd.parallel as parallel
    flags.lstm_cells, flags.lstm_layers = "512", 2
    rs = np.random.RandomState(5)
    B, F, D, V = 64, 24, 64, 300
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.02).to(dev)
    nf = torch.from_numpy(rs.randint(1, F + 1, size=B).astype(np.int32)).to(dev)
    monkeypatch.setattr(seq_ops, "DP_LAYER_BUCKETS", algo == "allreduce")   # one transport with per-layer reports, one without
    n0, l0 = seq_ops.NATIVE_CALLS["bwd"], seq_ops.DP_LAYER_BUCKETS_USED[0]
    want, pw = _lstm_steps(dev, None, q, y, nf)
    assert seq_ops.DP_LAYER_BUCKETS_USED[0] == l0, "no reducer: the gradients are reported on the step's own stream"
    red = parallel.GradReducer(bucket_bytes=1 << 20, algo=algo)
    got, pg = _lstm_steps(dev, red, q, y, nf)
    assert seq_ops.NATIVE_CALLS["bwd"] == n0 + 4 and red.world == 1 and red.active
    # per-layer buckets: each layer's gradients were reported from the side stream that waits for that layer's completion point
    # (yt8m_lstm_stack_layer_done_wait), layer 1 before layer 0
    assert seq_ops.DP_LAYER_BUCKETS_USED[0] == l0 + (2 if algo == "allreduce" else 0)
    prev = ctypes.c_int(-1)
    L.check(L.lib().yt8m_lstm_persist_reserve_cus(0, ctypes.byref(prev)))
    assert prev.value == 0, "detach() must give the CU headroom back"
    assert torch.equal(want, got)
    for a, b in zip(pw, pg):
        assert torch.equal(a, b)


def test_persistent_recurrences_next_to_a_cu_hogging_kernel(dev, flags, one_rank_rccl):
    """A long-running kernel on a side stream holds 48 CUs (what the RCCL kernels of a large all-reduce do during the backward pass
    under data parallelism) while the step's persistent recurrences -- which need every workgroup resident -- are launched: with the
    reducer's CU headroom they neither time out nor change a bit, and the step stays within a bounded factor of the unloaded one."""
    import time
    import yt8m_amd.parallel as parallel
    flags.lstm_cells, flags.lstm_layers = "1024", 2
    rs = np.random.RandomState(6)
    B, F, D, V = 128, 24, 128, 300
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.02).to(dev)
    nf = torch.full((B,), F, dtype=torch.int32, device=dev)
    want, _ = _lstm_steps(dev, None, q, y, nf, steps=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lstm_steps(dev, None, q, y, nf, steps=3)
    base = time.perf_counter() - t0
    side = torch.cuda.Stream()
    sink = torch.zeros((4,), device=dev)

    def hog():                                                  # ~48 workgroups x ~10 ms of back-to-back MFMAs
        L.check(L.lib().yt8m_probe_mfma_f32(10000, 48, _p(sink), ctypes.c_void_p(side.cuda_stream)))

    red = parallel.GradReducer(bucket_bytes=1 << 20, reserve_cus=64)
    t0 = time.perf_counter()
    got, _ = _lstm_steps(dev, red, q, y, nf, steps=3, hog=hog)
    loaded = time.perf_counter() - t0
    assert torch.equal(want, got)
    assert loaded < 8 * base + 0.5, (loaded, base)


def test_x1x3_alpha_without_a_rank1_term(dev):
    """yt8m_gemm_x1x3_nt_ex with alpha != 1 and neither rowscale nor colsum (the per-part layer-0 weight-gradient product since the
    rank-1 remainder moved to one pass per step): C (+)= alpha * A1 . B3^T."""
    lib = L.lib()
    rs = np.random.RandomState(3)
    B, F, D, N = 16, 4, 64, 256
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    dz = rs.randn(F * B, N).astype(np.float32)
    qd, dzd = torch.from_numpy(q).to(dev), torch.from_numpy(dz).to(dev)
    FB = F * B
    qT = torch.empty(((D + 31) // 32) * ((FB + 15) // 16) * 1024, dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_u8_frames_image_t(_p(qd), None, B, F, D, _p(qT), _stream()))
    dzT = torch.empty(lib.yt8m_x3_image_bytes(N, FB), dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_x3_split(_p(dzd), FB, N, N, 1.0, None, _p(dzT), _stream()))
    C = torch.full((D, N), 1.5, device=dev)
    ws = torch.empty(lib.yt8m_gemm_workspace_bytes(), dtype=torch.uint8, device=dev)
    L.check(lib.yt8m_gemm_x1x3_nt_ex(D, N, FB, _p(qT), 0, _p(dzT), 0, _p(C), N, None, 0.25, None, None, 0.0, 1.0, _p(ws), ws.numel(), _stream()))
    xm = _tm(q.astype(np.float64) - 128.0, B, F)
    ref = 1.5 + 0.25 * xm.T @ dz.astype(np.float64)
    assert np.abs(C.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()


# ---- the bf16 configuration's products on one-plane operand images ------------------------------------------------
def test_b1_image_gemm_matches_row_major_bf16(dev):
    """yt8m_bf16_image + yt8m_gemm_b1_nt_grouped against the row-major bf16 kernel on the same bfloat16 values (identical products,
    fp32 accumulation in another order): ragged shapes, bias, accumulate, grouped launch; and the image equals plane 0 of the
    three-plane split bit for bit (round to nearest even)."""
    from oracle import x3_ref
    import yt8m_amd.ops as ops
    rs = np.random.RandomState(2)
    x = (rs.randn(70, 37) * np.exp(rs.randn(70, 37) * 3)).astype(np.float32)
    ip, it = ops.bf16_image(torch.from_numpy(x).to(dev), both=True)
    wp, wt = x3_ref.image(x)[:, :, 0], x3_ref.image(np.ascontiguousarray(x.T))[:, :, 0]
    assert np.array_equal(ip.buf.cpu().numpy().view(np.uint16).reshape(wp.shape), wp)
    assert np.array_equal(it.buf.cpu().numpy().view(np.uint16).reshape(wt.shape), wt)
    g = torch.Generator(device=dev).manual_seed(4)
    items_i, items_r = [], []
    for (M, N, K) in ((1000, 777, 1000), (300, 5000, 72), (2048, 4800, 1024)):
        A, Bm = torch.randn((M, K), device=dev, generator=g), torch.randn((N, K), device=dev, generator=g)
        bias = torch.randn((N,), device=dev, generator=g)
        c0 = torch.randn((M, N), device=dev, generator=g)
        items_i.append(dict(A=ops.bf16_image(A), B=ops.bf16_image(Bm), bias=bias, out=c0.clone(), beta=1.0))
        Ab, Bb = ops._bf16_empty(M, K, dev), ops._bf16_empty(N, K, dev)
        Ab.copy_(A)
        Bb.copy_(Bm)
        items_r.append(dict(A=Ab, B=Bb, bias=bias, out=c0.clone(), beta=1.0))
    got = ops.gemm_b1_grouped(items_i)
    want = ops.gemm_bf16_nt_grouped(items_r)
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())


def test_moe_head_bf16_on_images_matches_row_major(dev, flags, monkeypatch):
    """MoeModel with --compute_dtype=bfloat16 at a size where the head's three products take the image kernel (the fused mixing
    backward writes dL/dZ straight into operand images): predictions, loss and all gradients against the row-major bf16 path --
    same bfloat16 operands, so they agree to fp32 accumulation order."""
    import yt8m_amd.ops as ops
    import yt8m_amd.video_level_models as vlm
    flags.compute_dtype = "bfloat16"
    rs = np.random.RandomState(9)
    B, D, V = 2048, 1024, 1600
    x = torch.from_numpy(rs.randn(B, D).astype(np.float32)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.01).to(dev)
    res = {}
    for images in (True, False):
        monkeypatch.setattr(ops, "B1_IMAGES", images)
        calls = {"b1": 0}
        real = ops.gemm_b1_grouped

        def spy(items, _real=real, _c=calls):
            _c["b1"] += len(items)
            return _real(items)

        monkeypatch.setattr(ops, "gemm_b1_grouped", spy)
        g = reset_default_graph(device=dev, seed=0)
        tg = train.TrainGraph(vlm.MoeModel(), batch_size=B, graph=g)
        out = tg.forward(x, y)
        g.finalize()
        out = tg.forward(x, y)
        loss = tg.loss(out, y)
        loss.backward()
        torch.cuda.synchronize()
        assert (calls["b1"] >= 4) == images, calls
        res[images] = (out["predictions"].detach().clone(), float(loss), g.grads.detach().clone())
        monkeypatch.setattr(ops, "gemm_b1_grouped", real)
    (pa, la, ga), (pb, lb, gb) = res[True], res[False]
    assert float((pa - pb).abs().max()) < 1e-5 and abs(la - lb) <= 1e-5 * abs(lb)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max())


# ---- the composite's attention branch straight from the reader's uint8 frames ------------------------------
def _u8_case(dev, B, F, D, A, seed):
    rs = np.random.RandomState(seed)
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = rs.randint(0, F + 1, size=B).astype(np.int32)
    nf[0] = F
    if B > 1:
        nf[1] = 1
    a0, c0 = 4.0 / 255.0, 4.0 / 512.0 - 2.0
    xt = a0 * q.astype(np.float64) + c0                                  # W/utils.py:23-38
    live = np.arange(F)[None, :] < nf[:, None]
    r64 = live / np.maximum(np.sqrt((xt ** 2).sum(-1)), 1e-6)            # default_transformer.py:4-8: mask, l2-normalise
    x64 = xt * r64[..., None]
    return q, nf, r64, x64, rs


@pytest.mark.parametrize("B,F,D,A", [(3, 10, 64, 3), (5, 37, 1152, 8), (2, 300, 1152, 8), (4, 21, 260, 12)])
def test_u8_attention_primitives_vs_fp64(dev, B, F, D, A):
    """yt8m_u8_frame_scales / yt8m_skinny_fwd_u8 / yt8m_skinny_dw_u8 / yt8m_attn_pool_fwd_u8 / yt8m_attn_pool_dw_u8 against the
    float64 form on the dequantised, masked, l2-normalised frames (ragged num_frames incl. 0, 1 and F; D with a partial 256-wide
    lane block; A on both register widths)."""
    lib = L.lib()
    q, nf, r64, x64, rs = _u8_case(dev, B, F, D, A, seed=B * 1000 + D)
    qd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(nf).to(dev)
    r = seq_ops.u8_frame_scales(qd, nfd)
    assert np.abs(r.cpu().numpy() - r64).max() <= 2e-6 * r64.max()
    assert (r.cpu().numpy()[np.arange(F)[None, :] >= nf[:, None]] == 0).all()
    # logit FC: y = x . W + bias
    W = (rs.randn(D, A) * 0.3).astype(np.float32)
    bias = rs.randn(A).astype(np.float32)
    Wd, bd = torch.from_numpy(W).to(dev), torch.from_numpy(bias).to(dev)
    cs = Wd.sum(dim=0)
    y = torch.empty((B * F, A), device=dev)
    L.check(lib.yt8m_skinny_fwd_u8(_p(qd), D, _p(Wd), A, _p(bd), _p(r), _p(cs), _p(y), A, B * F, D, A, 0.0, _stream()))
    ref = x64.reshape(B * F, D) @ W.astype(np.float64) + bias
    assert np.abs(y.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    # weight gradient: dW = x^T dy (accumulating on top of an existing gradient)
    dy = rs.randn(B * F, A).astype(np.float32)
    dyd = torch.from_numpy(dy).to(dev)
    ws = torch.empty(max(1, lib.yt8m_skinny_workspace_bytes(B * F, D, A)), dtype=torch.uint8, device=dev)
    dW = torch.full((D, A), 0.5, device=dev)
    L.check(lib.yt8m_skinny_dw_u8(_p(qd), D, _p(dyd), A, _p(r), _p(dW), A, B * F, D, A, 1.0, _p(ws), ws.numel(), _stream()))
    refw = 0.5 + x64.reshape(B * F, D).T @ dy.astype(np.float64)
    assert np.abs(dW.cpu().numpy() - refw).max() < 2e-5 * max(1.0, np.abs(refw).max())
    # pooling and its weight gradient
    w = rs.rand(B, F, A).astype(np.float32)
    wd = torch.from_numpy(w).to(dev)
    C = seq_ops.pool_u8_raw(wd, qd, r)
    refC = np.einsum("bfa,bfd->bad", w.astype(np.float64), x64)
    assert np.abs(C.cpu().numpy() - refC).max() < 2e-5 * max(1.0, np.abs(refC).max())
    dC = rs.randn(B, A, D).astype(np.float32)
    dCd = torch.from_numpy(dC).to(dev)
    wg = wd.clone().requires_grad_(True)
    seq_ops.pool_tn_u8(wg, qd, r).backward(dCd)
    refdw = np.einsum("bfd,bad->bfa", x64, dC.astype(np.float64))
    assert np.abs(wg.grad.cpu().numpy() - refdw).max() < 2e-5 * max(1.0, np.abs(refdw).max())


def test_composite_attention_branch_never_materialises_float_frames(dev, flags, monkeypatch):
    """configs[4] composite on raw uint8 frames: no call of the [B,F,D] dequantise (yt8m_dequant_l2norm_u8) in a training step, and
    the step equals the one taken on the float frames (same weights) to fp32 rounding."""
    import yt8m_amd.losses as losses
    import yt8m_amd.ops as ops
    import yt8m_amd.feature_transform as ft
    rs = np.random.RandomState(41)
    B, F, D, V, A, Lc = 6, 24, 128, 40, 4, 2
    flags.netvlad_cluster_size, flags.netvlad_hidden_size, flags.lstm_attentions = 64, 32, A
    flags.deep_chain_layers, flags.deep_chain_relu_cells = Lc, 8
    flags.support_type = ",".join(["label"] * Lc)
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    nf = torch.from_numpy(np.array([24, 1, 7, 24, 13, 2], dtype=np.int32)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.1).to(dev)
    calls = {"dequant": 0}
    real = ops.dequant_l2norm

    def spy(*a, **kw):
        calls["dequant"] += 1
        return real(*a, **kw)

    outs = {}
    for mode in ("u8", "float"):
        g = reset_default_graph(device=dev, seed=3)
        tg = train.TrainGraph(flm.GatedNetVLADAttentionChainModel(), batch_size=B, graph=g, multitask=True,
                              label_loss_fn=losses.MultiTaskCrossEntropyLoss(),
                              transformer_class=ft.DefaultTransformer if mode == "u8" else ft.IdenticalTransformer)
        x = q if mode == "u8" else real(q, nf)
        monkeypatch.setattr(ops, "dequant_l2norm", spy)
        calls["dequant"] = 0
        res = tg.forward(x, y, nf)
        g.finalize()
        res = tg.forward(x, y, nf)
        loss = tg.loss(res, y)
        loss.backward()
        monkeypatch.setattr(ops, "dequant_l2norm", real)
        if mode == "u8":
            assert calls["dequant"] == 0, "the uint8 composite wrote a float copy of the frames"
        outs[mode] = (res["predictions"].detach().clone(), float(loss.detach()),
                      {k: v.grad.detach().clone() for k, v in g.vars.items() if v.trainable and v.grad is not None})
    pu, lu, gu = outs["u8"]
    pf, lf, gf = outs["float"]
    assert float((pu - pf).abs().max()) < 2e-5 and abs(lu - lf) < 2e-5 * abs(lf)
    for k in gf:
        scale = float(gf[k].abs().max())               # the logit bias gradient is zero up to rounding (softmax over the frames)
        assert float((gu[k] - gf[k]).abs().max()) < 5e-4 * scale + 1e-7, k


# ---- NetVLAD tail: descriptor-wide l2-normalisation as a per-video scale of the hidden layer (no pass over [B,K,D]) ------------
@pytest.mark.parametrize("B,F,K,D", [(5, 9, 8, 64), (3, 7, 64, 1152), (2, 5, 4, 260)])
def test_vlad_finish_q_scale_equals_l2_normalize_long_form(dev, B, F, K, D):
    """vlad, q = vlad_finish(..., want_q=True); (vlad . W) * rsqrt(max(sum_k q, eps)) against float64 autograd of
    l2_normalize(intra_normalise(agg - n c)) . W: value and the gradients w.r.t. agg, the assignments, the centres and W --
    including a video without frames (all rows clamped at zero) and a cluster whose residual norm is below sqrt(eps) (the one case
    where the gradient through q is not zero)."""
    from yt8m_amd.variables import xavier_uniform
    import yt8m_amd.ops as ops
    rs = np.random.RandomState(B * 100 + K)
    Hh = 16
    g = reset_default_graph(device=dev, seed=0)
    g.begin_step()
    cen = g.get_variable("c", (K, D), xavier_uniform)
    W = g.get_variable("w", (K * D, Hh), xavier_uniform)
    g.finalize()
    cnp = (rs.randn(K, D) * 0.3).astype(np.float32)
    Wnp = (rs.randn(K * D, Hh) / np.sqrt(K * D)).astype(np.float32)
    cen.data.copy_(torch.from_numpy(cnp).to(dev))
    W.data.copy_(torch.from_numpy(Wnp).to(dev).view(W.data.shape))
    a = rs.rand(B, F, K).astype(np.float32)
    a[0] = 0.0                                                        # a video without frames: n = 0, agg = 0 -> every row clamped
    agg = rs.randn(B, K, D).astype(np.float32)
    agg[0] = 0.0
    cnp[1] = 0.0                                                      # (exact residual: a centre at the origin)
    cen.data.copy_(torch.from_numpy(cnp).to(dev))
    agg[1, 1] = (rs.randn(D) * 1e-8).astype(np.float32)               # residual norm ~ 1e-8 * sqrt(D) < sqrt(eps) = 1e-6
    wout = rs.randn(B, Hh).astype(np.float32)
    g.begin_step()
    ad, aggd = torch.from_numpy(a).to(dev).requires_grad_(True), torch.from_numpy(agg).to(dev).requires_grad_(True)
    vlad, q = seq_ops.vlad_finish(aggd, ad, cen, want_q=True)
    scale = torch.rsqrt(torch.clamp(q.sum(dim=1), min=1e-12)).unsqueeze(1)
    h = ops.linear(vlad.reshape(B, K * D), W, None) * scale
    (h * torch.from_numpy(wout).to(dev)).sum().backward()
    # float64 long form
    a64, agg64 = T64(a).requires_grad_(True), T64(agg).requires_grad_(True)
    c64, W64 = T64(cnp).requires_grad_(True), T64(Wnp).requires_grad_(True)
    pre = agg64 - a64.sum(1).unsqueeze(2) * c64.unsqueeze(0)
    u = pre * torch.rsqrt(torch.clamp((pre * pre).sum(2, keepdim=True), min=1e-12))
    v = u.reshape(B, K * D)
    v = v * torch.rsqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12))
    h64 = v @ W64
    (h64 * T64(wout)).sum().backward()
    tol = lambda ref: 2e-5 * max(1e-6, float(ref.abs().max()))
    assert float((h.detach().cpu().double() - h64.detach()).abs().max()) < tol(h64.detach())
    assert float((aggd.grad.cpu().double() - agg64.grad).abs().max()) < tol(agg64.grad)
    assert float((ad.grad.cpu().double() - a64.grad).abs().max()) < tol(a64.grad)
    assert float((cen.grad.cpu().double().view(K, D) - c64.grad).abs().max()) < tol(c64.grad)
    assert float((W.grad.cpu().double().view(K * D, Hh) - W64.grad).abs().max()) < tol(W64.grad)


def test_lstm_stack_layer_done_wait_contract(dev, flags):
    """yt8m_lstm_stack_layer_done_wait: refuses a layer no backward call has recorded; after a backward call both layers can be
    waited for from another stream, and a stream that waited sees the layer's final gradient (same bits as after a device sync)."""
    lib = L.lib()
    flags.lstm_cells, flags.lstm_layers = "256", 2
    rs = np.random.RandomState(9)
    B, F, D, V = 32, 40, 64, 50
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.05).to(dev)
    nf = torch.from_numpy(rs.randint(1, F + 1, size=B).astype(np.int32)).to(dev)
    assert lib.yt8m_lstm_stack_layer_done_wait(7, None) != 0                  # never recorded (2 layers at most so far)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    n0 = seq_ops.NATIVE_CALLS["bwd"]
    res = tg.forward(q, y, nf)
    g.finalize()
    res = tg.forward(q, y, nf)
    tg.loss(res, y).backward()
    if seq_ops.NATIVE_CALLS["bwd"] == n0:
        pytest.skip("native stack not engaged for this shape")
    side = torch.cuda.Stream(device=dev)
    w1 = [v for k, v in g.vars.items() if "cell_1" in k and "kernel" in k or k.endswith("multi_rnn_cell/cell_1/basic_lstm_cell/kernel")]
    ws = [v for v in g.trainable_variables() if v.grad is not None and v.grad.numel() >= 256 * 4 * 256]
    copies = []
    for layer in (1, 0):
        L.check(lib.yt8m_lstm_stack_layer_done_wait(layer, ctypes.c_void_p(side.cuda_stream)))
        with torch.cuda.stream(side):
            copies.append([v.grad.clone() for v in ws])
    torch.cuda.synchronize()
    final = [v.grad.clone() for v in ws]
    # after waiting for layer 0 (the last one to finish) every LSTM gradient is final
    for a, b in zip(copies[1], final):
        assert torch.equal(a, b)
