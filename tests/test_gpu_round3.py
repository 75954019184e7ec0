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


def test_check_persist_errors_sees_an_early_launch(dev, flags):
    """seq_ops.check_persist_errors() (called by checkpoint.save and bench.py) raises when an EARLIER launch of a step's many
    launches on the resident workspace timed out -- not just the last one -- and reads every workspace on its own stream."""
    flags.lstm_cells, flags.lstm_layers = "256", 2
    rs = np.random.RandomState(3)
    B, F, D, V = 32, 12, 64, 17
    q = torch.from_numpy(rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)).to(dev)
    y = torch.from_numpy(rs.rand(B, V) < 0.15).to(dev)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    tg.step(q, y)
    seq_ops.check_persist_errors()
    assert seq_ops._PERSIST_WS, "the persistent recurrence did not engage"
    ws, _ = next(iter(seq_ops._PERSIST_WS.values()))
    L.check(L.lib().yt8m_lstm_persist_debug_fault(_p(ws), _stream()))
    tg.step(q, y)                                                              # dozens of launches on the same workspace
    with pytest.raises(L.Yt8mHipError):
        seq_ops.check_persist_errors()
    seq_ops.check_persist_errors()


# ---- headline backward at headline length ------------------------------------------------------------------
def test_headline_recurrence_full_length_vs_fp64(dev):
    """H = 1024, two layers, F = 300, B = 32 (two 16-row tiles per row group), ragged lengths incl. 0 and F, the product's own time
    partition (one forward launch per layer, three backward parts -- NOT the test fixture's override): outputs, final states, dx
    and every weight / bias gradient against fp64 autograd of torch_ref.lstm_stack.  300 steps x 2 layers of dh feed-through:
    the tolerances are the F = 10 test's (test_gpu_round2.py::test_persistent_lstm_vs_fp64_oracle_full_width), i.e. no error
    growth with sequence length is tolerated beyond them."""
    from oracle import torch_ref
    from test_gpu_round2 import _stack_run
    B, F, D, H = 32, 300, 128, 1024
    if not L.lib().yt8m_lstm_persist_bwd_supported(B, H):
        pytest.skip("persistent recurrence not available on this device")
    assert seq_ops.PERSIST_FWD_CHUNKS == 1 and seq_ops.PERSIST_BWD_CHUNKS == 3, "this test must run the product's partition"
    rs = np.random.RandomState(12)
    nfh = rs.randint(1, F + 1, size=B).astype(np.int32)
    nfh[:6] = [F, 0, 1, F, 299, 150]
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
    B, F, D, V = 32, 24, 64, 17
    q = rs.randint(0, 256, size=(B, F, D)).astype(np.uint8)
    nf = np.resize(np.array([24, 0, 1, 4, 24, 7, 2, 23, 5, 13], dtype=np.int32), B)
    y = rs.rand(B, V) < 0.15
    qd, yd, nfd = torch.from_numpy(q).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(nf).to(dev)
    g = reset_default_graph(device=dev, seed=0)
    tg = train.TrainGraph(flm.LstmModel(), batch_size=B, graph=g)
    tg.forward(qd, yd, nfd)
    g.finalize()
    P = {k: (rs.randn(*v.shape) * 0.3).astype(np.float32) for k, v in g.vars.items()}
    for k, v in g.vars.items():
        v.data.copy_(torch.from_numpy(P[k]).to(dev).view(v.data.shape))
    res = tg.forward(qd, yd, nfd, fuse_loss=False)
    loss = tg.loss(res, yd)
    loss.backward()
    seq_ops.check_persist_errors()
    assert seq_ops._PERSIST_WS, "the persistent recurrence did not engage"
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
