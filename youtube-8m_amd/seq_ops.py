"""Frame-axis autograd ops over the C ABI: BasicLSTM layer (whole recurrence in one call), attention weights,
NetVLAD assignment / aggregation, batched pooling GEMMs.  See include/yt8m_hip.h for the kernel contracts."""
import ctypes

import torch

from . import _lib, ops
from .ops import _p, _stream, _dev, _f32c, _token


def _nf(num_frames):
    return None if num_frames is None else num_frames.to(torch.int32).contiguous()


class _LstmLayer(torch.autograd.Function):
    """One BasicLSTMCell layer under tf.nn.dynamic_rnn, time-major.

    x_tm [F,B,in]; W "weights" [in+H, 4H] (rows 0..in-1 act on x_t, the rest on h); b "biases" [4H].
    Returns (out_tm [F,B,H], c_final [B,H], h_final [B,H]).  Forward = one hoisted input GEMM over all steps +
    yt8m_lstm_layer_fwd; backward = yt8m_lstm_layer_bwd + four hoisted GEMMs / column sums."""

    @staticmethod
    def forward(ctx, x_tm, token, W, b, num_frames, forget_bias):
        x_tm = _f32c(x_tm)
        _dev(x_tm)
        F, B, Din = x_tm.shape
        H = W.data.shape[1] // 4
        assert W.data.shape[0] == Din + H, "cell weights must be [in + H, 4H]"
        z = torch.empty((F, B, 4 * H), dtype=torch.float32, device=x_tm.device)
        ops.gemm(x_tm.view(F * B, Din), W.data[:Din], out=z.view(F * B, 4 * H), bias=b.data)
        cs = torch.empty((F + 1, B, H), dtype=torch.float32, device=x_tm.device)
        hs = torch.empty((F + 1, B, H), dtype=torch.float32, device=x_tm.device)
        cs[0].zero_()
        hs[0].zero_()
        out = torch.empty((F, B, H), dtype=torch.float32, device=x_tm.device)
        nf = _nf(num_frames)
        Wh = W.data[Din:]
        ws = ops._workspace(x_tm.device)
        _lib.check(_lib.lib().yt8m_lstm_layer_fwd(_p(z), _p(Wh), 4 * H, _p(cs), _p(hs), _p(out), _p(nf), F, B, H,
                                                  float(forget_bias), _p(ws), ws.numel() * 4, _stream()))
        ctx.save_for_backward(x_tm)
        ctx.state = (z, cs, hs, nf, W, b)
        ctx.set_materialize_grads(False)
        return out, cs[F], hs[F]

    @staticmethod
    def backward(ctx, dout, dc_final, dh_final):
        (x_tm,) = ctx.saved_tensors
        gates, cs, hs, nf, W, b = ctx.state
        ctx.state = None
        F, B, Din = x_tm.shape
        H = W.data.shape[1] // 4
        dev = x_tm.device
        dz = torch.empty((F, B, 4 * H), dtype=torch.float32, device=dev)
        work = torch.empty((4, B, H), dtype=torch.float32, device=dev)
        dout = None if dout is None else _f32c(dout)
        dc_final = None if dc_final is None else _f32c(dc_final)
        dh_final = None if dh_final is None else _f32c(dh_final)
        Wh = W.data[Din:]
        ws = ops._workspace(dev)
        _lib.check(_lib.lib().yt8m_lstm_layer_bwd(_p(gates), _p(Wh), 4 * H, _p(cs), _p(dout), _p(dc_final), _p(dh_final),
                                                  _p(dz), _p(work), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _stream()))
        dz2 = dz.view(F * B, 4 * H)
        if W.grad is not None:
            beta = W.grad_beta()
            ops.gemm(x_tm.view(F * B, Din), dz2, out=W.grad[:Din], transA=True, beta=beta, role="dw")
            ops.gemm(hs[:F].view(F * B, H), dz2, out=W.grad[Din:], transA=True, beta=beta, role="dw")
            W.grad_done()
        if b.grad is not None:
            ops.colsum(dz2, b.grad.view(-1), beta=b.grad_beta())
            b.grad_done()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dz2, W.data[:Din], transB=True).view(F, B, Din)
        return dx, None, None, None, None, None


def lstm_layer(x_tm, W, b, num_frames, forget_bias=1.0):
    return _LstmLayer.apply(x_tm, _token(W._graph), W, b, num_frames, forget_bias)


class _GruLayer(torch.autograd.Function):
    """One tf.contrib.rnn.GRUCell layer under tf.nn.dynamic_rnn (W/all_frame_models/gru_pooling_model.py:34-47), time-major.
    x_tm [F,B,in]; Wg "gates/weights" [in+H, 2H] (r | u), bg "gates/biases" [2H]; Wc "candidate/weights" [in+H, H],
    bc "candidate/biases" [H].  Returns (out_tm [F,B,H], h_final [B,H]).  The input halves of both projections are hoisted
    GEMMs over all steps; yt8m_gru_layer_fwd / _bwd hold the time loop."""

    @staticmethod
    def forward(ctx, x_tm, token, Wg, bg, Wc, bc, num_frames):
        frames = x_tm if isinstance(x_tm, U8FrameImages) else None     # the reader's bytes (layer 0): see u8_hoisted_fwd
        if frames is None:
            x_tm = _f32c(x_tm)
            _dev(x_tm)
            F, B, Din = x_tm.shape
            dev = x_tm.device
        else:
            F, B, Din = frames.F, frames.B, frames.D
            dev = frames.q.device
        H = Wc.data.shape[1]
        assert Wg.data.shape == (Din + H, 2 * H) and Wc.data.shape[0] == Din + H, "GRU weights must be [in + H, 2H] / [in + H, H]"
        x2 = x_tm.view(F * B, Din) if frames is None else None
        zg = torch.empty((F, B, 2 * H), dtype=torch.float32, device=dev)
        zc = torch.empty((F, B, H), dtype=torch.float32, device=dev)
        bf = ops.FLAGS.compute_dtype == "bfloat16"
        # the hoisted input projections declare the h2 role like the LSTM stack's (l2-normalised frames / GRU outputs |h| <= 1 against one
        # weight matrix: three f16 products instead of six bf16 ones; ops._hoisted_role)
        if frames is not None:
            u8_hoisted_fwd(frames, Wg.data[:Din], bg.data, zg.view(F * B, 2 * H))
            u8_hoisted_fwd(frames, Wc.data[:Din], bc.data, zc.view(F * B, H))
        else:
            ops.gemm_any(x2, Wg.data[:Din], out=zg.view(F * B, 2 * H), bias=bg.data, bf16=bf, role=ops._hoisted_role(F * B, 2 * H, Din, bf))
            ops.gemm_any(x2, Wc.data[:Din], out=zc.view(F * B, H), bias=bc.data, bf16=bf, role=ops._hoisted_role(F * B, H, Din, bf))
        hs = torch.empty((F + 1, B, H), dtype=torch.float32, device=dev)
        hs[0].zero_()
        rh = torch.empty((F, B, H), dtype=torch.float32, device=dev)
        out = torch.empty((F, B, H), dtype=torch.float32, device=dev)
        nf = _nf(num_frames)
        L = _lib.lib()
        # one persistent launch per direction (csrc/gru_persist.inl) where the shape allows it, else the per-step kernels (2 + 3 launches
        # per time step); YT8M_GRU_PERSIST=0 forces the latter
        ctx.pws = None
        if F > 0 and L.yt8m_gru_persist_supported(B, H) and (GRU_PERSIST_FWD or GRU_PERSIST_BWD):
            main = torch.cuda.current_stream(dev)
            ctx.pws = _persist_ws(dev, main, "gru", L.yt8m_gru_persist_workspace_bytes(B, H, F))
        if ctx.pws is not None and GRU_PERSIST_FWD:
            _lib.check(L.yt8m_gru_persist_fwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(rh), _p(out),
                                              _p(nf), 0, F, B, H, _p(ctx.pws), ctx.pws.numel(), _stream()))
        else:
            ws = ops._workspace(dev)
            _lib.check(L.yt8m_gru_layer_fwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(rh),
                                            _p(out), _p(nf), F, B, H, _p(ws), ws.numel() * 4, _stream()))
        ctx.save_for_backward(x_tm if frames is None else frames.r)
        ctx.frames, ctx.dims = frames, (F, B, Din)
        ctx.state = (zg, zc, hs, rh, nf, Wg, bg, Wc, bc)
        ctx.bf16 = bf
        ctx.set_materialize_grads(False)
        return out, hs[F]

    @staticmethod
    def backward(ctx, dout, dh_final):
        (x_tm,) = ctx.saved_tensors
        frames = ctx.frames
        zg, zc, hs, rh, nf, Wg, bg, Wc, bc = ctx.state
        ctx.state = None
        F, B, Din = ctx.dims
        H = Wc.data.shape[1]
        dev = x_tm.device
        dzg = torch.empty((F, B, 2 * H), dtype=torch.float32, device=dev)
        dzc = torch.empty((F, B, H), dtype=torch.float32, device=dev)
        work = torch.empty((3, B, H), dtype=torch.float32, device=dev)
        dout = None if dout is None else _f32c(dout)
        dh_final = None if dh_final is None else _f32c(dh_final)
        if ctx.pws is not None and GRU_PERSIST_BWD:
            if dh_final is not None:
                work[0].copy_(dh_final)
            else:
                work[0].zero_()
            _lib.check(_lib.lib().yt8m_gru_persist_bwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(dout),
                                                       _p(dzg), _p(dzc), _p(work), _p(nf), 0, F, B, H, _p(ctx.pws), ctx.pws.numel(),
                                                       _stream()))
            ctx.pws = None
        else:
            ws = ops._workspace(dev)
            _lib.check(_lib.lib().yt8m_gru_layer_bwd(_p(zg), _p(zc), _p(Wg.data[Din:]), 2 * H, _p(Wc.data[Din:]), H, _p(hs), _p(dout),
                                                     _p(dh_final), _p(dzg), _p(dzc), _p(work), _p(nf), F, B, H, _p(ws),
                                                     ws.numel() * 4, _stream()))
        x2 = x_tm.view(F * B, Din) if frames is None else None
        g2, c2 = dzg.view(F * B, 2 * H), dzc.view(F * B, H)
        bf = ctx.bf16
        if Wg.grad is not None:
            beta = Wg.grad_beta()
            if frames is not None:
                u8_hoisted_dw(frames, g2, Wg.grad[:Din], beta)
            else:
                ops.gemm_any(x2, g2, out=Wg.grad[:Din], transA=True, beta=beta, role="dw", bf16=bf)
            ops.gemm_any(hs[:F].view(F * B, H), g2, out=Wg.grad[Din:], transA=True, beta=beta, role="dw", bf16=bf)
            Wg.grad_done()
        if Wc.grad is not None:
            beta = Wc.grad_beta()
            if frames is not None:
                u8_hoisted_dw(frames, c2, Wc.grad[:Din], beta)
            else:
                ops.gemm_any(x2, c2, out=Wc.grad[:Din], transA=True, beta=beta, role="dw", bf16=bf)
            ops.gemm_any(rh.view(F * B, H), c2, out=Wc.grad[Din:], transA=True, beta=beta, role="dw", bf16=bf)
            Wc.grad_done()
        if bg.grad is not None:
            ops.colsum(g2, bg.grad.view(-1), beta=bg.grad_beta())
            bg.grad_done()
        if bc.grad is not None:
            ops.colsum(c2, bc.grad.view(-1), beta=bc.grad_beta())
            bc.grad_done()
        dx = None
        if ctx.needs_input_grad[0] and frames is None:
            dx = ops.hoisted_dx(g2, Wg.data[:Din], bf16=bf)
            ops.hoisted_dx(c2, Wc.data[:Din], out=dx, beta=1.0, bf16=bf)
            dx = dx.view(F, B, Din)
        return dx, None, None, None, None, None, None


def gru_layer(x_tm, Wg, bg, Wc, bc, num_frames):
    return _GruLayer.apply(x_tm, _token(Wg._graph), Wg, bg, Wc, bc, num_frames)


class _LnLstmLayer(torch.autograd.Function):
    """One tf.contrib.rnn.LayerNormBasicLSTMCell layer under tf.nn.dynamic_rnn
    (W/all_frame_models/layernorm_lstm_memory_model.py:37-58), time-major.  W "weights" [in+H, 4H] (no bias); gammas / betas:
    five Variables each, in the order input, transform, forget, output, state.  keep_prob < 1 = the cell's dropout_keep_prob
    (tf.nn.dropout on the candidate g; mask = Philox stream of `seed`, replayed in backward).
    Returns (out_tm, c_final, h_final) -- c is the layer-normalised cell state, as in the TF cell."""

    @staticmethod
    def forward(ctx, x_tm, token, W, gammas, betas, num_frames, forget_bias, keep_prob, seed):
        frames = x_tm if isinstance(x_tm, U8FrameImages) else None     # the reader's bytes (layer 0): see u8_hoisted_fwd
        if frames is None:
            x_tm = _f32c(x_tm)
            _dev(x_tm)
            F, B, Din = x_tm.shape
            dev = x_tm.device
        else:
            F, B, Din = frames.F, frames.B, frames.D
            dev = frames.q.device
        H = W.data.shape[1] // 4
        assert W.data.shape[0] == Din + H, "cell weights must be [in + H, 4H]"
        bf = ops.FLAGS.compute_dtype == "bfloat16"
        if frames is not None:
            z = torch.empty((F, B, 4 * H), dtype=torch.float32, device=dev)
            u8_hoisted_fwd(frames, W.data[:Din], None, z.view(F * B, 4 * H))
        else:
            z = ops.gemm_any(x_tm.view(F * B, Din), W.data[:Din], bf16=bf, role=ops._hoisted_role(F * B, 4 * H, Din, bf)).view(F, B, 4 * H)
        gamma = torch.stack([v.data for v in gammas]).contiguous()
        beta = torch.stack([v.data for v in betas]).contiguous()
        stats = torch.zeros((F, B, 10), dtype=torch.float32, device=dev)
        cs = torch.empty((F + 1, B, H), dtype=torch.float32, device=dev)
        hs = torch.empty((F + 1, B, H), dtype=torch.float32, device=dev)
        cs[0].zero_()
        hs[0].zero_()
        out = torch.empty((F, B, H), dtype=torch.float32, device=dev)
        nf = _nf(num_frames)
        ws = ops._workspace(dev)
        _lib.check(_lib.lib().yt8m_lnlstm_layer_fwd(_p(z), _p(W.data[Din:]), 4 * H, _p(gamma), _p(beta), _p(stats), _p(cs), _p(hs),
                                                    _p(out), _p(nf), F, B, H, float(forget_bias), float(keep_prob), int(seed),
                                                    _p(ws), ws.numel() * 4, _stream()))
        ctx.save_for_backward(x_tm if frames is None else frames.r)
        ctx.frames, ctx.dims = frames, (F, B, Din)
        ctx.state = (z, gamma, beta, stats, cs, hs, nf, W, gammas, betas, float(forget_bias), float(keep_prob), int(seed))
        ctx.bf16 = bf
        ctx.set_materialize_grads(False)
        return out, cs[F], hs[F]

    @staticmethod
    def backward(ctx, dout, dc_final, dh_final):
        (x_tm,) = ctx.saved_tensors
        frames = ctx.frames
        z, gamma, beta, stats, cs, hs, nf, W, gammas, betas, fb, keep, seed = ctx.state
        ctx.state = None
        F, B, Din = ctx.dims
        H = W.data.shape[1] // 4
        dev = x_tm.device
        dz = torch.empty((F, B, 4 * H), dtype=torch.float32, device=dev)
        dyb = torch.empty((F, B, 5 * H), dtype=torch.float32, device=dev)
        dyg = torch.empty((F, B, 5 * H), dtype=torch.float32, device=dev)
        work = torch.empty((4, B, H), dtype=torch.float32, device=dev)
        dout = None if dout is None else _f32c(dout)
        dc_final = None if dc_final is None else _f32c(dc_final)
        dh_final = None if dh_final is None else _f32c(dh_final)
        ws = ops._workspace(dev)
        _lib.check(_lib.lib().yt8m_lnlstm_layer_bwd(_p(z), _p(W.data[Din:]), 4 * H, _p(gamma), _p(beta), _p(stats), _p(cs), _p(dout),
                                                    _p(dc_final), _p(dh_final), _p(dz), _p(dyb), _p(dyg), _p(work), _p(nf), F, B, H,
                                                    fb, keep, seed, _p(ws), ws.numel() * 4, _stream()))
        dz2 = dz.view(F * B, 4 * H)
        if W.grad is not None:
            b = W.grad_beta()
            if frames is not None:
                u8_hoisted_dw(frames, dz2, W.grad[:Din], b)
            else:
                ops.gemm_any(x_tm.view(F * B, Din), dz2, out=W.grad[:Din], transA=True, beta=b, role="dw", bf16=ctx.bf16)
            ops.gemm_any(hs[:F].view(F * B, H), dz2, out=W.grad[Din:], transA=True, beta=b, role="dw", bf16=ctx.bf16)
            W.grad_done()
        yb, yg = dyb.view(F * B, 5 * H), dyg.view(F * B, 5 * H)
        for k in range(5):
            for v, src in ((gammas[k], yg), (betas[k], yb)):
                if v.grad is not None:
                    ops.colsum(src[:, k * H:(k + 1) * H], v.grad.view(-1), beta=v.grad_beta())
                    v.grad_done()
        dx = None
        if ctx.needs_input_grad[0] and frames is None:
            dx = ops.hoisted_dx(dz2, W.data[:Din], bf16=ctx.bf16).view(F, B, Din)
        return dx, None, None, None, None, None, None, None, None


def lnlstm_layer(x_tm, W, gammas, betas, num_frames, forget_bias=1.0, keep_prob=1.0, seed=None):
    if seed is None:
        seed = W._graph.next_random_seed() if float(keep_prob) < 1.0 else 0
    return _LnLstmLayer.apply(x_tm, _token(W._graph), W, tuple(gammas), tuple(betas), num_frames, forget_bias, keep_prob, seed)


# ---- MultiRNNCell stack, pipelined over time chunks ----------------------------------------------------------------
_SIDE = {}


def _side_streams(device, L, persistent=False, separate_gemm=False):
    """Per device: one stream per layer (its projection / dx GEMMs and its recurrence run in order on it) and one for the
    weight-gradient GEMMs.  Measured on the 2-layer BASELINE configs[3] stack (B = 128): a separate GEMM stream per layer is
    no better (51.7 vs 49.0 ms/step).  The layer streams are HIGH priority when the recurrence is the persistent kernel: it needs
    every workgroup resident, and behind a high-priority queue it takes freed CUs before the remaining workgroups of a running
    weight-gradient GEMM do (a partially resident recurrence spins on its CUs while the GEMM crawls on the rest: 28.6 vs 46.6
    ms/step run to run without it).  With the per-step kernels (600 launches) priority queues are far worse (96 ms), so those
    keep normal streams.  separate_gemm (forward wavefront of half-chip recurrences): the projections get normal-priority streams
    of their own -- on the layer stream a projection's workgroups compete at high priority with the OTHER layer's recurrence for
    the CUs it needs (one run in three took 41 instead of 24.6 ms/step)."""
    pool = _SIDE.setdefault((device, bool(persistent)), dict(r=[], g=[], w=None))
    if persistent and len(pool["r"]) < L:
        # the library's own streams (csrc/lstm_stack.hip): every stream of a process is multiplexed onto a few hardware queues, and a
        # second set of layer streams next to the native stack's put both layers of this orchestration on ONE queue (bf16 variant of
        # the bench: 19.9 -> 46 ms/step)
        with torch.cuda.device(device):
            arr, w = (ctypes.c_void_p * L)(), ctypes.c_void_p()
            _lib.check(_lib.lib().yt8m_lstm_stack_streams(L, arr, ctypes.byref(w)))
        pool["r"] = [torch.cuda.ExternalStream(int(arr[l]), device=device) for l in range(L)]
        pool["w"] = torch.cuda.ExternalStream(int(w.value), device=device)
    while len(pool["r"]) < L:
        pool["r"].append(torch.cuda.Stream(device=device, priority=REC_STREAM_PRIORITY if persistent else 0))
    if pool["w"] is None:
        pool["w"] = torch.cuda.Stream(device=device)
    if separate_gemm:
        while len(pool["g"]) < L:
            pool["g"].append(torch.cuda.Stream(device=device))
        return pool["r"][:L], pool["g"][:L], pool["w"]
    return pool["r"][:L], pool["r"][:L], pool["w"]


def _chunks(F, n):
    n = max(1, min(int(n), F))
    step = (F + n - 1) // n
    return [(t0, min(step, F - t0)) for t0 in range(0, F, step)]


import os as _os
PERSIST = _os.environ.get("YT8M_LSTM_PERSIST", "1") != "0"        # persistent recurrence kernels (csrc/lstm_persist.hip)
U8_BETA = 128.0 * 4.0 / 255.0 + (4.0 / 512.0 - 2.0)     # dequantise(q) = (4/255) (q - 128) + U8_BETA
BWD_CHUNKS = int(_os.environ.get("YT8M_LSTM_BWD_CHUNKS", "0"))    # 0: same partition as the forward pass
# explicit backward partition as fractions of F in forward-time order, e.g. "1,2,3" -> chunks of F/6, F/3, F/2 (the backward pass
# runs them last to first: a long first launch, a short last one whose weight-gradient tail is short)
BWD_PARTS = [float(v) for v in _os.environ.get("YT8M_LSTM_BWD_PARTS", "").split(",") if v]


def _bwd_parts(F, fwd_parts, persistent=False):
    if BWD_PARTS:
        tot, edges, acc = sum(BWD_PARTS), [0], 0.0
        for v in BWD_PARTS:
            acc += v
            edges.append(min(F, int(round(F * acc / tot))))
        edges[-1] = F
        return [(a, b - a) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    if BWD_CHUNKS > 0:
        return _chunks(F, BWD_CHUNKS)
    if persistent and PERSIST_BWD_CHUNKS > 0:
        return _chunks(F, PERSIST_BWD_CHUNKS)
    return fwd_parts
# Partition of the time axis when the recurrence runs on the persistent kernels (0: the caller's `chunks`).  A whole-chip forward
# launch leaves no CU for the other layer's projection, so cutting the forward pass only adds launches: ONE chunk.  The backward
# launches take half the chip and run beside the weight-gradient GEMMs: three parts (shorter lone first / last phases and a
# shorter GEMM tail than two; BASELINE configs[3], B = 128: 24.3 ms/step against 24.8 with 2 + 2, 24.4 with 1 + 4).
PERSIST_FWD_CHUNKS = int(_os.environ.get("YT8M_LSTM_PERSIST_FWD_CHUNKS", "1"))
PERSIST_BWD_CHUNKS = int(_os.environ.get("YT8M_LSTM_PERSIST_BWD_CHUNKS", "3"))
PERSIST_DBROWS = False
# half-chip forward recurrences of two layers side by side (opt-in: 24.6 ms/step against 25.3 at B = 128, H = 1024 when the
# projections share the high-priority layer streams, but one run in three then took 41 ms; with projection streams of their own --
# what the code does -- it is stable at 27.9)
FWD_WAVEFRONT = _os.environ.get("YT8M_LSTM_FWD_WAVEFRONT", "0") != "0"
FWD_WAVEFRONT_SHARED = _os.environ.get("YT8M_LSTM_FWD_WAVEFRONT", "0") == "2"   # projections on the (high-priority) layer streams
FWD_WAVEFRONT_CHUNKS = int(_os.environ.get("YT8M_LSTM_FWD_WAVEFRONT_CHUNKS", "10"))
REC_STREAM_PRIORITY = int(_os.environ.get("YT8M_REC_STREAM_PRIORITY", "-1"))
PERSIST_STEP_IMAGES = _os.environ.get("YT8M_PERSIST_STEP_IMAGES", "1") != "0"
PERSIST_STEP_IMAGES_MAX_BYTES = 8 << 30                 # per layer; larger launches keep the two-image exchange
_PERSIST_WS = {}      # (device, caller's stream, layer, bytes) -> scratch workspace of that layer's persistent launches
_PERSIST_WS_MAX = 16


def _persist_ws(dev, main, l, nbytes):
    """Scratch of the persistent recurrence launches of layer `l` (control words + exchange images; nothing in it outlives a
    launch): ONE resident buffer per (device, calling stream, layer, size), reused by every step.  Every launch that touches it is
    ordered through the calling stream -- a stack's forward and backward both start from an event of that stream and join it
    before they return -- so two users of one key never overlap on the GPU.  Allocating it per step pinned ~0.6 GB per layer and
    step (32 steps' worth before the old table was cleared) and put two hipMalloc calls into every backward pass; on some boxes
    each of those waits ~17 ms on a DMA fence inside the driver, which made the step host-bound (25 -> 38-42 ms).
    The buffer's first line is the sticky error word of include/yt8m_hip.h: zeroed here, once; a buffer is only dropped from the
    table after its status has been read (a time-out is never lost with the eviction)."""
    key = (dev.index, main.cuda_stream, l, int(nbytes))
    ent = _PERSIST_WS.get(key)
    if ent is not None:
        _PERSIST_WS[key] = _PERSIST_WS.pop(key)                    # LRU order (dicts keep insertion order)
    else:
        if len(_PERSIST_WS) >= _PERSIST_WS_MAX:                    # other shapes: drop the least recently used entry
            _retire(_PERSIST_WS.pop(next(iter(_PERSIST_WS))), _check_ws)
            global _PERSIST_EVICTIONS
            _PERSIST_EVICTIONS += 1
            if _PERSIST_EVICTIONS in (64, 1024):                     # a model cycling through > 16 shapes re-zeroes ~0.6 GB per miss
                import warnings
                warnings.warn("yt8m_amd: %d persistent-recurrence workspaces evicted; raise seq_ops._PERSIST_WS_MAX" % _PERSIST_EVICTIONS)
        ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=dev)   # zeroed once: the sticky error word starts clear
        ent = (ws, main)
        _PERSIST_WS[key] = ent
    return ent[0]


_PERSIST_EVICTIONS = 0
_STACK_SCRATCH_MAX = int(_os.environ.get("YT8M_STACK_SCRATCH_MAX", "8"))
# Evicted workspaces that something may still launch on (an autograd ctx between its forward and its backward keeps the tensor
# alive): weak references, checked -- and dropped -- by the next check_persist_errors(), so a time-out in such a backward pass is
# still reported.  A tensor nobody holds any more cannot be launched on and needs no check beyond the one at eviction.
_RETIRED = []


def _retire(ent, checker):
    import weakref
    checker(*ent)                                                    # what ran so far (synchronises that stream)
    _RETIRED.append((weakref.ref(ent[0]), ent[1:], checker))
    del ent


def _check_ws(ws, stream):
    """Reads (and clears) the sticky error word of one workspace on ITS device and the stream its launches are ordered through."""
    with torch.cuda.device(ws.device):
        _lib.check(_lib.lib().yt8m_lstm_persist_status(_p(ws), ctypes.c_void_p(stream.cuda_stream)))


def check_persist_errors():
    """Synchronises and raises if ANY persistent-recurrence launch since the previous check gave up waiting for a tile (bounded
    spins set a sticky error word instead of hanging the GPU; no later launch clears it).  The training loop calls this at its
    checkpoints; tests run with PERSIST_CHECK per launch."""
    err = None
    for ws, stream in list(_PERSIST_WS.values()):
        try:
            _check_ws(ws, stream)
        except _lib.Yt8mHipError as e:                               # read (= clear) every word before raising
            err = e
    for ent in list(_STACK_SCRATCH.values()):
        try:
            _check_stack(*ent)
        except _lib.Yt8mHipError as e:
            err = e
    retired, _RETIRED[:] = list(_RETIRED), []
    for ref, rest, checker in retired:
        ws = ref()
        if ws is None:
            continue
        try:
            checker(ws, *rest)
        except _lib.Yt8mHipError as e:
            err = e
        _RETIRED.append((ref, rest, checker))                        # still alive: stays on the list
    if err is not None:
        raise err
PERSIST_BWD = _os.environ.get("YT8M_LSTM_PERSIST_BWD", "1") != "0"
# GRUCell on the persistent protocol (csrc/gru_persist.inl), per direction; the library switch YT8M_GRU_PERSIST=0 turns both off.
# Measured at B = 128, H = 1024, F = 300 (profiles/r6_gru_persist.txt): forward 13.9 us/step in one launch against 17.6 in 600 -> ON;
# backward 29.0 us/step against 20.3 for the three per-step launches (two exchange rounds per step on a K = 3H reduction: every
# item's fragment fetch is exposed with four tiles per workgroup) -> parity-tested, OPT-IN.
GRU_PERSIST_FWD = _os.environ.get("YT8M_GRU_PERSIST_FWD", "1") != "0"
GRU_PERSIST_BWD = _os.environ.get("YT8M_GRU_PERSIST_BWD", "0") != "0"
PERSIST_CHECK = _os.environ.get("YT8M_PERSIST_CHECK", "0") == "1"  # debug: synchronise + check the timeout word after each launch
X3 = _os.environ.get("YT8M_GEMM_X3", "1") != "0"      # hoisted fp32 products on the bf16 pipe (three-plane split, csrc/gemm_x3.hip)
X3_MIN_ROWS = 1024                                      # F * B below which the fp32-MFMA kernel's smaller tiles win
REC_BF16 = _os.environ.get("YT8M_REC_BF16", "1") != "0"       # compute_dtype=bfloat16: bf16 operands for the recurrent product too (csrc/lstm_bf16.hip); False = hoisted
                      # products only


NATIVE_STACK = _os.environ.get("YT8M_LSTM_STACK_NATIVE", "1") != "0"   # whole stack behind yt8m_lstm_stack_fwd / _bwd (csrc/lstm_stack.hip)
NATIVE_DROPOUT = _os.environ.get("YT8M_NATIVE_DROPOUT", "1") != "0"   # DropoutWrapper inside yt8m_lstm_stack_fwd / _bwd (round 6)
_STACK_SCRATCH = {}   # (device, calling stream, description) -> zero-initialised scratch of the native stack (resident, like _PERSIST_WS)
NATIVE_CALLS = {"fwd": 0, "bwd": 0}     # how often the native path ran (tests assert that it engaged)


# --compute_dtype=bfloat16 on the native stack (round 4): hoisted products on one-plane bf16 operand images (b1 kernel), the
# recurrence itself on the fp32-grade persistent kernels -- i.e. "bf16 operands" applies to the input projections, dx and the
# weight gradients only (torch_ref.lstm_stack(bf16_operands="input") is its value emulation).  YT8M_LSTM_STACK_BF16=0 keeps the
# Python orchestration with the per-step bf16 recurrence kernels (csrc/lstm_bf16.hip).
NATIVE_BF16 = _os.environ.get("YT8M_LSTM_STACK_BF16", "1") != "0"


def _stack_desc(B, F, D, H, L, u8, forget_bias, need_dx, bf16=False, keep_prob=None, seeds=None):
    # 0 = the library's own partition (csrc/lstm_stack.hip: one forward launch per layer, three unequal backward parts) unless the
    # environment / a test names a number of parts
    fwd = PERSIST_FWD_CHUNKS if (PERSIST_FWD_CHUNKS != 1 or "YT8M_LSTM_PERSIST_FWD_CHUNKS" in _os.environ) else 0
    bwd = PERSIST_BWD_CHUNKS if (PERSIST_BWD_CHUNKS != 3 or "YT8M_LSTM_PERSIST_BWD_CHUNKS" in _os.environ) else 0
    d = _lib.LstmStackDesc(int(B), int(F), int(D), int(H), int(L), int(bool(u8)) | (2 if bf16 else 0), float(forget_bias), int(fwd),
                           int(bwd), int(bool(need_dx)))                       # input_u8: bit 0 uint8 frames, bit 1 bf16 operand images
    if keep_prob is not None and float(keep_prob) < 1.0:                       # DropoutWrapper(input_keep_prob) inside the stack (ABI 4)
        d.input_keep_prob = float(keep_prob)
        for l, sd in enumerate(list(seeds)[:8]):
            d.dropout_seed[l] = int(sd) & 0xFFFFFFFFFFFFFFFF
    return d


def _stack_scratch(dev, main, desc):
    # (the size is part of the key: the library's layout depends on its knobs too -- a test that flips YT8M_STACK_H2 in-process must
    # not be handed the other layout's buffer)
    need = _lib.lib().yt8m_lstm_stack_scratch_bytes(ctypes.byref(desc))
    key = (dev.index, main.cuda_stream, need) + tuple(getattr(desc, f) for f, _ in desc._fields_
                                                      if f not in ("forget_bias", "input_keep_prob", "reserved0", "dropout_seed"))
    ent = _STACK_SCRATCH.get(key)
    if ent is not None:
        _STACK_SCRATCH[key] = _STACK_SCRATCH.pop(key)                # least recently USED goes first (dicts keep insertion order)
    else:
        if len(_STACK_SCRATCH) >= _STACK_SCRATCH_MAX:                # a few GB each: keep the table small
            _retire(_STACK_SCRATCH.pop(next(iter(_STACK_SCRATCH))), _check_stack)
        ent = (torch.zeros(need, dtype=torch.uint8, device=dev), main, desc)      # zeroed ONCE: the sticky time-out words start clear
        _STACK_SCRATCH[key] = ent
    return ent[0]


def _check_stack(scratch, stream, desc):
    with torch.cuda.device(scratch.device):
        _lib.check(_lib.lib().yt8m_lstm_stack_status(ctypes.byref(desc), _p(scratch), ctypes.c_void_p(stream.cuda_stream)))


def _tape_view(lib, desc, tape, layer, which, shape):
    ptr = ctypes.c_void_p()
    _lib.check(lib.yt8m_lstm_stack_view(ctypes.byref(desc), _p(tape), layer, which, ctypes.byref(ptr)))
    off = ptr.value - tape.data_ptr()
    n = 4
    for d in shape:
        n *= d
    return tape[off:off + n].view(torch.float32).view(*shape)


# Per-layer reports of the native stack's gradients under a reducer (each layer's collective starts when THAT layer's last
# weight-gradient product is done: yt8m_lstm_stack_layer_done_wait).  Opt-in: on one GPU with a 1-rank RCCL group the extra stream
# costs 1.5-1.7 ms of the 23.6 ms step (profiles/r3_force_reducer.md; more hardware queues -- GPU_MAX_HW_QUEUES=8 -- do not help),
# more than the ~0.3 ms of layer-1 all-reduce it can hide on eight.
DP_LAYER_BUCKETS = _os.environ.get("YT8M_DP_LAYER_BUCKETS", "0") != "0"
DP_LAYER_BUCKETS_USED = [0]
_DP_SIDE = {}


def _dp_side_stream(dev):
    st = _DP_SIDE.get(dev)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _DP_SIDE[dev] = st
    return st


# ---- clip + Adam of the already-final gradients inside the recurrent stack's backward pass ---------------------------------------
# train.TrainGraph.step (single device) leaves its optimiser arguments in graph.early_optimizer before backward(); when the native
# stack's backward pass starts, every trainable variable that has reported its gradient final (Variable.grad_done: for LstmModel the
# MoE head, 97 of 114 M parameters) gets its per-tensor clip + Adam update on the library's weight-gradient stream, behind the
# operand-image preparation and while the top layer's first recurrence runs alone on half the chip
# (yt8m_lstm_stack_set_early_optimizer: the library enqueues it).  The same kernels on the same numbers as the end-of-step pass -- which then only covers what
# is left (graph.early_done).  A side stream that started at once took CUs from that recurrence (25.2 -> 26.7 ms/step in round 2);
# here the recurrence is enqueued first.  YT8M_EARLY_ADAM=0 turns it off.
EARLY_ADAM = _os.environ.get("YT8M_EARLY_ADAM", "1") != "0"
EARLY_ADAM_RUNS = [0]      # how often the early pass ran (tests assert that it engaged)


class _EarlyOpt(object):
    """The early pass as a descriptor the library consumes (yt8m_lstm_stack_set_early_optimizer, round 5): yt8m_lstm_stack_bwd
    enqueues sqnorm / Adam (/ the image tile pass) of `ranges` itself -- no Python runs inside the native call any more (the ctypes
    callback of rounds 3-4 swallowed exceptions and had no equivalent for a non-Python host)."""

    def __init__(self, graph, ranges):
        self.graph, self.ranges = graph, ranges
        a = graph.early_optimizer
        o = _lib.OptRanges()
        o.w, o.m, o.v, o.g = graph.params.data_ptr(), graph.adam_m.data_ptr(), graph.adam_v.data_ptr(), graph.grads.data_ptr()
        o.chunks, o.tensor_chunk_start = graph.chunks.data_ptr(), graph.chunk_start_dev.data_ptr()
        self._tcs = (ctypes.c_int32 * len(graph.chunk_start))(*graph.chunk_start)
        o.tensor_chunk_start_host = ctypes.cast(self._tcs, ctypes.c_void_p)
        o.l2, o.partial, o.norms = graph.l2.data_ptr(), graph.partial.data_ptr(), graph.norms.data_ptr()
        wi = getattr(graph, "wimg", None)
        if wi is not None and wi.active:                             # image-owning matrices: the tile pass rewrites their images
            self._jt = (ctypes.c_int32 * len(wi.job_tensor))(*wi.job_tensor)
            self._tb = (ctypes.c_int64 * len(wi.tile_base))(*wi.tile_base)
            o.skip_tensor, o.jobs = wi.skip_dev.data_ptr(), wi.jobs_dev.data_ptr()
            o.job_tensor_host, o.job_tile_base_host = ctypes.cast(self._jt, ctypes.c_void_p), ctypes.cast(self._tb, ctypes.c_void_p)
            o.njobs = len(wi.job_tensor)
        o.nranges = len(ranges)
        for i, (lo, hi) in enumerate(ranges):
            o.range_lo[i], o.range_hi[i] = lo, hi
        o.gscale, o.clip, o.lr_t, o.beta1, o.beta2, o.eps = 1.0, a["clip"], a["lr_t"], a["beta1"], a["beta2"], a["eps"]
        side = getattr(graph, "side_pending", None)
        if side:                                                     # the head's weight gradients may still be on their side stream
            o.after_stream = side[-1].cuda_stream
        self.desc = o

    def finish(self, lib, ok):
        _lib.check(lib.yt8m_lstm_stack_set_early_optimizer(None))    # (a call that failed before the window keeps it armed)
        self.graph.early_active = None
        if ok:                                                       # the native call enqueued the pass: the end-of-step pass
            self.graph.early_done = list(self.ranges)                # covers the rest (train.TrainGraph.step)
            EARLY_ADAM_RUNS[0] += 1


def _ready_ranges(ready):
    """Maximal runs [lo, hi) of True in `ready` (trainable-variable indices whose gradients are final)."""
    ranges, i, n = [], 0, len(ready)
    while i < n:
        if ready[i]:
            j = i
            while j + 1 < n and ready[j + 1]:
                j += 1
            ranges.append((i, j + 1))
            i = j + 1
        else:
            i += 1
    return ranges


def _early_optimizer_hook(graph, lib):
    if not EARLY_ADAM or graph is None or getattr(graph, "early_optimizer", None) is None or graph.grad_ready_hook is not None:
        return None
    if getattr(graph, "early_done", None):
        return None                                                  # one recurrent stack per step takes the head
    tv = graph.trainable_variables()
    ranges = _ready_ranges([bool(getattr(v, "_done_reported", False)) and v.grad_written for v in tv])
    # worth a launch pair only for a substantial share of the parameters
    if not ranges or sum(tv[k].numel() for lo, hi in ranges for k in range(lo, hi)) < (1 << 22):
        return None
    if len(ranges) > 8:                                              # the descriptor carries eight; the rest waits for the end of the step
        ranges = sorted(sorted(ranges, key=lambda r: -sum(tv[k].numel() for k in range(*r)))[:8])
    e = _EarlyOpt(graph, ranges)
    _lib.check(lib.yt8m_lstm_stack_set_early_optimizer(ctypes.byref(e.desc)))
    graph.early_active = list(ranges)                                # Variable.grad_beta refuses later contributions to these
    return e


class _LstmStack(torch.autograd.Function):
    """MultiRNNCell([BasicLSTMCell] * L) under tf.nn.dynamic_rnn (W/all_frame_models/lstm_model.py:34-47), time-major, as ONE
    op so that the layers can be pipelined: the sequence is cut into time chunks; layer l's hoisted input projection of
    chunk c (a GEMM stream) starts as soon as layer l-1 has finished the recurrence of chunk c, and its own recurrence of
    chunk c (a high-priority stream per layer) follows -- while layer l-1 is already in chunk c+1.  The MFMA-bound GEMMs then
    overlap the latency-bound recurrence steps.  Backward runs the same wavefront in reverse time order (top layer first;
    the dx GEMM of a chunk releases the layer below) and every weight-gradient GEMM goes to a further stream, accumulating
    chunk by chunk.  Same kernels and the same per-step arithmetic as the unpipelined op; chunks = 1 is the sequential form.

    input_keep_prob < 1 is tf.contrib.rnn.DropoutWrapper(cell, input_keep_prob) around every layer
    (W/all_frame_models/lstm_memory_model.py:36-45): the INPUT of each layer (frames for layer 0, the outputs of the layer
    below otherwise -- never the recurrent state) goes through tf.nn.dropout with a fresh mask per time step.  Here: one
    Philox key per layer (seeds), element index = position in the [F,B,in] tensor, applied chunk by chunk on the layer's
    stream right before the projection GEMM (in place on the lower layer's output buffer, which nothing else reads) and
    replayed on the dx chunk in backward.

    bf16 (compute_dtype=bfloat16): the hoisted products (input projection, its weight gradients and dx) take bf16 copies of
    their operands on the bf16 MFMA path (fp32 accumulate, fp32 master weights and gradients); the recurrence -- state, gates,
    recurrent product -- stays fp32.

    apply(x_tm [F,B,D], token, num_frames, forget_bias, chunks, input_keep_prob, seeds, bf16, W_0, b_0, ..., W_{L-1}, b_{L-1})
      -> (out_top [F,B,H], c_0, h_0, ..., c_{L-1}, h_{L-1})"""

    @staticmethod
    def forward(ctx, x_tm, token, num_frames, forget_bias, chunks, input_keep_prob, seeds, bf16, *wb):
        lib = _lib.lib()
        nf = _nf(num_frames)
        q_raw = None
        L = len(wb) // 2
        Hs = [w.data.shape[1] // 4 for w in wb[0::2]]
        if x_tm.dtype == torch.uint8:
            B, F = x_tm.shape[0], x_tm.shape[1]                    # raw reader output is batch-major
        else:
            F, B = x_tm.shape[0], x_tm.shape[1]
        pers = PERSIST and all(lib.yt8m_lstm_persist_supported(B, h) for h in Hs)
        parts = _chunks(F, chunks)
        # fp32 stack on the persistent kernels: the partition is the stack's own (see PERSIST_FWD_CHUNKS), `chunks` is the caller's
        # hint for the per-step kernels' layer pipeline
        own = pers and not bf16 and _os.environ.get("YT8M_PERSIST_CUS") is None
        # The product's own configuration runs entirely inside the library (csrc/lstm_stack.hip: partition, streams, operand images,
        # product forms -- the path bench.py measures); everything else (caller's chunks, dropout, bf16 operands, per-step kernels,
        # tuning knobs) keeps the orchestration below, built from the same per-call entry points.
        drop_ = input_keep_prob is not None and float(input_keep_prob) < 1.0
        nat_ok = pers and _os.environ.get("YT8M_PERSIST_CUS") is None and (not bf16 or NATIVE_BF16)
        # (round 6: DropoutWrapper(input_keep_prob) stays on the native stack -- the mask is applied where the operand images are built,
        # csrc/lstm_stack.hip -- for a float input on the f16 product forms; yt8m_lstm_stack_supported refuses the rest and the
        # orchestration below takes over.  YT8M_NATIVE_DROPOUT=0 keeps dropout on the orchestration.)
        u8_in = x_tm.dtype == torch.uint8
        drop_native = drop_ and NATIVE_DROPOUT and not u8_in and not bf16 and L <= 8
        if (NATIVE_STACK and nat_ok and X3 and PERSIST_BWD and PERSIST_STEP_IMAGES and (not drop_ or drop_native) and not FWD_WAVEFRONT and
                PERSIST_FWD_CHUNKS > 0 and PERSIST_BWD_CHUNKS > 0 and BWD_CHUNKS == 0 and not BWD_PARTS and len(set(Hs)) == 1):
            u8 = x_tm.dtype == torch.uint8
            D0 = x_tm.shape[2]
            desc = _stack_desc(B, F, D0, Hs[0], L, u8, forget_bias, (not u8) and bool(ctx.needs_input_grad[0]), bf16=bool(bf16),
                               keep_prob=input_keep_prob if drop_ else None, seeds=seeds if drop_ else None)
            if (u8 or x_tm.dtype == torch.float32) and wb[0].data.shape[0] == D0 + Hs[0] and lib.yt8m_lstm_stack_supported(ctypes.byref(desc)):
                return _LstmStack._native_forward(ctx, lib, desc, x_tm, nf, wb)
        if own and PERSIST_FWD_CHUNKS > 0:
            parts = _chunks(F, PERSIST_FWD_CHUNKS)
        bwd_parts = _bwd_parts(F, _chunks(F, chunks), own)
        # Wavefront of half-chip forward recurrences (opt-in, see FWD_WAVEFRONT): with the recurrent product on the bf16 pipe a layer's
        # recurrence is bound by its dependency chain, not by matrix time, so two layers can run side by side on half the chip each
        # (8 chains per workgroup) at 11.4 us / step and layer against 8.5 on the whole chip one after the other; finer time chunks
        # shorten the ramps.  The backward pass keeps the caller's partition.
        half_fwd = (FWD_WAVEFRONT and pers and L >= 2 and PERSIST_STEP_IMAGES and not bf16 and F >= 8 * FWD_WAVEFRONT_CHUNKS and
                    all(lib.yt8m_lstm_persist_fwd_on_bf16_pipe(B, h) for h in Hs) and _os.environ.get("YT8M_PERSIST_CUS") is None)
        if half_fwd:
            parts = _chunks(F, max(int(chunks), FWD_WAVEFRONT_CHUNKS))
        if x_tm.dtype == torch.uint8:
            # raw reader output [B,F,D] (batch-major): the layer-0 projection takes the bytes themselves (csrc/u8proj.hip) --
            # one conversion pass writes (q - 128) as bf16 in time-major order (three copies side by side: the three bf16
            # terms of alpha W share one concatenated reduction), the row norms, and the fp32 time-major frames that only
            # the weight-gradient product still reads
            _dev(x_tm)
            q_raw = x_tm.contiguous()
            Bq, Fq, Dq = q_raw.shape
            rrow = torch.empty((Fq * Bq,), dtype=torch.float32, device=q_raw.device)
            x_tm = torch.empty((Fq, Bq, Dq), dtype=torch.float32, device=q_raw.device)
            Qb = Qimg = None
            # one-plane operand image for the x3 kernel (three exact products per element pair) when every time chunk starts on
            # a 32-row group of the image; otherwise three bf16 copies side by side for the plain bf16 kernel
            if X3 and Dq % 16 == 0 and all((t0 * Bq) % 32 == 0 for t0, _ in parts):
                Qimg = torch.empty(((Fq * Bq + 31) // 32) * (Dq // 16) * 1024, dtype=torch.uint8, device=q_raw.device)
                _lib.check(lib.yt8m_u8_frames_image(_p(q_raw), _p(nf), Bq, Fq, Dq, 1e-12, _p(Qimg), _p(x_tm), _p(rrow), _stream()))
            else:
                Qb = ops._bf16_empty(Fq * Bq, 3 * Dq, q_raw.device)
                _lib.check(lib.yt8m_u8_frames_to_bf16_tm(_p(q_raw), _p(nf), Bq, Fq, Dq, 1e-12, 3, _p(Qb), Qb.stride(0), _p(x_tm), _p(rrow),
                                                         _stream()))
        x_tm = _f32c(x_tm)
        _dev(x_tm)
        Ws, bs = wb[0::2], wb[1::2]
        assert (F, B) == tuple(x_tm.shape[:2])
        dev = x_tm.device
        main = torch.cuda.current_stream(dev)
        rs, gs, _ = _side_streams(dev, L, pers, separate_gemm=half_fwd and not FWD_WAVEFRONT_SHARED)
        ctx.pers = pers
        bf16 = bool(bf16) and B % 2 == 0 and min(T for _, T in parts) * B >= ops.BF16_MIN_ROWS
        drop = input_keep_prob is not None and float(input_keep_prob) < 1.0
        layers, inp = [], x_tm
        for l in range(L):                                          # every buffer comes from the main stream's pool
            if drop and l == 0:
                inp = torch.empty_like(x_tm)                        # dropped copy of the frames (the input itself is data)
            Din = inp.shape[2]
            H = Ws[l].data.shape[1] // 4
            assert Ws[l].data.shape[0] == Din + H, "cell weights must be [in + H, 4H]"
            st = dict(x=inp, Din=Din, H=H, W=Ws[l], b=bs[l],
                      z=torch.empty((F, B, 4 * H), dtype=torch.float32, device=dev),
                      cs=torch.empty((F + 1, B, H), dtype=torch.float32, device=dev),
                      hs=torch.empty((F + 1, B, H), dtype=torch.float32, device=dev),
                      out=torch.empty((F, B, H), dtype=torch.float32, device=dev))
            npk = lib.yt8m_lstm_packed_floats(B, H)
            st["Wp"] = torch.empty(npk, dtype=torch.float32, device=dev) if npk else None
            # persistent recurrence (csrc/lstm_persist.hip): one launch per (layer, chunk), W_h resident in registers
            pws = lib.yt8m_lstm_persist_workspace_bytes(B, H) if PERSIST else 0
            if pws and PERSIST_STEP_IMAGES:
                # one exchange image per step of the longest launch (forward or backward partition): XCD-L2-shared state fetch
                tmax = max([T for _, T in parts] + [T for _, T in bwd_parts])
                big = lib.yt8m_lstm_persist_workspace_bytes_steps(B, H, tmax)
                if big <= PERSIST_STEP_IMAGES_MAX_BYTES:
                    pws = big
            st["pws"] = _persist_ws(dev, main, l, pws) if pws else None
            layers.append(st)
            inp = st["out"]
        start = torch.cuda.Event()
        start.record(main)
        for l, st in enumerate(layers):
            gs[l].wait_event(start)
            with torch.cuda.stream(rs[l]):
                rs[l].wait_event(start)
                st["cs"][0].zero_()
                st["hs"][0].zero_()
                if st["Wp"] is not None:
                    _lib.check(lib.yt8m_lstm_pack(_p(st["W"].data[st["Din"]:]), 4 * st["H"], st["H"], _p(st["Wp"]), None, _stream()))
                st["bf16"] = bf16 and st["Din"] % 2 == 0
                # decided on the whole sequence, not the chunk: every partition of the time axis runs the same arithmetic
                st["x3"] = X3 and not st["bf16"] and F * B >= X3_MIN_ROWS and st["H"] >= 128
                if l == 0 and q_raw is not None and not drop and not st["bf16"]:
                    # 3-way bf16 split of (4/255) W_x, transposed ([4H, 3 D], K-contiguous) + the column sums of W_x
                    Din_, H_ = st["Din"], st["H"]
                    if Qimg is not None:
                        st["W3T"] = ops.x3_split(st["W"].data[:Din_], plain=False, trans=True, scale=4.0 / 255.0)[1]
                    else:
                        st["W3T"] = ops._bf16_empty(4 * H_, 3 * Din_, dev)
                        _lib.check(lib.yt8m_split3_bf16_t(_p(st["W"].data[:Din_]), 4 * H_, Din_, 4 * H_, 4.0 / 255.0, _p(st["W3T"]),
                                                          st["W3T"].stride(0), _stream()))
                    st["Wcs"] = torch.empty((4 * H_,), dtype=torch.float32, device=dev)
                    ops.colsum(st["W"].data[:Din_], st["Wcs"])
                if st["x3"] and "W3T" not in st:                    # W_x^T as the K-contiguous operand of the projection
                    st["WxT3"] = ops.x3_split(st["W"].data[:st["Din"]], plain=False, trans=True)[1]
                st["rec16"] = st["bf16"] and REC_BF16 and lib.yt8m_lstm_packed16_elems(B, st["H"]) > 0
                if st["rec16"]:
                    st["pws"] = None
                if st["bf16"]:                                      # W_x^T once per step, K-contiguous for the NT product
                    st["WxT"] = ops.cast_bf16(st["W"].data[:st["Din"]], transpose=True)
                if st["rec16"]:                                     # bf16 operands for the recurrent product too
                    H_ = st["H"]
                    st["hs16"] = torch.empty((F + 1, B, H_), dtype=torch.bfloat16, device=dev)
                    st["hs16"][0].zero_()
                    st["Wp16"] = torch.empty(H_ * 4 * H_, dtype=torch.bfloat16, device=dev)
                    _lib.check(lib.yt8m_lstm_pack_bf16(_p(st["W"].data[st["Din"]:]), 4 * H_, H_, _p(st["Wp16"]), None, _stream()))
        r_done = [[torch.cuda.Event() for _ in parts] for _ in range(L)]
        if half_fwd:
            _lib.check(lib.yt8m_lstm_persist_set_cus(torch.cuda.get_device_properties(dev).multi_processor_count // 2, -1))
        try:
            for c, (t0, T) in enumerate(parts):
                for l, st in enumerate(layers):
                    Din, H = st["Din"], st["H"]
                    with torch.cuda.stream(gs[l]):                      # hoisted input projection of the chunk
                        if l > 0:
                            gs[l].wait_event(r_done[l - 1][c])
                        if drop:
                            xc = st["x"][t0:t0 + T]
                            src = x_tm[t0:t0 + T] if l == 0 else xc
                            _lib.check(lib.yt8m_dropout_f32(_p(src), _p(xc), xc.numel(), float(input_keep_prob), int(seeds[l]),
                                                            t0 * B * Din, _stream()))
                        if "W3T" in st and Qimg is not None:
                            zc = st["z"][t0:t0 + T].view(T * B, 4 * H)
                            ws = ops._workspace(dev)
                            _lib.check(lib.yt8m_gemm_x1x3_nt(T * B, 4 * H, Din, _p(Qimg[(t0 * B // 32) * (Din // 16) * 1024:]), _p(st["W3T"].buf),
                                                             _p(zc), 4 * H, _p(st["b"].data), _p(rrow[t0 * B:]), _p(st["Wcs"]), U8_BETA,
                                                             _p(ws), ws.numel() * 4, _stream()))
                        elif "W3T" in st:
                            zc = st["z"][t0:t0 + T].view(T * B, 4 * H)
                            ops.gemm_bf16_nt_grouped([dict(A=Qb[t0 * B:(t0 + T) * B], B=st["W3T"], out=zc)])
                            _lib.check(lib.yt8m_rowscale_bias_f32(_p(zc), T * B, 4 * H, 4 * H, _p(rrow[t0 * B:]), _p(st["Wcs"]),
                                                                  U8_BETA, _p(st["b"].data), _stream()))
                        elif st["bf16"]:
                            ops.gemm_bf16_nt_grouped([dict(A=ops.cast_bf16(st["x"][t0:t0 + T].view(T * B, Din)), B=st["WxT"],
                                                           out=st["z"][t0:t0 + T].view(T * B, 4 * H), bias=st["b"].data)])
                        elif st["x3"]:
                            xi = ops.x3_split(st["x"][t0:t0 + T].view(T * B, Din))[0]
                            ops.gemm_x3_grouped([dict(A=xi, B=st["WxT3"], out=st["z"][t0:t0 + T].view(T * B, 4 * H), bias=st["b"].data)])
                        else:
                            ops.gemm(st["x"][t0:t0 + T].view(T * B, Din), st["W"].data[:Din], out=st["z"][t0:t0 + T].view(T * B, 4 * H),
                                     bias=st["b"].data)
                        g_ev = torch.cuda.Event()
                        g_ev.record(gs[l])
                    with torch.cuda.stream(rs[l]):                      # recurrence steps of the chunk
                        rs[l].wait_event(g_ev)
                        ws = ops._workspace(dev)
                        if st["rec16"]:
                            _lib.check(lib.yt8m_lstm_steps_fwd_bf16(_p(st["z"]), _p(st["Wp16"]), _p(st["cs"]), _p(st["hs"]), _p(st["hs16"]),
                                                                    _p(st["out"]), _p(nf), t0, T, B, H, float(forget_bias), _stream()))
                        elif st["pws"] is not None:
                            _lib.check(lib.yt8m_lstm_persist_fwd(_p(st["z"]), _p(st["W"].data[Din:]), 4 * H, _p(st["cs"]), _p(st["hs"]),
                                                                 _p(st["out"]), _p(nf), t0, T, B, H, float(forget_bias), _p(st["pws"]),
                                                                 st["pws"].numel(), _stream()))
                            if PERSIST_CHECK:
                                _lib.check(lib.yt8m_lstm_persist_status(_p(st["pws"]), _stream()))
                        else:
                            _lib.check(lib.yt8m_lstm_steps_fwd(_p(st["z"]), _p(st["W"].data[Din:]), 4 * H, _p(st["Wp"]), _p(st["cs"]),
                                                               _p(st["hs"]), _p(st["out"]), _p(nf), t0, T, B, H, float(forget_bias),
                                                               _p(ws), ws.numel() * 4, _stream()))
                        r_done[l][c].record(rs[l])
        finally:
            if half_fwd:                                        # process-wide CU budget: restored whatever happens in the loop
                _lib.check(lib.yt8m_lstm_persist_set_cus(-1, -1))
        for l in range(L):
            main.wait_event(r_done[l][-1])
        # the backward pass may cut time differently (all buffers are whole-layer): its first recurrence chunk runs with nothing
        # beside it, so shorter chunks shorten that pipeline fill; the forward recurrence owns the chip and wants few launches
        ctx.layers, ctx.nf, ctx.parts = layers, nf, bwd_parts
        ctx.drop = (float(input_keep_prob), tuple(int(v) for v in seeds)) if drop else None
        ctx.set_materialize_grads(False)
        outs = [layers[-1]["out"]]
        for st in layers:
            outs += [st["cs"][F], st["hs"][F]]
        return tuple(outs)

    @staticmethod
    def _native_forward(ctx, lib, desc, x, nf, wb):
        x = x.contiguous()
        _dev(x)
        dev = x.device
        main = torch.cuda.current_stream(dev)
        L, B, F, H = desc.L, desc.B, desc.F, desc.H
        Ws, bs = wb[0::2], wb[1::2]
        for l, w in enumerate(Ws):
            assert w.data.is_contiguous() and tuple(w.data.shape) == ((desc.D if l == 0 else H) + H, 4 * H), "cell weights must be [in + H, 4H]"
        tape = torch.empty(lib.yt8m_lstm_stack_tape_bytes(ctypes.byref(desc)), dtype=torch.uint8, device=dev)
        scratch = _stack_scratch(dev, main, desc)
        Wp = (ctypes.c_void_p * L)(*[w.data.data_ptr() for w in Ws])
        bp = (ctypes.c_void_p * L)(*[b.data.data_ptr() for b in bs])
        _lib.check(lib.yt8m_lstm_stack_fwd(ctypes.byref(desc), _p(x), _p(nf), Wp, bp, _p(tape), tape.numel(), _p(scratch), scratch.numel(),
                                           _stream()))
        NATIVE_CALLS["fwd"] += 1
        if PERSIST_CHECK:
            _check_stack(scratch, main, desc)
        ctx.native = (desc, tape, scratch, x, nf, Ws, bs)
        g_ = Ws[0]._graph
        if g_ is not None and torch.is_grad_enabled():
            # the stack's backward pass will follow whatever consumes these outputs: ops that run before it in the backward pass (the
            # classifier head) may leave their weight gradients to a side stream instead of the chain the first recurrence waits on
            g_.defer_head_dw = True
        ctx.layers = None
        ctx.set_materialize_grads(False)
        outs = [_tape_view(lib, desc, tape, L - 1, 0, (F, B, H))]
        for l in range(L):
            outs += [_tape_view(lib, desc, tape, l, 1, (B, H)), _tape_view(lib, desc, tape, l, 2, (B, H))]
        return tuple(outs)

    @staticmethod
    def _native_backward(ctx, dout_top, dfinal):
        desc, tape, scratch, x, nf, Ws, bs = ctx.native
        ctx.native = None
        lib = _lib.lib()
        L = desc.L
        dev = x.device
        keep = []

        def c32(t):
            if t is None:
                return None
            t = _f32c(t)
            keep.append(t)
            return t

        dout_top = c32(dout_top)
        dcs = [c32(dfinal[2 * l]) for l in range(L)]
        dhs = [c32(dfinal[2 * l + 1]) for l in range(L)]
        arr = lambda ts: (ctypes.c_void_p * L)(*[None if t is None else t.data_ptr() for t in ts])
        dW = [w.grad if w.grad is not None else None for w in Ws]
        db = [b.grad if b.grad is not None else None for b in bs]
        for g in dW + db:
            assert g is None or g.is_contiguous()
        bW = (ctypes.c_float * L)(*[float(w.grad_beta()) if w.grad is not None else 0.0 for w in Ws])
        bb = (ctypes.c_float * L)(*[float(b.grad_beta()) if b.grad is not None else 0.0 for b in bs])
        dx = torch.empty((desc.F, desc.B, desc.D), dtype=torch.float32, device=dev) if desc.need_dx else None
        Wp = (ctypes.c_void_p * L)(*[w.data.data_ptr() for w in Ws])
        early = _early_optimizer_hook(Ws[0]._graph, lib)
        ok = False
        try:
            _lib.check(lib.yt8m_lstm_stack_bwd(ctypes.byref(desc), _p(x), _p(nf), Wp, _p(tape), tape.numel(), _p(scratch), scratch.numel(),
                                               _p(dout_top), arr(dcs), arr(dhs), arr(dW), arr(db), bW, bb, _p(dx), _stream()))
            ok = True
        finally:
            if early is not None:
                early.finish(lib, ok)
        NATIVE_CALLS["bwd"] += 1
        ops.join_side_work(Ws[0]._graph)                             # gradients an earlier op left on a side stream: visible from here on
        if PERSIST_CHECK:
            _check_stack(scratch, torch.cuda.current_stream(dev), desc)
        g = Ws[0]._graph
        if g is not None and g.grad_ready_hook is not None and DP_LAYER_BUCKETS:
            # data parallel: report each layer's gradients from a side stream that waits for THAT layer only (the library records
            # the point on its weight-gradient stream: layer L-1's gradients are final a whole last part before layer 0's), so
            # the reducer's collective for layer L-1 is on the wire while layer 0's last weight-gradient products still run
            side = _dp_side_stream(dev)
            for l in reversed(range(L)):
                _lib.check(lib.yt8m_lstm_stack_layer_done_wait(l, ctypes.c_void_p(side.cuda_stream)))
                with torch.cuda.stream(side):
                    for v in (Ws[l], bs[l]):
                        if v.grad is not None:
                            v.grad_done()
            DP_LAYER_BUCKETS_USED[0] += 1
        else:
            for v in list(Ws) + list(bs):
                if v.grad is not None:
                    v.grad_done()
        return (dx, None, None, None, None, None, None, None) + (None,) * (2 * L)

    @staticmethod
    def backward(ctx, dout_top, *dfinal):
        if getattr(ctx, "native", None) is not None:
            return _LstmStack._native_backward(ctx, dout_top, dfinal)
        layers, nf, parts = ctx.layers, ctx.nf, ctx.parts
        ctx.layers = None
        L = len(layers)
        F, B, _ = layers[0]["x"].shape
        dev = layers[0]["x"].device
        lib = _lib.lib()
        main = torch.cuda.current_stream(dev)
        rs, gs, sw = _side_streams(dev, L, ctx.pers)
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(layers[0]["x"]) if need_dx else None
        for st in layers:
            st.pop("WxT", None)
            st.pop("WxT3", None)
            st.pop("W3T", None)
            st.pop("Wcs", None)
            st.pop("Wp16", None)
            st.pop("hs16", None)
        for l, st in enumerate(layers):                            # buffers come from the main stream's pool
            H = st["H"]
            st["dz"] = torch.empty((F, B, 4 * H), dtype=torch.float32, device=dev)
            st["work"] = torch.empty((4, B, H), dtype=torch.float32, device=dev)
            st["dout"] = (None if dout_top is None else _f32c(dout_top)) if l == L - 1 else \
                torch.empty((F, B, H), dtype=torch.float32, device=dev)
            npk = lib.yt8m_lstm_packed_floats(B, H)
            st["Wq"] = torch.empty(npk, dtype=torch.float32, device=dev) if npk else None
        start = torch.cuda.Event()
        start.record(main)
        sw.wait_event(start)
        for l, st in enumerate(layers):
            H = st["H"]
            gs[l].wait_event(start)
            with torch.cuda.stream(rs[l]):
                rs[l].wait_event(start)
                dc, dh = dfinal[2 * l], dfinal[2 * l + 1]
                if dh is None:
                    st["work"][0].zero_()
                else:
                    st["work"][0].copy_(dh)
                if dc is None:
                    st["work"][1].zero_()
                else:
                    st["work"][1].copy_(dc)
                st["phase"] = 0
                if st["rec16"]:
                    st["dz16"] = torch.empty((F, B, 4 * H), dtype=torch.bfloat16, device=dev)
                    st["Wq16"] = torch.empty(H * 4 * H, dtype=torch.bfloat16, device=dev)
                    _lib.check(lib.yt8m_lstm_pack_bf16(_p(st["W"].data[st["Din"]:]), 4 * H, H, None, _p(st["Wq16"]), _stream()))
                elif st["Wq"] is not None:
                    _lib.check(lib.yt8m_lstm_pack(_p(st["W"].data[st["Din"]:]), 4 * H, H, None, _p(st["Wq"]), _stream()))
        wbeta = {}
        last = []
        for c in range(len(parts) - 1, -1, -1):
            t0, T = parts[c]
            dx_ev = None                                            # dx GEMM of the layer above for this chunk
            for l in range(L - 1, -1, -1):
                st = layers[l]
                Din, H, W, b = st["Din"], st["H"], st["W"], st["b"]
                dzc = st["dz"][t0:t0 + T].view(T * B, 4 * H)
                with torch.cuda.stream(rs[l]):
                    if dx_ev is not None:
                        rs[l].wait_event(dx_ev)
                    ws = ops._workspace(dev)
                    if st["rec16"]:
                        _lib.check(lib.yt8m_lstm_steps_bwd_bf16(_p(st["z"]), _p(st["Wq16"]), _p(st["cs"]), _p(st["dout"]), _p(st["dz"]),
                                                                _p(st["dz16"]), _p(st["work"]), st["phase"], _p(nf), t0, T, B, H,
                                                                _stream()))
                    elif st.get("pws") is not None and PERSIST_BWD and lib.yt8m_lstm_persist_bwd_supported(B, H):
                        # bias gradient as per-row sums in the epilogue: measured SLOWER (37.1 -> 40.3 ms/step: four more scattered
                        # read-modify-writes per lane and item sit in the CU's memory queue in front of the exchange loads); off
                        if PERSIST_DBROWS and "dbrows" not in st and b.grad is not None:
                            st["dbrows"] = torch.zeros((B, 4 * H), dtype=torch.float32, device=dev)
                        _lib.check(lib.yt8m_lstm_persist_bwd(_p(st["z"]), _p(W.data[Din:]), 4 * H, _p(st["cs"]), _p(st["dout"]),
                                                             _p(st["dz"]), _p(st["work"]), st["phase"], _p(st.get("dbrows")), _p(nf),
                                                             t0, T, B, H, _p(st["pws"]), st["pws"].numel(), _stream()))
                        if PERSIST_CHECK:
                            _lib.check(lib.yt8m_lstm_persist_status(_p(st["pws"]), _stream()))
                    else:
                        _lib.check(lib.yt8m_lstm_steps_bwd(_p(st["z"]), _p(W.data[Din:]), 4 * H, _p(st["Wq"]), _p(st["cs"]), _p(st["dout"]),
                                                           _p(st["dz"]), _p(st["work"]), st["phase"], _p(nf), t0, T, B, H, _p(ws),
                                                           ws.numel() * 4, _stream()))
                    st["phase"] = (st["phase"] + T) % 2
                    rb = torch.cuda.Event()
                    rb.record(rs[l])
                dx_ev = None
                dzb = dzT = None
                if st["bf16"]:                                      # dz feeds dx (plain) and both dW products (transposed)
                    with torch.cuda.stream(gs[l]):
                        gs[l].wait_event(rb)
                        if st["rec16"]:                         # the step kernels already wrote the plain bf16 copy
                            dzb, dzT = st["dz16"][t0:t0 + T].view(T * B, 4 * H), ops.cast_bf16(dzc, transpose=True)
                        else:
                            dzb, dzT = ops.cast_bf16_both(dzc)
                        if "Wxb" not in st:
                            st["Wxb"] = ops.cast_bf16(W.data[:Din])
                        cast_ev = torch.cuda.Event()
                        cast_ev.record(gs[l])
                dz3 = None
                if st["x3"] and (l > 0 or need_dx):                 # dz as stored feeds dx: on the layer stream (critical path);
                    with torch.cuda.stream(gs[l]):                  # its transposed image (both dW products) is made on `sw`
                        gs[l].wait_event(rb)
                        dz3 = ops.x3_split(dzc)[0]
                        if "Wx3" not in st:
                            st["Wx3"] = ops.x3_split(W.data[:Din])[0]
                if l > 0 or need_dx:
                    with torch.cuda.stream(gs[l]):
                        gs[l].wait_event(rb)
                        dst = layers[l - 1]["dout"] if l > 0 else dx
                        if st["x3"]:
                            ops.gemm_x3_grouped([dict(A=dz3, B=st["Wx3"], out=dst[t0:t0 + T].view(T * B, Din))])
                        elif st["bf16"]:
                            ops.gemm_bf16_nt_grouped([dict(A=dzb, B=st["Wxb"], out=dst[t0:t0 + T].view(T * B, Din))])
                        else:
                            ops.gemm(dzc, W.data[:Din], out=dst[t0:t0 + T].view(T * B, Din), transB=True)
                        if ctx.drop is not None:
                            ops.dropout_(dst[t0:t0 + T], ctx.drop[0], ctx.drop[1][l], t0 * B * Din)
                        dx_ev = torch.cuda.Event()
                        dx_ev.record(gs[l])
                        if c == 0:
                            last.append(dx_ev)
                with torch.cuda.stream(sw):
                    sw.wait_event(rb)
                    if W.grad is not None:
                        beta = wbeta.get(id(W))
                        if beta is None:
                            beta = W.grad_beta()
                            wbeta[id(W)] = 1.0
                        if st["bf16"]:
                            sw.wait_event(cast_ev)
                            dzT.record_stream(sw)
                            ops.gemm_bf16_nt_grouped([
                                dict(A=ops.cast_bf16(st["x"][t0:t0 + T].view(T * B, Din), transpose=True), B=dzT, out=W.grad[:Din], beta=beta),
                                dict(A=ops.cast_bf16(st["hs"][t0:t0 + T].view(T * B, H), transpose=True), B=dzT, out=W.grad[Din:], beta=beta)])
                        elif st["x3"]:
                            dzT3 = ops.x3_split(dzc, plain=False, trans=True)[1]
                            xT3 = ops.x3_split(st["x"][t0:t0 + T].view(T * B, Din), plain=False, trans=True)[1]
                            hT3 = ops.x3_split(st["hs"][t0:t0 + T].view(T * B, H), plain=False, trans=True)[1]
                            ops.gemm_x3_grouped([dict(A=xT3, B=dzT3, out=W.grad[:Din], beta=beta),
                                                 dict(A=hT3, B=dzT3, out=W.grad[Din:], beta=beta)])
                        else:
                            ops.gemm(st["x"][t0:t0 + T].view(T * B, Din), dzc, out=W.grad[:Din], transA=True, beta=beta, role="dw")
                            ops.gemm(st["hs"][t0:t0 + T].view(T * B, H), dzc, out=W.grad[Din:], transA=True, beta=beta, role="dw")
                    if b.grad is not None and ("dbrows" not in st or c == 0):
                        beta = wbeta.get(id(b))
                        if beta is None:
                            beta = b.grad_beta()
                            wbeta[id(b)] = 1.0
                        # persistent backward: the per-row sums are complete after the last (earliest) chunk -> one [B,4H] column sum
                        ops.colsum(st["dbrows"] if "dbrows" in st else dzc, b.grad.view(-1), beta=beta)
        fin = torch.cuda.Event()
        fin.record(sw)                                              # sw waited for every recurrence chunk
        main.wait_event(fin)
        for e in last:
            main.wait_event(e)
        for st in layers:
            if st["W"].grad is not None:
                st["W"].grad_done()
            if st["b"].grad is not None:
                st["b"].grad_done()
        return (dx, None, None, None, None, None, None, None) + (None,) * (2 * L)


def lstm_stack(x_tm, num_frames, weights_biases, forget_bias=1.0, chunks=4, input_keep_prob=None, seeds=None, bf16=False):
    """weights_biases: [(W_0, b_0), ...] Variables.  Returns (out_top, [(c_l, h_l), ...]).
    input_keep_prob < 1: DropoutWrapper(input_keep_prob) on every layer; seeds = one Philox key per layer (default: the
    graph's random stream)."""
    flat = [v for wb in weights_biases for v in wb]
    if input_keep_prob is not None and float(input_keep_prob) < 1.0 and seeds is None:
        seeds = [flat[0]._graph.next_random_seed() for _ in weights_biases]
    res = _LstmStack.apply(x_tm, _token(flat[0]._graph), num_frames, forget_bias, int(chunks), input_keep_prob,
                           tuple(seeds) if seeds is not None else None, bool(bf16), *flat)
    return res[0], [(res[1 + 2 * l], res[2 + 2 * l]) for l in range(len(weights_biases))]


class _AttnSoftmax(torch.autograd.Function):
    """mask * softmax over frames, renormalised (lstm_attention_max_pooling_model.py:59-60).  act, w: [B,F,A]."""

    @staticmethod
    def forward(ctx, act, num_frames):
        act = _f32c(act)
        _dev(act)
        B, F, A = act.shape
        w = torch.empty_like(act)
        nf = _nf(num_frames)
        _lib.check(_lib.lib().yt8m_attn_softmax_fwd(_p(act), _p(nf), _p(w), B, F, A, _stream()))
        ctx.save_for_backward(w)
        ctx.nf = nf
        return w

    @staticmethod
    def backward(ctx, dw):
        (w,) = ctx.saved_tensors
        dw = _f32c(dw)
        B, F, A = w.shape
        dact = torch.empty_like(w)
        _lib.check(_lib.lib().yt8m_attn_softmax_bwd(_p(w), _p(dw), _p(ctx.nf), _p(dact), B, F, A, _stream()))
        return dact, None


def attention_weights(act, num_frames):
    return _AttnSoftmax.apply(act, num_frames)


class _SoftmaxRows(torch.autograd.Function):
    """a = softmax over the last axis, zeroed on padding frames (NetVLAD assignment)."""

    @staticmethod
    def forward(ctx, s, num_frames):
        s = _f32c(s)
        _dev(s)
        B, F, K = s.shape
        a = torch.empty_like(s)
        nf = _nf(num_frames)
        _lib.check(_lib.lib().yt8m_softmax_rows_fwd(_p(s), _p(nf), _p(a), B, F, K, _stream()))
        ctx.save_for_backward(a)
        ctx.nf = nf
        return a

    @staticmethod
    def backward(ctx, da):
        (a,) = ctx.saved_tensors
        da = _f32c(da)
        B, F, K = a.shape
        ds = torch.empty_like(a)
        _lib.check(_lib.lib().yt8m_softmax_rows_bwd(_p(a), _p(da), _p(ctx.nf), _p(ds), B, F, K, _stream()))
        return ds, None


def masked_softmax_rows(s, num_frames):
    return _SoftmaxRows.apply(s, num_frames)


POOL_STREAM_MIN = 1 << 22      # elements of x from which the streaming pooling kernels replace the padded batched GEMM


def _pool_stream_ok(w, x):
    B, F, A = w.shape
    H = x.shape[2]
    return (A <= 16 and x.numel() >= POOL_STREAM_MIN and x.data_ptr() % 16 == 0
            and _lib.lib().yt8m_attn_pool_supported(B, F, A, H) == 1)


class _PoolTN(torch.autograd.Function):
    """C[b] = w[b]^T . x[b]   (w [B,F,A], x [B,F,H] -> [B,A,H]): attention pooling
    (lstm_attention_max_pooling_model.py:63) and NetVLAD aggregation (Appendix B).  A handful of attentions over a large
    frame block takes the streaming kernels (yt8m_attn_pool_*: x crosses HBM once per pass); otherwise one batched GEMM."""

    @staticmethod
    def forward(ctx, w, x):
        w, x = _f32c(w), _f32c(x)
        ctx.save_for_backward(w, x)
        if _pool_stream_ok(w, x):
            B, F, A = w.shape
            H = x.shape[2]
            C = torch.empty((B, A, H), dtype=torch.float32, device=x.device)
            _lib.check(_lib.lib().yt8m_attn_pool_fwd(_p(w), _p(x), _p(C), B, F, A, H, _stream()))
            return C
        return ops.gemm_batched(w, x, transA=True)

    @staticmethod
    def backward(ctx, dC):
        w, x = ctx.saved_tensors
        dC = _f32c(dC)
        if _pool_stream_ok(w, x):
            B, F, A = w.shape
            H = x.shape[2]
            dw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
            dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
            if dw is not None or dx is not None:
                _lib.check(_lib.lib().yt8m_attn_pool_bwd(_p(w), _p(x), _p(dC), _p(dw), _p(dx), B, F, A, H, _stream()))
            return dw, dx
        dw = ops.gemm_batched(x, dC, transB=True) if ctx.needs_input_grad[0] else None   # [F,H].[H,A]
        dx = ops.gemm_batched(w, dC) if ctx.needs_input_grad[1] else None                # [F,A].[A,H]
        return dw, dx


def pool_tn(w, x):
    return _PoolTN.apply(w, x)


# ---- attention branch straight from the reader's uint8 frames (csrc/gemm_skinny.hip, uint8 rows) -----------------------------
# x = diag(rs) (a0 q + c0): Dequantize (W/utils.py:23-38), mask and l2-normalise (W/all_feature_transform/default_transformer.py:4-8)
# of the frames W/readers.py:178-187 hands over as uint8, folded into the logit FC, the pooling and their gradients.

def u8_attention_supported(q, A):
    if q.dtype != torch.uint8 or q.dim() != 3 or not q.is_cuda:
        return False
    B, F, D = q.shape
    L = _lib.lib()
    return bool(D % 4 == 0 and D <= 2048 and L.yt8m_attn_pool_supported(B, F, A, D) == 1 and L.yt8m_skinny_supported(B * F, D, A) == 1)


def u8_frame_scales(q, num_frames, eps=1e-12):
    """rs [B,F] = 1 / ||a0 q + c0|| per frame, 0 for the padding frames."""
    q = q.contiguous()
    _dev(q)
    B, F, D = q.shape
    rs = torch.empty((B, F), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().yt8m_u8_frame_scales(_p(q), _p(_nf(num_frames)), B, F, D, float(eps), _p(rs), _stream()))
    return rs


def pool_u8_raw(w, q, rs):
    """C [B,A,D] = (w (.) rs)^T (a0 q + c0) without autograd (w [B,F,A])."""
    B, F, A = w.shape
    D = q.shape[2]
    C = torch.empty((B, A, D), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().yt8m_attn_pool_fwd_u8(_p(w), _p(q), _p(rs), _p(C), B, F, A, D, _stream()))
    return C


class _PoolU8(torch.autograd.Function):
    """att[b] = w[b]^T x[b] of W/all_frame_models/lstm_attention_max_pooling_model.py:63 with x from the raw frames; the
    gradient goes to the weights only (the frames are the input)."""

    @staticmethod
    def forward(ctx, w, q, rs):
        w = _f32c(w)
        _dev(w, q, rs)
        ctx.save_for_backward(q, rs)
        ctx.shape = tuple(w.shape)
        return pool_u8_raw(w, q, rs)

    @staticmethod
    def backward(ctx, dC):
        q, rs = ctx.saved_tensors
        dC = _f32c(dC)
        B, F, A = ctx.shape
        D = q.shape[2]
        dCsum = dC.sum(dim=2)                                         # [B,A]: the rank-1 remainder of the dequantise affine
        dw = torch.empty((B, F, A), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().yt8m_attn_pool_dw_u8(_p(q), _p(rs), _p(dC), _p(dCsum), _p(dw), B, F, A, D, _stream()))
        return dw, None, None


def pool_tn_u8(w, q, rs):
    return _PoolU8.apply(w, q, rs)


class _AttnLogitsU8(torch.autograd.Function):
    """act [B,F,A] = slim.fully_connected(concat(x, parts...)) of lstm_attention_max_pooling_model.py:51-56 /
    lstm_positional_attention_max_pooling_model.py:77-84 with x from the raw frames: W rows [0, D) meet the frames, the following rows
    the other parts in concatenation order -- kinds[i] = 0: a per-frame float tensor [B,F,K_i] (the LSTM outputs, a positional embedding),
    kinds[i] = 1: a per-video vector [B,K_i] that the reference tiles over the frames (the mean frame)."""

    @staticmethod
    def forward(ctx, token, W, b, q, rs, kinds, *parts):
        _dev(q, rs)
        B, F, D = q.shape
        N = W.data.shape[1]
        parts = [_f32c(p).reshape(B * F, p.shape[-1]) if k == 0 else _f32c(p) for k, p in zip(kinds, parts)]
        assert W.data.shape[0] == D + sum(p.shape[1] for p in parts), "parts do not add up to the weight's input width"
        Wx = W.data[:D]
        cs = Wx.sum(dim=0)
        y = torch.empty((B * F, N), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().yt8m_skinny_fwd_u8(_p(q), D, _p(Wx), Wx.stride(0), _p(b.data if b is not None else None), _p(rs), _p(cs),
                                                 _p(y), N, B * F, D, N, 0.0, _stream()))
        k0 = D
        for kind, p in zip(kinds, parts):
            K = p.shape[1]
            Wi = W.data[k0:k0 + K]
            if kind == 1:
                t = ops.gemm(p, Wi)                                   # [B, N]: tiny
                y.view(B, F, N).add_(t.view(B, 1, N))
            elif ops.skinny_ok(B * F, K, N, p):                       # per-frame float part: accumulates into y
                ops.skinny_fwd(p, Wi, None, y, beta=1.0)
            else:
                ops.gemm(p, Wi, out=y, beta=1.0)
            k0 += K
        ctx.save_for_backward(q, rs, *parts)
        ctx.W, ctx.b, ctx.kinds = W, b, tuple(kinds)
        return y.view(B, F, N)

    @staticmethod
    def backward(ctx, dy):
        q, rs = ctx.saved_tensors[:2]
        parts = ctx.saved_tensors[2:]
        W, b, kinds = ctx.W, ctx.b, ctx.kinds
        B, F, D = q.shape
        dy = _f32c(dy).view(B * F, -1)
        N = dy.shape[1]
        wbeta = W.grad_beta() if (W.trainable and W.grad is not None) else None
        if wbeta is not None:
            gx = W.grad[:D]
            ws = ops._workspace(q.device)
            _lib.check(_lib.lib().yt8m_skinny_dw_u8(_p(q), D, _p(dy), N, _p(rs), _p(gx), gx.stride(0), B * F, D, N, float(wbeta),
                                                    _p(ws), ws.numel() * 4, _stream()))
        dparts = []
        dyg = None
        k0 = D
        for i, (kind, p) in enumerate(zip(kinds, parts)):
            K = p.shape[1]
            Wi = W.data[k0:k0 + K]
            need = ctx.needs_input_grad[6 + i]
            if kind == 1:
                if dyg is None:
                    dyg = dy.view(B, F, N).sum(dim=1)                 # [B, N]
                if wbeta is not None:
                    ops.gemm(p, dyg, out=W.grad[k0:k0 + K], transA=True, beta=wbeta, role="dw")
                dparts.append(ops.gemm(dyg, Wi, transB=True) if need else None)
            else:
                sk = ops.skinny_ok(B * F, K, N, p, dy)
                if wbeta is not None:
                    if sk:
                        ops.skinny_dw(p, dy, W.grad[k0:k0 + K], beta=wbeta)
                    else:
                        ops.gemm(p, dy, out=W.grad[k0:k0 + K], transA=True, beta=wbeta, role="dw")
                dparts.append(((ops.skinny_dx(dy, Wi) if sk else ops.gemm(dy, Wi, transB=True)).view(B, F, K)) if need else None)
            k0 += K
        if wbeta is not None:
            W.grad_done()
        if b is not None and b.trainable and b.grad is not None:
            ops.colsum(dy, b.grad.view(-1), beta=b.grad_beta())
            b.grad_done()
        return (None, None, None, None, None, None) + tuple(dparts)


def attention_logits_u8(q, rs, mean_x, W, b, parts=None):
    """Default form (parts None): concat(x, tile(mean_x)).  parts: the tensors that follow the frames in the concatenation, in order --
    [B,F,K] per-frame float tensors and / or [B,K] per-video vectors (tiled over the frames)."""
    if parts is None:
        parts = [mean_x]
    kinds = tuple(1 if p.dim() == 2 else 0 for p in parts)
    return _AttnLogitsU8.apply(_token(W._graph), W, b, q, rs, kinds, *parts)


U8_ALPHA = 4.0 / 255.0


def u8_cnn_supported(q):
    """The einsum CNN of W/all_frame_models/cnn_deep_combine_chain_model.py:60-82 straight from the reader's bytes (u8_cnn): the frame
    image kernels want D % 16 == 0, the shifted weight-gradient products whole K blocks per shift (B % 16 == 0)."""
    if q.dtype != torch.uint8 or q.dim() != 3 or not q.is_cuda:
        return False
    B, F, D = q.shape
    return bool(D % 16 == 0 and 16 <= D <= 2048 and B % 16 == 0 and F >= 1 and _lib.lib().yt8m_u8_proj_supported(D))


class U8FrameImages(object):
    """Operand images of one batch of raw frames q [B,F,D] uint8, made ONCE per step and shared by every product that reads the frames:
    `img` = (q - 128) as a one-plane half image in TIME-major row order t B + b (exact; yt8m_u8_frames_image_f16), `r` [F B] = 1 / ||a0 q + c0||
    per frame (0 on the padding frames), `trans()` = the transposed image (K = frame rows) for weight gradients, made on first use."""

    def __init__(self, q, num_frames, eps=1e-12):
        q = q.contiguous()
        _dev(q)
        self.q, self.nf = q, _nf(num_frames)
        self.B, self.F, self.D = q.shape
        M = self.F * self.B
        self.img = torch.empty(((M + 31) // 32) * (self.D // 16) * 1024, dtype=torch.uint8, device=q.device)
        self.r = torch.empty((M,), dtype=torch.float32, device=q.device)
        _lib.check(_lib.lib().yt8m_u8_frames_image_f16(_p(q), _p(self.nf), self.B, self.F, self.D, float(eps), _p(self.img), None, _p(self.r),
                                                       _stream()))
        self._t = None

    def trans(self):
        if self._t is None:
            M = self.F * self.B
            self._t = torch.empty(((self.D + 31) // 32) * ((M + 15) // 16) * 1024, dtype=torch.uint8, device=self.q.device)
            _lib.check(_lib.lib().yt8m_u8_frames_image_t_f16(_p(self.q), _p(self.nf), self.B, self.F, self.D, _p(self._t), _stream()))
        return self._t


def _u8_cnn_dense(frames, filters):
    """y [F B rows (t B + b), sum N_k]: for every filter k and shift i < fs_k
        y_k[rows i B ..] += x[rows .. M - i B] . W_k[i D : (i + 1) D]
    as ONE yt8m_gemm_h1x2_nt_ex launch on the frames' half image (two f16 products against the K range [i D, (i + 1) D) of the filter's
    half-plane image (alpha W_k)^T -- one image and one device-measured scale per filter --, dequantise / l2-normalise affine in the
    epilogue), accumulated in place: in time-major row order a shift by i frames is a row offset of i B."""
    B, F, D = frames.B, frames.F, frames.D
    M = F * B
    dev = frames.q.device
    lib = _lib.lib()
    Ntot = sum(W.data.shape[1] for W in filters)
    y = torch.empty((M, Ntot), dtype=torch.float32, device=dev)
    ws = ops._workspace(dev)
    c0 = 0
    for W in filters:
        assert W.data.shape[0] % D == 0 and W.data.is_contiguous(), "cnn filter must be [fs * D, N]"
        fs, N = W.data.shape[0] // D, W.data.shape[1]
        _, w2 = ops.h2_split(W.data, plain=False, trans=True, scale=U8_ALPHA, dynamic=True)     # [N rows, K = fs D]
        cs = torch.empty((fs, N), dtype=torch.float32, device=dev)                                  # column sums per shift slice
        for i in range(fs):
            ops.colsum(W.data[i * D:(i + 1) * D], cs[i])
        for i in range(fs):
            rows = M - i * B
            if rows <= 0:
                continue
            bptr = ctypes.c_void_p(w2.buf.data_ptr() + i * (D // 16) * 2048)                    # K blocks of 16: two 1 KiB half planes each
            cptr = ctypes.c_void_p(y.data_ptr() + (i * B * Ntot + c0) * 4)
            _lib.check(lib.yt8m_gemm_h1x2_nt_ex(rows, N, D, _p(frames.img), 0, bptr, fs * D // 16 if fs > 1 else 0, cptr, Ntot, None, 1.0,
                                                _p(w2.dinv), _p(frames.r), _p(cs[i]), U8_BETA, 1.0 if i else 0.0, _p(ws), ws.numel() * 4,
                                                _stream()))
        c0 += N
    return y


def u8_hoisted_supported(q):
    """A recurrent layer's hoisted input projection straight from the reader's bytes (u8_hoisted_fwd / _dw: GRU and LayerNorm-LSTM layers
    outside the native stack): D % 16 == 0, whole K blocks of frame rows for the weight gradient (F B % 16 == 0)."""
    if q.dtype != torch.uint8 or q.dim() != 3 or not q.is_cuda:
        return False
    B, F, D = q.shape
    return bool(D % 16 == 0 and 16 <= D <= 2048 and (F * B) % 16 == 0 and F >= 1 and _lib.lib().yt8m_u8_proj_supported(D))


def u8_hoisted_fwd(frames, Wrows, bias, out2d):
    """out2d [F B rows (t B + b), N] = x . Wrows (+ bias), x = the dequantised, l2-normalised, padding-masked frames (W/utils.py:23-38,
    readers.py:178-187, default_transformer.py:4-8): two f16 products of the frames' half image against (alpha Wrows)^T under a
    device-measured scale, the affine remainder in the epilogue (yt8m_gemm_h1x2_nt_ex) -- the form the native LSTM stack's layer 0 takes."""
    M, D = frames.F * frames.B, frames.D
    N = Wrows.shape[1]
    dev = frames.q.device
    _, w2 = ops.h2_split(Wrows, plain=False, trans=True, scale=U8_ALPHA, dynamic=True)
    cs = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.colsum(Wrows, cs)
    ws = ops._workspace(dev)
    _lib.check(_lib.lib().yt8m_gemm_h1x2_nt_ex(M, N, D, _p(frames.img), 0, _p(w2.buf), 0, _p(out2d), N, _p(bias), 1.0, _p(w2.dinv),
                                               _p(frames.r), _p(cs), U8_BETA, 0.0, _p(ws), ws.numel() * 4, _stream()))


def u8_hoisted_dw(frames, dz2d, gWrows, beta):
    """gWrows [D, N] (beta = 0 / 1: overwrite / accumulate) (+)= x^T . dz2d: the transposed byte image against (r (.) dz)^T written by one
    pass over dz (yt8m_h2_split_ex), the affine remainder as a rank-1 term of the epilogue."""
    M, D = frames.F * frames.B, frames.D
    N = dz2d.shape[1]
    dev = dz2d.device
    lib = _lib.lib()
    nb = max(lib.yt8m_x3_image_bytes(N, M) // 3 * 2, 16)
    word = ops.h2_absmax(dz2d)
    dzT = torch.empty(nb, dtype=torch.uint8, device=dev)
    dzTs = torch.empty(nb, dtype=torch.uint8, device=dev)
    ntile = (M + 63) // 64
    cp = torch.empty((ntile, N), dtype=torch.float32, device=dev)
    cps = torch.empty((ntile, N), dtype=torch.float32, device=dev)
    _lib.check(lib.yt8m_h2_split_ex(_p(dz2d), M, N, N, 1.0, _p(word), _p(frames.r), None, _p(dzT), _p(dzTs), _p(cp), _p(cps), _stream()))
    csr = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.colsum(cps, csr)
    ws = ops._workspace(dev)
    _lib.check(lib.yt8m_gemm_h1x2_nt_ex(D, N, M, _p(frames.trans()), (M + 15) // 16, _p(dzTs), 0, _p(gWrows), N, None, U8_ALPHA, _p(word), None,
                                        _p(csr), U8_BETA / U8_ALPHA, float(beta), _p(ws), ws.numel() * 4, _stream()))


class _CnnU8(torch.autograd.Function):
    """cnn_output [B,F,sum N_k] of cnn_deep_combine_chain_model.py:60-82 -- for every filter k of length fs_k, einsum("ijk,kl->ijl") of
    concat(x, x shifted by 1 frame, ..., by fs_k - 1 frames) with W_k [fs_k D, N_k] -- without the concatenations and without a float
    copy of the frames (_u8_cnn_dense).  Backward: the slice's weight gradient x[.. M - i B]^T . dy_k[i B ..] from the transposed byte
    image at K offset 0 (the recurrent stack's layer-0 form).  The frames are data: no dx.  (The model pools this output over the frames:
    _PooledCnnU8 is the form it uses; this one is the plain `cnn` of the reference class.)"""

    @staticmethod
    def forward(ctx, token, frames, *filters):
        y = _u8_cnn_dense(frames, filters)
        ctx.frames, ctx.filters = frames, filters
        return y.view(frames.F, frames.B, y.shape[1]).transpose(0, 1).contiguous()          # [B,F,N] (layout glue)

    @staticmethod
    def backward(ctx, dy):
        frames, filters = ctx.frames, ctx.filters
        B, F, D = frames.B, frames.F, frames.D
        M = F * B
        dev = dy.device
        lib = _lib.lib()
        Ntot = dy.shape[2]
        dyt = _f32c(dy).transpose(0, 1).contiguous().view(M, Ntot)      # time-major rows, as the images
        ws = ops._workspace(dev)
        nb = lambda rows_, K: max(lib.yt8m_x3_image_bytes(rows_, K) // 3 * 2, 16)
        qT = None
        c0 = 0
        for W in filters:
            fs, N = W.data.shape[0] // D, W.data.shape[1]
            if W.trainable and W.grad is not None:
                wbeta = float(W.grad_beta())
                if qT is None:
                    qT = frames.trans()
                for i in range(fs):
                    rows = M - i * B
                    gW = W.grad[i * D:(i + 1) * D]
                    if rows <= 0:
                        if wbeta == 0.0:
                            gW.zero_()
                        continue
                    part = dyt[i * B:, c0:c0 + N].contiguous()          # dy of the frames that saw this shift
                    word = ops.h2_absmax(part)
                    dzT = torch.empty(nb(N, rows), dtype=torch.uint8, device=dev)
                    dzTs = torch.empty(nb(N, rows), dtype=torch.uint8, device=dev)
                    ntile = (rows + 63) // 64
                    cp = torch.empty((ntile, N), dtype=torch.float32, device=dev)
                    cps = torch.empty((ntile, N), dtype=torch.float32, device=dev)
                    _lib.check(lib.yt8m_h2_split_ex(_p(part), rows, N, N, 1.0, _p(word), _p(frames.r), None, _p(dzT), _p(dzTs), _p(cp),
                                                    _p(cps), _stream()))
                    csr = cps.sum(0).contiguous()
                    _lib.check(lib.yt8m_gemm_h1x2_nt_ex(D, N, rows, _p(qT), (M + 15) // 16, _p(dzTs), 0, _p(gW), N, None, U8_ALPHA, _p(word),
                                                        None, _p(csr), U8_BETA / U8_ALPHA, wbeta, _p(ws), ws.numel() * 4, _stream()))
                W.grad_done()
            c0 += N
        return (None, None) + (None,) * len(filters)


def _u8_cnn_partials(frames, filters):
    """z [F B rows (t B + b), sum_k fs_k N_k] with z[., zbase_k + i N_k + n] = x . W_k[i D : (i + 1) D][:, n]: ONE product for the whole CNN
    (every filter's shift slices side by side as the rows of one half-plane image under one device-measured scale; N = 1152 instead of
    six launches of N = 128 / 256 on 256-wide tiles), the shifts are summed by the pooling pass (yt8m_timepool_shiftmax_f32).  Needs
    N_k % 32 == 0 (a slice then is a whole number of the image's 32-row groups)."""
    B, F, D = frames.B, frames.F, frames.D
    M = F * B
    dev = frames.q.device
    lib = _lib.lib()
    Ntz = sum(W.data.shape[0] // D * W.data.shape[1] for W in filters)
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    for W in filters:                                                                            # max |W| over all filters (atomicMax)
        _lib.check(lib.yt8m_h2_absmax(_p(W.data), W.data.shape[0], W.data.shape[1], W.data.shape[1], _p(word), _stream()))
    nbytes = lambda rows, K: lib.yt8m_x3_image_bytes(rows, K) // 3 * 2
    img = torch.empty(sum(W.data.shape[0] // D * nbytes(W.data.shape[1], D) for W in filters), dtype=torch.uint8, device=dev)
    cs = torch.empty((Ntz,), dtype=torch.float32, device=dev)
    off, c = 0, 0
    for W in filters:
        fs, N = W.data.shape[0] // D, W.data.shape[1]
        for i in range(fs):
            Wi = W.data[i * D:(i + 1) * D]
            _lib.check(lib.yt8m_h2_split(_p(Wi), D, N, N, U8_ALPHA, _p(word), None, ctypes.c_void_p(img.data_ptr() + off), None, _stream()))
            ops.colsum(Wi, cs[c:c + N])
            off += nbytes(N, D)
            c += N
    z = torch.empty((M, Ntz), dtype=torch.float32, device=dev)
    ws = ops._workspace(dev)
    _lib.check(lib.yt8m_gemm_h1x2_nt_ex(M, Ntz, D, _p(frames.img), 0, _p(img), 0, _p(z), Ntz, None, 1.0, _p(word), _p(frames.r), _p(cs), U8_BETA,
                                        0.0, _p(ws), ws.numel() * 4, _stream()))
    return z


class _PooledCnnU8(torch.autograd.Function):
    """tf.reduce_max(cnn_output, axis=1) of cnn_deep_combine_chain_model.py:100-106 on the raw frames: the dense products of
    _u8_cnn_dense, the pooling in time-major order with the argmax kept (yt8m_timepool_max_f32), and a backward that uses what the
    pooling did to the gradient -- it is non-zero at ONE frame per (video, column), so each filter slice's gradient is B gathered frame
    rows per column (yt8m_u8_cnn_pool_dw) instead of a [D, F B] x [F B, N] product."""

    @staticmethod
    def forward(ctx, token, frames, *filters):
        B, F, D = frames.B, frames.F, frames.D
        dev = frames.q.device
        Ntot = sum(W.data.shape[1] for W in filters)
        out = torch.empty((B, Ntot), dtype=torch.float32, device=dev)
        idx = torch.empty((B, Ntot), dtype=torch.int32, device=dev)
        if all(W.data.shape[1] % 32 == 0 for W in filters) and len(filters) <= 8:
            z = _u8_cnn_partials(frames, filters)                     # one product; the pooling pass sums the shifts
            fs = (ctypes.c_int32 * len(filters))(*[W.data.shape[0] // D for W in filters])
            nc = (ctypes.c_int32 * len(filters))(*[W.data.shape[1] for W in filters])
            _lib.check(_lib.lib().yt8m_timepool_shiftmax_f32(_p(z), F, B, z.shape[1], len(filters), fs, nc, _p(out), _p(idx), Ntot, _stream()))
        else:
            y = _u8_cnn_dense(frames, filters)                        # one product per (filter, shift), accumulated in place
            _lib.check(_lib.lib().yt8m_timepool_max_f32(_p(y), F, B, Ntot, Ntot, _p(out), _p(idx), Ntot, _stream()))
        ctx.frames, ctx.filters, ctx.idx = frames, filters, idx
        return out

    @staticmethod
    def backward(ctx, g):
        frames, filters, idx = ctx.frames, ctx.filters, ctx.idx
        ctx.idx = None
        B, F, D = frames.B, frames.F, frames.D
        g = _f32c(g)
        Ntot = g.shape[1]
        lib = _lib.lib()
        c0 = 0
        for W in filters:
            fs, N = W.data.shape[0] // D, W.data.shape[1]
            if W.trainable and W.grad is not None:
                wbeta = float(W.grad_beta())
                _lib.check(lib.yt8m_u8_cnn_pool_dw(_p(frames.q), _p(frames.r), ctypes.c_void_p(idx.data_ptr() + c0 * 4),
                                                   ctypes.c_void_p(g.data_ptr() + c0 * 4), Ntot, B, F, D, N, fs, _p(W.grad), wbeta,
                                                   _stream()))
                W.grad_done()
            c0 += N
        return (None, None) + (None,) * len(filters)


def u8_cnn(frames, filters):
    """frames: U8FrameImages; filters: the cnn-filter Variables [fs_k D, N_k] in output-column order -> cnn_output [B,F,sum N_k]."""
    return _CnnU8.apply(_token(filters[0]._graph), frames, *filters)


def u8_cnn_maxpool(frames, filters):
    """... -> reduce_max over the frames of that output, [B, sum N_k] (needs sum N_k % 4 == 0: else pool u8_cnn's output)."""
    return _PooledCnnU8.apply(_token(filters[0]._graph), frames, *filters)


def vlad_q_supported(D):
    """The intra-normalisation can hand out q = ||vlad[b,k,:]||^2 (csrc/netvlad.hip, register-resident kernels)."""
    return bool(_lib.lib().yt8m_vlad_finish_q_supported(int(D)))


def _finish_fwd(agg, a, n, centres, eps, want_q):
    B, K, D = agg.shape
    vlad = torch.empty_like(agg)
    q = torch.empty((B, K), dtype=torch.float32, device=agg.device) if want_q else None
    F = a.shape[1] if a is not None else 0
    if want_q:
        _lib.check(_lib.lib().yt8m_vlad_finish_q_fwd(_p(agg), _p(a), _p(centres.data), _p(vlad), _p(n), _p(q), B, F, K, D, eps, _stream()))
    else:
        _lib.check(_lib.lib().yt8m_vlad_finish_fwd(_p(agg), _p(a), _p(centres.data), _p(vlad), _p(n), B, F, K, D, eps, _stream()))
    return vlad, q


def _finish_bwd(agg, n, c, dvlad, dq, eps):
    B, K, D = agg.shape
    dagg = torch.empty_like(agg)
    dn = torch.empty((B, K), dtype=torch.float32, device=agg.device)
    dc = c.grad if c.grad is not None else None
    beta = c.grad_beta() if dc is not None else 0.0
    if dq is not None:
        _lib.check(_lib.lib().yt8m_vlad_finish_q_bwd(_p(agg), _p(n), _p(c.data), _p(dvlad), _p(dq), _p(dagg), _p(dn), _p(dc), beta, B, K,
                                                     D, eps, _stream()))
    else:
        _lib.check(_lib.lib().yt8m_vlad_finish_bwd(_p(agg), _p(n), _p(c.data), _p(dvlad), _p(dagg), _p(dn), _p(dc), beta, B, K, D, eps,
                                                   _stream()))
    if dc is not None:
        c.grad_done()
    return dagg, dn


class _VladFinish(torch.autograd.Function):
    """vlad[b,k,:] = l2norm_D(agg[b,k,:] - (sum_f a[b,f,k]) * c[k,:])   (residual aggregation +
    intra-normalisation in one pass, yt8m_vlad_finish_fwd/bwd); c is a Variable.  want_q: also q [B,K] = ||vlad[b,k,:]||^2
    (differentiable: its gradient reaches the clamped rows only), for the caller's descriptor-wide l2-normalisation."""

    @staticmethod
    def forward(ctx, agg, a, token, centres, eps, want_q):
        agg, a = _f32c(agg), _f32c(a)
        _dev(agg, a)
        B, K, D = agg.shape
        n = torch.empty((B, K), dtype=torch.float32, device=agg.device)
        vlad, q = _finish_fwd(agg, a, n, centres, eps, want_q)
        ctx.save_for_backward(agg, n)
        ctx.centres, ctx.F, ctx.eps, ctx.want_q = centres, a.shape[1], eps, want_q
        return (vlad, q) if want_q else vlad

    @staticmethod
    def backward(ctx, dvlad, dq=None):
        agg, n = ctx.saved_tensors
        if dvlad is None:
            dvlad = torch.zeros_like(agg)
        dagg, dn = _finish_bwd(agg, n, ctx.centres, _f32c(dvlad), None if dq is None else _f32c(dq), ctx.eps)
        da = dn.unsqueeze(1).expand(-1, ctx.F, -1) if ctx.needs_input_grad[1] else None   # broadcast view, no copy
        return dagg, da, None, None, None, None


def vlad_finish(agg, a, centres, eps=1e-12, want_q=False):
    return _VladFinish.apply(agg, a, _token(centres._graph), centres, eps, bool(want_q))


# ---- fused NetVLAD pooling on raw uint8 frames (csrc/netvlad_fused.hip) ---------------------------------------------
_NV_WS = {}


def _netvlad_workspace(B, F, D, K, device):
    need = _lib.lib().yt8m_netvlad_workspace_bytes(B, F, D, K)
    ws = _NV_WS.get(device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _NV_WS[device] = ws
    return ws


def netvlad_fused_supported(q, K):
    return (q.dtype == torch.uint8 and q.dim() == 3 and q.is_cuda
            and bool(_lib.lib().yt8m_netvlad_supported(q.shape[0], q.shape[1], q.shape[2], K)))


def netvlad_fwd_u8(q, num_frames, Wc, bc, nsplit=2, eps=1e-12):
    """(cT [B,K,Fp], n [B,K], agg [B,K,D]) from raw uint8 frames (yt8m_netvlad_fwd_u8).  cT[b,k,f] = a[b,f,k] / ||deq(q[b,f])||
    with the frames contiguous and zero padded to Fp = 32*ceil(F/32): what the aggregation and the backward consume."""
    _dev(q, Wc, bc)
    q = q.contiguous()
    B, F, D = q.shape
    K = Wc.shape[1]
    nf = _nf(num_frames)
    Fp = (F + 31) // 32 * 32
    cT = torch.empty((B, K, Fp), dtype=torch.float32, device=q.device)
    n = torch.empty((B, K), dtype=torch.float32, device=q.device)
    agg = torch.empty((B, K, D), dtype=torch.float32, device=q.device)
    ws = _netvlad_workspace(B, F, D, K, q.device)
    _lib.check(_lib.lib().yt8m_netvlad_fwd_u8(_p(q), _p(nf), _p(_f32c(Wc)), _p(_f32c(bc)), B, F, D, K, int(nsplit), eps, _p(cT),
                                              _p(n), _p(agg), _p(ws), ws.numel(), _stream()))
    return cT, n, agg


def netvlad_bwd_u8(q, num_frames, cT, dagg, dn, dWc, dWc_beta, dbc, dbc_beta, nsplit=2, eps=1e-12):
    _dev(q, cT, dagg, dn, dWc, dbc)
    B, F, D = q.shape
    K = cT.shape[1]
    nf = _nf(num_frames)
    ws = _netvlad_workspace(B, F, D, K, q.device)
    _lib.check(_lib.lib().yt8m_netvlad_bwd_u8(_p(q), _p(nf), _p(cT), _p(_f32c(dagg)), _p(_f32c(dn)), B, F, D, K, int(nsplit), eps,
                                              _p(dWc), float(dWc_beta), _p(dbc), float(dbc_beta), _p(ws), ws.numel(), _stream()))


class _NetVladPoolU8(torch.autograd.Function):
    """uint8 frames -> intra-normalised VLAD descriptor [B,K,D], the dequantise + l2-normalise of
    the input pipeline folded into the two GEMMs; backward writes dW_c, db_c, dcentres into the gradient arena.  want_q: as
    _VladFinish."""

    @staticmethod
    def forward(ctx, q, num_frames, token, Wc, bc, centres, nsplit, eps, want_q):
        q = q.contiguous()
        cT, n, agg = netvlad_fwd_u8(q, num_frames, Wc.data, bc.data, nsplit)
        vlad, qn = _finish_fwd(agg, None, n, centres, eps, want_q)
        ctx.saved = (q, num_frames, cT, agg, n)
        ctx.vars = (Wc, bc, centres)
        ctx.cfg = (nsplit, eps)
        return (vlad, qn) if want_q else vlad

    @staticmethod
    def backward(ctx, dvlad, dq=None):
        q, num_frames, cT, agg, n = ctx.saved
        Wc, bc, c = ctx.vars
        nsplit, eps = ctx.cfg
        ctx.saved = None
        if dvlad is None:
            dvlad = torch.zeros_like(agg)
        dagg, dn = _finish_bwd(agg, n, c, _f32c(dvlad), None if dq is None else _f32c(dq), eps)
        if Wc.grad is not None and bc.grad is not None:
            netvlad_bwd_u8(q, num_frames, cT, dagg, dn, Wc.grad, Wc.grad_beta(), bc.grad.view(-1), bc.grad_beta(), nsplit)
            Wc.grad_done()
            bc.grad_done()
        return (None,) * 9


def netvlad_pool_u8(q, num_frames, Wc, bc, centres, nsplit=2, eps=1e-12, want_q=False):
    return _NetVladPoolU8.apply(q, num_frames, _token(Wc._graph), Wc, bc, centres, int(nsplit), eps, bool(want_q))
