"""Frame-axis autograd ops over the C ABI: BasicLSTM layer (whole recurrence in one call), attention weights,
NetVLAD assignment / aggregation, batched pooling GEMMs.  See include/yt8m_hip.h for the kernel contracts."""
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
            ops.gemm(x_tm.view(F * B, Din), dz2, out=W.grad[:Din], transA=True, beta=beta)
            ops.gemm(hs[:F].view(F * B, H), dz2, out=W.grad[Din:], transA=True, beta=beta)
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


class _PoolTN(torch.autograd.Function):
    """C[b] = w[b]^T . x[b]   (w [B,F,A], x [B,F,H] -> [B,A,H]): attention pooling
    (lstm_attention_max_pooling_model.py:63) and NetVLAD aggregation (Appendix B) as one batched GEMM."""

    @staticmethod
    def forward(ctx, w, x):
        w, x = _f32c(w), _f32c(x)
        ctx.save_for_backward(w, x)
        return ops.gemm_batched(w, x, transA=True)

    @staticmethod
    def backward(ctx, dC):
        w, x = ctx.saved_tensors
        dC = _f32c(dC)
        dw = ops.gemm_batched(x, dC, transB=True) if ctx.needs_input_grad[0] else None   # [F,H].[H,A]
        dx = ops.gemm_batched(w, dC) if ctx.needs_input_grad[1] else None                # [F,A].[A,H]
        return dw, dx


def pool_tn(w, x):
    return _PoolTN.apply(w, x)


class _VladFinish(torch.autograd.Function):
    """vlad[b,k,:] = l2norm_D(agg[b,k,:] - (sum_f a[b,f,k]) * c[k,:])   (residual aggregation +
    intra-normalisation in one pass, yt8m_vlad_finish_fwd/bwd); c is a Variable."""

    @staticmethod
    def forward(ctx, agg, a, token, centres, eps):
        agg, a = _f32c(agg), _f32c(a)
        _dev(agg, a)
        B, K, D = agg.shape
        F = a.shape[1]
        vlad = torch.empty_like(agg)
        n = torch.empty((B, K), dtype=torch.float32, device=agg.device)
        _lib.check(_lib.lib().yt8m_vlad_finish_fwd(_p(agg), _p(a), _p(centres.data), _p(vlad), _p(n), B, F, K, D, eps, _stream()))
        ctx.save_for_backward(agg, n)
        ctx.centres, ctx.F, ctx.eps = centres, F, eps
        return vlad

    @staticmethod
    def backward(ctx, dvlad):
        agg, n = ctx.saved_tensors
        c = ctx.centres
        dvlad = _f32c(dvlad)
        B, K, D = agg.shape
        dagg = torch.empty_like(agg)
        dn = torch.empty((B, K), dtype=torch.float32, device=agg.device)
        dc = c.grad if c.grad is not None else None
        beta = c.grad_beta() if dc is not None else 0.0
        _lib.check(_lib.lib().yt8m_vlad_finish_bwd(_p(agg), _p(n), _p(c.data), _p(dvlad), _p(dagg), _p(dn), _p(dc), beta, B, K, D,
                                                   ctx.eps, _stream()))
        if dc is not None:
            c.grad_done()
        da = dn.unsqueeze(1).expand(-1, ctx.F, -1) if ctx.needs_input_grad[1] else None   # broadcast view, no copy
        return dagg, da, None, None, None


def vlad_finish(agg, a, centres, eps=1e-12):
    return _VladFinish.apply(agg, a, _token(centres._graph), centres, eps)


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
    the input pipeline folded into the two GEMMs; backward writes dW_c, db_c, dcentres into the gradient arena."""

    @staticmethod
    def forward(ctx, q, num_frames, token, Wc, bc, centres, nsplit, eps):
        q = q.contiguous()
        cT, n, agg = netvlad_fwd_u8(q, num_frames, Wc.data, bc.data, nsplit)
        B, K, D = agg.shape
        vlad = torch.empty_like(agg)
        _lib.check(_lib.lib().yt8m_vlad_finish_fwd(_p(agg), None, _p(centres.data), _p(vlad), _p(n), B, q.shape[1], K, D, eps,
                                                   _stream()))
        ctx.saved = (q, num_frames, cT, agg, n)
        ctx.vars = (Wc, bc, centres)
        ctx.cfg = (nsplit, eps)
        return vlad

    @staticmethod
    def backward(ctx, dvlad):
        q, num_frames, cT, agg, n = ctx.saved
        Wc, bc, c = ctx.vars
        nsplit, eps = ctx.cfg
        ctx.saved = None
        dvlad = _f32c(dvlad)
        B, K, D = agg.shape
        dagg = torch.empty_like(agg)
        dn = torch.empty((B, K), dtype=torch.float32, device=agg.device)
        dc = c.grad if c.grad is not None else None
        beta = c.grad_beta() if dc is not None else 0.0
        _lib.check(_lib.lib().yt8m_vlad_finish_bwd(_p(agg), _p(n), _p(c.data), _p(dvlad), _p(dagg), _p(dn), _p(dc), beta, B, K, D,
                                                   eps, _stream()))
        if dc is not None:
            c.grad_done()
        if Wc.grad is not None and bc.grad is not None:
            netvlad_bwd_u8(q, num_frames, cT, dagg, dn, Wc.grad, Wc.grad_beta(), bc.grad.view(-1), bc.grad_beta(), nsplit)
            Wc.grad_done()
            bc.grad_done()
        return (None,) * 8


def netvlad_pool_u8(q, num_frames, Wc, bc, centres, nsplit=2, eps=1e-12):
    return _NetVladPoolU8.apply(q, num_frames, _token(Wc._graph), Wc, bc, centres, int(nsplit), eps)
