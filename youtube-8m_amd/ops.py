"""Thin Python wrappers + autograd plumbing over the C ABI (include/yt8m_hip.h).

torch is used for device memory, streams and the autograd tape only; every FLOP of the hot path is a
HIP kernel in libyt8m_hip.so.  No op has a CPU / eager fallback: host tensors raise.

Gradient routing: parameters are NOT autograd leaves.  Each op's backward writes dW directly into the
variable's slice of the gradient arena (GEMM beta = 0 on first write, 1 afterwards) and returns None for
it.  A dummy requires-grad scalar (``graph.token``) is threaded through every op so that backward runs
even when the op's data input is plain data (first layer).
"""
import ctypes
import os

import torch

from . import _lib
from . import wimg as _wimg
from .flags import FLAGS, DEFINE_string
from .variables import get_default_graph

# new: compute dtype of the large GEMMs (BASELINE config 5 is bf16; configs 1-4 are fp32)
DEFINE_string("compute_dtype", "float32", "float32 (exact fp32 MFMA) or bfloat16 (bf16 MFMA operands, fp32 accumulate, fp32 "
              "master weights / gradients / optimiser) for the fully-connected and MoE-head GEMMs; the fused NetVLAD pooling "
              "uses single-f16 operands instead of the f16 hi+lo split.")

ACT = {"sigmoid": 0, "relu": 1, "relu6": 2, "tanh": 3, "elu": 4}
XENT_EPS = 10e-6  # W/losses.py:115


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Yt8mHipError("yt8m_amd ops run on the MI355X only (got a %s tensor); there is no CPU fallback"
                                    % t.device.type)


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _rowmajor2d(t):
    """Accepts 2-D tensors whose inner stride is 1 (row-major with an arbitrary leading dimension)."""
    if t.dim() != 2:
        raise ValueError("expected a 2-D tensor")
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    if t.stride(1) != 1 and t.shape[1] != 1:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    if ld < t.shape[1]:
        t = t.contiguous()
        ld = t.shape[1]
    return t, ld


# ------------------------------------------------------------------------------------------ raw kernels
_WS = {}


def _workspace(device):
    """Split-K workspace of the persistent GEMM: one per (device, stream) -- 256 MiB each of the 288 GB -- so that GEMMs
    running concurrently on different streams (the pipelined LSTM stack) never share scratch."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(_lib.lib().yt8m_gemm_workspace_bytes() // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def _problem(A, B, out, transA, transB, bias, beta):
    A, lda = _rowmajor2d(A)
    B, ldb = _rowmajor2d(B)
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if transB else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError("gemm: inner dimensions differ (%d vs %d)" % (K, K2))
    if out is None:
        if beta != 0.0:
            raise ValueError("beta != 0 needs an output tensor")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if out.dtype != torch.float32 or out.dim() != 2 or tuple(out.shape) != (M, N) or (out.stride(1) != 1 and N != 1):
        raise ValueError("gemm: bad output tensor")
    ldc = out.stride(0) if M > 1 else max(N, 1)
    if bias is not None:
        bias = _f32c(bias)
        if bias.numel() != N:
            raise ValueError("bias size mismatch")
    pr = _lib.GemmProblem(M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc,
                          bias.data_ptr() if bias is not None else None, float(beta))
    return pr, out, (A, B, bias)


X3 = os.environ.get("YT8M_GEMM_X3", "1") != "0"        # large fp32 products on the bf16 pipe (three-plane split, csrc/gemm_x3.hip)


def _x3_wins(shapes, transA=False, transB=False):
    """The library's cost estimate (csrc/gemm_auto.hip: six bf16 products of split operands on 256 x 256 tiles plus the split passes
    against the fp32-MFMA kernel on 128 x 128 tiles) for every (M, N, K) of a group; the dispatch itself is per product."""
    shapes = [s_ for s_ in shapes if s_[0] * s_[1] * s_[2] != 0]
    return bool(shapes) and all(_lib.lib().yt8m_gemm_x3_pays(int(M), int(N), int(K)) for M, N, K in shapes)


GEMM_ROLE_DW = 0x100          # include/yt8m_hip.h YT8M_GEMM_ROLE_DW
GEMM_ROLE_H2 = 0x200          # include/yt8m_hip.h YT8M_GEMM_ROLE_H2


def gemm_grouped(items, transA=False, transB=False, role=None):
    """items: list of dicts(A=, B=, out=None, bias=None, beta=0.0) sharing transA/transB -> list of outputs.
    role="dw" DECLARES the products weight gradients x^T . dz (transA, neither operand a weight): the library may then run them as
    three f16 products under one device-measured scale per operand (the "h2" contract of include/yt8m_hip.h); without it a
    product keeps the six-product / fp32 forms whatever its transposition flags.
    fp32 operands either way; the library picks the kernel per product (yt8m_gemm_auto_grouped: large products run as six bf16
    MFMA products of three-plane split operands -- fp32-grade error at twice the rate --, the rest on the fp32 MFMA kernel) and
    groups the launches; the operand images live in a scratch tensor sized by the library's own query."""
    probs, outs, keep = [], [], []
    for it in items:
        _dev(it["A"], it["B"], it.get("out"), it.get("bias"))
        pr, out, k = _problem(it["A"], it["B"], it.get("out"), transA, transB, it.get("bias"), it.get("beta", 0.0))
        probs.append(pr)
        outs.append(out)
        keep.append(k)
    if role not in (None, "dw", "h2"):
        raise ValueError("role must be None, 'dw' or 'h2'")
    ta = int(bool(transA)) | (GEMM_ROLE_DW if (role == "dw" and transA and not transB) else 0) | (GEMM_ROLE_H2 if role == "h2" else 0)
    ws = _workspace(outs[0].device)
    lib = _lib.lib()
    for lo in range(0, len(probs), 64):
        part = probs[lo:lo + 64]
        arr = (_lib.GemmProblem * len(part))(*part)
        if not X3:
            for i in range(0, len(part), 4):
                sub = (_lib.GemmProblem * len(part[i:i + 4]))(*part[i:i + 4])
                _lib.check(lib.yt8m_gemm_f32_grouped(int(transA), int(transB), len(part[i:i + 4]), sub, _p(ws), ws.numel() * 4, _stream()))
            continue
        nb = lib.yt8m_gemm_auto_scratch_bytes(ta, int(transB), len(part), arr)
        img = torch.empty(nb, dtype=torch.uint8, device=outs[0].device) if nb else None
        words = [(it.get("absmaxA"), it.get("absmaxB")) for it in items[lo:lo + 64]]
        if any(a is not None or b is not None for a, b in words):
            # operands whose maximum the caller already has on the device (a [1] float tensor holding it as float bits: yt8m_h2_absmax's
            # form) skip their absmax pass when they take the h2 form
            pa = (ctypes.c_void_p * len(part))(*[a.data_ptr() if a is not None else None for a, _ in words])
            pb = (ctypes.c_void_p * len(part))(*[b.data_ptr() if b is not None else None for _, b in words])
            _lib.check(lib.yt8m_gemm_auto_grouped_ex(ta, int(transB), len(part), arr, pa, pb, _p(ws), ws.numel() * 4, _p(img), nb, None, _stream()))
        else:
            _lib.check(lib.yt8m_gemm_auto_grouped(ta, int(transB), len(part), arr, _p(ws), ws.numel() * 4, _p(img), nb, None, _stream()))
    return outs


def gemm(A, B, out=None, transA=False, transB=False, bias=None, beta=0.0, role=None):
    """out[M,N] = op(A) . op(B) (+ bias) (+ out if beta == 1), through the persistent scheduler (role: see gemm_grouped)."""
    return gemm_grouped([dict(A=A, B=B, out=out, bias=bias, beta=beta)], transA, transB, role=role)[0]


def gemm_simple(A, B, out=None, transA=False, transB=False, bias=None, beta=0.0):
    """Same contract through the plain one-tile-per-workgroup launch (yt8m_gemm_f32)."""
    _dev(A, B, out, bias)
    pr, out, keep = _problem(A, B, out, transA, transB, bias, beta)
    _lib.check(_lib.lib().yt8m_gemm_f32(int(transA), int(transB), pr.M, pr.N, pr.K, pr.A, pr.lda, pr.B, pr.ldb, pr.C, pr.ldc,
                                        pr.bias, float(beta), _stream()))
    return out


def _bf16_empty(rows, cols, device):
    """bf16 [rows, cols] whose rows start 16-byte aligned (row pitch padded to a multiple of 8 elements): the bf16 GEMMs then
    stay on their LDS-DMA path for any reduction length (V*(M+1) = 14148 is not a multiple of 8)."""
    pitch = (cols + 7) // 8 * 8
    return torch.empty((rows, pitch), dtype=torch.bfloat16, device=device)[:, :cols]


def cast_bf16(x, transpose=False):
    """fp32 [R,C] (inner stride 1) -> bf16 [R,C] or, transposed, [C,R] (yt8m_cast_f32_bf16, round to nearest even); the result
    is a view with a 16-byte aligned row pitch."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    R, C = x.shape
    out = _bf16_empty(C, R, x.device) if transpose else _bf16_empty(R, C, x.device)
    _lib.check(_lib.lib().yt8m_cast_f32_bf16(_p(x), R, C, ld, _p(out), out.stride(0), int(transpose), _stream()))
    return out


def gemm_any(A, B, out=None, transA=False, transB=False, bias=None, beta=0.0, bf16=False, role=None):
    """C = op(A) . op(B) (+ bias) (+ C); bf16=True takes bf16 copies of both operands (each cast into the K-contiguous layout the
    NT kernel wants) when the product is large enough to pay for the two cast passes, fp32 accumulate / output either way."""
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    if not (bf16 and _use_bf16(True, M, N, K, weight_operand=False) and K >= 32):
        return gemm(A, B, out=out, transA=transA, transB=transB, bias=bias, beta=beta, role=role)
    Ab = cast_bf16(A, transpose=transA)                # op(A)   [M, K]
    Bb = cast_bf16(B, transpose=not transB)            # op(B)^T [N, K]
    return gemm_bf16_nt_grouped([dict(A=Ab, B=Bb, out=out, bias=bias, beta=beta)])[0]


def cast_bf16_both(x):
    """fp32 [R,C] -> (bf16 [R,C], bf16 [C,R]) from one pass over the source (yt8m_cast_f32_bf16_dual)."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    R, C = x.shape
    plain = _bf16_empty(R, C, x.device)
    trans = _bf16_empty(C, R, x.device)
    _lib.check(_lib.lib().yt8m_cast_f32_bf16_dual(_p(x), R, C, ld, _p(plain), plain.stride(0), _p(trans), trans.stride(0), _stream()))
    return plain, trans


def gemm_bf16_nt_grouped(items):
    """items: dicts(A=bf16 [M,K], B=bf16 [N,K], out=None fp32 [M,N], bias=None, beta=0.0) -> fp32 outputs.
    C = A . B^T on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (yt8m_gemm_bf16_nt_grouped)."""
    probs, outs, keep = [], [], []
    for it in items:
        A, B = it["A"], it["B"]
        _dev(A, B, it.get("out"), it.get("bias"))
        if A.dtype != torch.bfloat16 or B.dtype != torch.bfloat16 or A.dim() != 2 or B.dim() != 2:
            raise TypeError("gemm_bf16_nt: operands must be 2-D bfloat16")
        A = A if A.stride(1) == 1 else A.contiguous()
        B = B if B.stride(1) == 1 else B.contiguous()
        M, K = A.shape
        N, K2 = B.shape
        if K != K2:
            raise ValueError("gemm_bf16_nt: inner dimensions differ (%d vs %d)" % (K, K2))
        out = it.get("out")
        beta = it.get("beta", 0.0)
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs an output tensor")
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or (out.stride(1) != 1 and N != 1):
            raise ValueError("gemm_bf16_nt: bad output tensor")
        bias = it.get("bias")
        if bias is not None:
            bias = _f32c(bias)
        lda = A.stride(0) if M > 1 else max(K, 1)
        ldb = B.stride(0) if N > 1 else max(K, 1)
        ldc = out.stride(0) if M > 1 else max(N, 1)
        probs.append(_lib.GemmProblem(M, N, K, A.data_ptr(), lda, B.data_ptr(), ldb, out.data_ptr(), ldc,
                                      bias.data_ptr() if bias is not None else None, float(beta)))
        outs.append(out)
        keep.append((A, B, bias))
    arr = (_lib.GemmProblem * len(probs))(*probs)
    ws = _workspace(outs[0].device)
    _lib.check(_lib.lib().yt8m_gemm_bf16_nt_grouped(len(probs), arr, _p(ws), ws.numel() * 4, _stream()))
    return outs


class X3Image:
    """Three-plane bf16 split of an fp32 matrix (csrc/gemm_x3.hip): `rows` x `K` logical shape, [row][ceil(K/16)][3][16] bf16."""
    __slots__ = ("buf", "rows", "K")

    def __init__(self, buf, rows, K):
        self.buf, self.rows, self.K = buf, rows, K


def _x3_empty(rows, K, device):
    n = _lib.lib().yt8m_x3_image_bytes(rows, K)
    return X3Image(torch.empty(max(n, 16), dtype=torch.uint8, device=device), rows, K)


def x3_split(x, plain=True, trans=False, scale=1.0):
    """fp32 [R, C] -> (X3Image of x as an [R rows, K = C] operand or None, X3Image of x^T as a [C rows, K = R] operand or None),
    both from one pass over x (yt8m_x3_split)."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    R, C = x.shape
    ip = it = None
    if plain:                                                # a weight matrix whose image the optimiser pass keeps current (wimg.py)
        r = _wimg.resident_image(x, R, C, ld, 0, 3, scale)
        ip = X3Image(r[0], R, C) if r else None
    if trans:
        r = _wimg.resident_image(x, R, C, ld, 1, 3, scale)
        it = X3Image(r[0], C, R) if r else None
    mk_p, mk_t = plain and ip is None, trans and it is None
    if mk_p:
        ip = _x3_empty(R, C, x.device)
    if mk_t:
        it = _x3_empty(C, R, x.device)
    if mk_p or mk_t:
        _lib.check(_lib.lib().yt8m_x3_split(_p(x), R, C, ld, float(scale), _p(ip.buf) if mk_p else None, _p(it.buf) if mk_t else None,
                                            _stream()))
    return ip, it


def gemm_x3_grouped(items):
    """items: dicts(A=X3Image [M rows, K], B=X3Image [N rows, K], out=None fp32 [M,N], bias=None, beta=0.0) -> fp32 outputs
    C = A . B^T from six bf16 MFMA products of the split operands (yt8m_gemm_x3_nt_grouped): fp32-grade accuracy."""
    probs, outs, keep = [], [], []
    for it in items:
        A, B = it["A"], it["B"]
        if not isinstance(A, X3Image) or not isinstance(B, X3Image):
            raise TypeError("gemm_x3: operands must be X3Image")
        if A.K != B.K:
            raise ValueError("gemm_x3: inner dimensions differ (%d vs %d)" % (A.K, B.K))
        M, N, K = A.rows, B.rows, A.K
        out = it.get("out")
        beta = it.get("beta", 0.0)
        _dev(A.buf, B.buf, out, it.get("bias"))
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs an output tensor")
            out = torch.empty((M, N), dtype=torch.float32, device=A.buf.device)
        if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or (out.stride(1) != 1 and N != 1):
            raise ValueError("gemm_x3: bad output tensor")
        bias = it.get("bias")
        if bias is not None:
            bias = _f32c(bias)
            if bias.numel() != N:
                raise ValueError("bias size mismatch")
        ldc = out.stride(0) if M > 1 else max(N, 1)
        probs.append(_lib.GemmProblem(M, N, K, A.buf.data_ptr(), 0, B.buf.data_ptr(), 0, out.data_ptr(), ldc,
                                      bias.data_ptr() if bias is not None else None, float(beta)))
        outs.append(out)
        keep.append((A, B, bias))
    ws = _workspace(outs[0].device)
    for i in range(0, len(probs), 4):                               # the library takes 1..4 problems per launch
        part = probs[i:i + 4]
        arr = (_lib.GemmProblem * len(part))(*part)
        _lib.check(_lib.lib().yt8m_gemm_x3_nt_grouped(len(part), arr, _p(ws), ws.numel() * 4, _stream()))
    return outs


# ---- h2: fp32 products as three f16 MFMA products of two-plane half images (csrc/gemm_x3.hip gemm_h2q_kernel, round 5) -------------
class H2Image(X3Image):
    """Two IEEE-half planes of S x (csrc/x3_image.h); `scale` = the host-side S (the caller's alpha carries 1 / S), `dinv` = the device
    word with max |x| (yt8m_h2_absmax) when the rest of the scale was chosen on the device, else None."""
    __slots__ = ("scale", "dinv")

    def __init__(self, buf, rows, K, scale=1.0, dinv=None):
        X3Image.__init__(self, buf, rows, K)
        self.scale, self.dinv = float(scale), dinv


def h2_absmax(x):
    """Device int32 word holding max |x| as float bits (yt8m_h2_absmax): the handle of a device-chosen h2 scale."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    word = torch.zeros(1, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().yt8m_h2_absmax(_p(x), x.shape[0], x.shape[1], ld, _p(word), _stream()))
    return word


def h2_split(x, plain=True, trans=False, scale=1.0, dynamic=False):
    """fp32 [R, C] -> (H2Image of x as an [R rows, K = C] operand or None, H2Image of x^T or None).  scale: host power of two with
    |scale x| < 65504; dynamic=True: the scale is measured on the device instead (one more pass over x)."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    R, C = x.shape
    lib = _lib.lib()
    ds = h2_absmax(x) if dynamic else None
    mk = lambda rows, K: torch.empty(max(lib.yt8m_x3_image_bytes(rows, K) // 3 * 2, 16), dtype=torch.uint8, device=x.device)
    bp = mk(R, C) if plain else None
    bt = mk(C, R) if trans else None
    _lib.check(lib.yt8m_h2_split(_p(x), R, C, ld, float(scale), _p(ds), _p(bp), _p(bt), None, _stream()))
    return (H2Image(bp, R, C, scale, ds) if plain else None, H2Image(bt, C, R, scale, ds) if trans else None)


def gemm_h2_grouped(items):
    """items: dicts(A=H2Image [M rows, K], B=H2Image [N rows, K], out=None fp32 [M,N], bias=None, beta=0.0) -> fp32 outputs
    C = A . B^T from three f16 MFMA products of the two-plane operands (yt8m_gemm_h2_nt_grouped); the images' scales are undone in
    the epilogue (host part as alpha, device part through pointers)."""
    probs, outs, keep, alphas, dsa, dsb = [], [], [], [], [], []
    for it in items:
        A, B = it["A"], it["B"]
        if not isinstance(A, H2Image) or not isinstance(B, H2Image):
            raise TypeError("gemm_h2: operands must be H2Image")
        if A.K != B.K:
            raise ValueError("gemm_h2: inner dimensions differ (%d vs %d)" % (A.K, B.K))
        M, N, K = A.rows, B.rows, A.K
        out, beta, bias = it.get("out"), it.get("beta", 0.0), it.get("bias")
        _dev(A.buf, B.buf, out, bias)
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs an output tensor")
            out = torch.empty((M, N), dtype=torch.float32, device=A.buf.device)
        if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or (out.stride(1) != 1 and N != 1):
            raise ValueError("gemm_h2: bad output tensor")
        if bias is not None:
            bias = _f32c(bias)
        ldc = out.stride(0) if M > 1 else max(N, 1)
        probs.append(_lib.GemmProblem(M, N, K, A.buf.data_ptr(), 0, B.buf.data_ptr(), 0, out.data_ptr(), ldc,
                                      bias.data_ptr() if bias is not None else None, float(beta)))
        alphas.append(1.0 / (A.scale * B.scale))
        dsa.append(A.dinv.data_ptr() if A.dinv is not None else None)
        dsb.append(B.dinv.data_ptr() if B.dinv is not None else None)
        outs.append(out)
        keep.append((A, B, bias))
    ws = _workspace(outs[0].device)
    for i in range(0, len(probs), 4):
        n = len(probs[i:i + 4])
        arr = (_lib.GemmProblem * n)(*probs[i:i + 4])
        al = (ctypes.c_float * n)(*alphas[i:i + 4])
        pa = (ctypes.c_void_p * n)(*dsa[i:i + 4])
        pb = (ctypes.c_void_p * n)(*dsb[i:i + 4])
        _lib.check(_lib.lib().yt8m_gemm_h2_nt_grouped(n, arr, al, pa, pb, _p(ws), ws.numel() * 4, _stream()))
    return outs


B1_IMAGES = os.environ.get("YT8M_BF16_IMAGES", "1") != "0"   # bf16 configuration: large products on one-plane operand images (gemm_b1_kernel)


def _b1_ok(M, N, K):
    """Large enough for the 256 x 256 image kernel to pay (at least half a round of tiles, a few K steps)."""
    return B1_IMAGES and ((M + 255) // 256) * ((N + 255) // 256) >= 64 and K >= 256


def bf16_image(x, transpose=False, both=False):
    """fp32 [R, C] -> one-plane bf16 operand image(s) of csrc/gemm_x3.hip (yt8m_bf16_image): the matrix as an [R rows, K = C]
    operand, its transpose as a [C rows, K = R] operand (transpose=True), or both from one pass (both=True -> (plain, trans))."""
    _dev(x)
    x, ld = _rowmajor2d(x)
    R, C = x.shape
    lib = _lib.lib()
    mk = lambda rows, K: X3Image(torch.empty(max(lib.yt8m_x3_image_bytes(rows, K) // 3, 16), dtype=torch.uint8, device=x.device), rows, K)
    want_p, want_t = both or not transpose, both or transpose
    ip = it = None
    if want_p:                                               # resident one-plane image of a weight matrix (wimg.py)
        r = _wimg.resident_image(x, R, C, ld, 0, 1)
        ip = X3Image(r[0], R, C) if r else None
    if want_t:
        r = _wimg.resident_image(x, R, C, ld, 1, 1)
        it = X3Image(r[0], C, R) if r else None
    mk_p, mk_t = want_p and ip is None, want_t and it is None
    if mk_p:
        ip = mk(R, C)
    if mk_t:
        it = mk(C, R)
    if mk_p or mk_t:
        _lib.check(lib.yt8m_bf16_image(_p(x), R, C, ld, 1.0, _p(ip.buf) if mk_p else None, _p(it.buf) if mk_t else None, _stream()))
    return (ip, it) if both else (it if transpose else ip)


def gemm_b1_grouped(items):
    """items: dicts(A=image [M rows, K], B=image [N rows, K], out=None fp32 [M,N], bias=None, beta=0.0) -> fp32 outputs;
    C = A . B^T on v_mfma_f32_32x32x16_bf16 from ONE-plane bf16 operand images (yt8m_gemm_b1_nt_grouped)."""
    probs, outs, keep = [], [], []
    for it in items:
        A, B = it["A"], it["B"]
        if A.K != B.K:
            raise ValueError("gemm_b1: inner dimensions differ (%d vs %d)" % (A.K, B.K))
        M, N, K = A.rows, B.rows, A.K
        out, beta, bias = it.get("out"), it.get("beta", 0.0), it.get("bias")
        _dev(A.buf, B.buf, out, bias)
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs an output tensor")
            out = torch.empty((M, N), dtype=it.get("out_dtype", torch.float32), device=A.buf.device)
        if out.dtype not in (torch.float32, torch.bfloat16) or tuple(out.shape) != (M, N) or (out.stride(1) != 1 and N != 1):
            raise ValueError("gemm_b1: bad output tensor")
        if out.dtype == torch.bfloat16 and (beta != 0.0 or out.stride(0) % 4 != 0):
            raise ValueError("gemm_b1: a bf16 output takes beta = 0 and a row pitch that is a multiple of 4")
        if bias is not None:
            bias = _f32c(bias)
        ldc = out.stride(0) if M > 1 else max(N, 1)
        probs.append(_lib.GemmProblem(M, N, K, A.buf.data_ptr(), 0, B.buf.data_ptr(), 0, out.data_ptr(), ldc,
                                      bias.data_ptr() if bias is not None else None, float(beta)))
        outs.append(out)
        keep.append((A, B, bias))
    ws = _workspace(outs[0].device)
    for i in range(0, len(probs), 4):
        part = probs[i:i + 4]
        arr = (_lib.GemmProblem * len(part))(*part)
        mask = sum(1 << j for j, o in enumerate(outs[i:i + 4]) if o.dtype == torch.bfloat16)
        if mask:                                  # bf16 outputs (round 6: the MoE logits of the bf16 configuration)
            _lib.check(_lib.lib().yt8m_gemm_b1_nt_grouped_bf16c(len(part), arr, mask, _p(ws), ws.numel() * 4, _stream()))
        else:
            _lib.check(_lib.lib().yt8m_gemm_b1_nt_grouped(len(part), arr, _p(ws), ws.numel() * 4, _stream()))
    return outs


def gemm_batched(A, B, out=None, transA=False, transB=False, beta=0.0):
    """Batched over dim 0 of 3-D contiguous tensors.  yt8m_gemm_f32_batched."""
    _dev(A, B, out)
    A, B = _f32c(A), _f32c(B)
    nb = A.shape[0]
    M, K = (A.shape[2], A.shape[1]) if transA else (A.shape[1], A.shape[2])
    K2, N = (B.shape[2], B.shape[1]) if transB else (B.shape[1], B.shape[2])
    if K != K2 or B.shape[0] != nb:
        raise ValueError("gemm_batched: shape mismatch")
    if out is None:
        out = torch.empty((nb, M, N), dtype=torch.float32, device=A.device)
    if not out.is_contiguous() or tuple(out.shape) != (nb, M, N):
        raise ValueError("gemm_batched: bad output tensor")
    _lib.check(_lib.lib().yt8m_gemm_f32_batched(int(transA), int(transB), M, N, K, _p(A), A.shape[2], A.shape[1] * A.shape[2],
                                                _p(B), B.shape[2], B.shape[1] * B.shape[2], _p(out), N, M * N,
                                                float(beta), nb, _stream()))
    return out


def colsum(X, out, beta=0.0):
    _dev(X, out)
    X, ld = _rowmajor2d(X)
    ws = _workspace(X.device)       # stream-ordered scratch shared with the GEMM's split-K partials
    _lib.check(_lib.lib().yt8m_colsum_f32(_p(X), X.shape[0], X.shape[1], ld, _p(out), float(beta), _p(ws),
                                          min(ws.numel() * 4, 1 << 22), _stream()))
    return out


def act_fwd(kind, x):
    _dev(x)
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().yt8m_act_fwd_f32(ACT[kind], _p(x), _p(y), x.numel(), _stream()))
    return y


def act_bwd(kind, y, dy):
    _dev(y, dy)
    y, dy = _f32c(y), _f32c(dy)
    dx = torch.empty_like(y)
    _lib.check(_lib.lib().yt8m_act_bwd_f32(ACT[kind], _p(y), _p(dy), _p(dx), y.numel(), _stream()))
    return dx


def l2norm_fwd(x, eps=1e-12):
    _dev(x)
    x = _f32c(x)
    cols = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().yt8m_l2norm_fwd_f32(_p(x), _p(y), x.numel() // max(cols, 1), cols, eps, _stream()))
    return y


def l2norm_bwd(x, dy, eps=1e-12):
    _dev(x, dy)
    x, dy = _f32c(x), _f32c(dy)
    cols = x.shape[-1]
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().yt8m_l2norm_bwd_f32(_p(x), _p(dy), _p(dx), x.numel() // max(cols, 1), cols, eps, _stream()))
    return dx


def dequant_l2norm(q, num_frames=None, eps=1e-12):
    """uint8 [B,F,D] -> float32 [B,F,D]: dequantise, zero the padding rows, L2-normalise each frame."""
    _dev(q, num_frames)
    if q.dtype != torch.uint8 or q.dim() != 3:
        raise TypeError("expected uint8 [B,F,D]")
    q = q.contiguous()
    B, F, D = q.shape
    x = torch.empty((B, F, D), dtype=torch.float32, device=q.device)
    nf = None if num_frames is None else num_frames.to(torch.int32).contiguous()
    _lib.check(_lib.lib().yt8m_dequant_l2norm_u8(_p(q), _p(nf), _p(x), B, F, D, eps, _stream()))
    return x


def dequant_mean_l2norm(q, num_frames=None, eps=1e-12):
    """uint8 [B,F,D] -> float32 [B,D]: mean of the dequantised valid frames, L2-normalised."""
    _dev(q, num_frames)
    if q.dtype != torch.uint8 or q.dim() != 3:
        raise TypeError("expected uint8 [B,F,D]")
    q = q.contiguous()
    B, F, D = q.shape
    x = torch.empty((B, D), dtype=torch.float32, device=q.device)
    nf = None if num_frames is None else num_frames.to(torch.int32).contiguous()
    _lib.check(_lib.lib().yt8m_dequant_mean_l2norm_u8(_p(q), _p(nf), _p(x), B, F, D, eps, _stream()))
    return x


def moe_mix_fwd(Zg, Ze, V, M):
    _dev(Zg, Ze)
    B = Zg.shape[0]
    p = torch.empty((B, V), dtype=torch.float32, device=Zg.device)
    if Zg.dtype == torch.bfloat16:                # logits written as bf16 by the b1 product (Z16_LOGITS)
        _lib.check(_lib.lib().yt8m_moe_mix_fwd_bf16z(_p(Zg), _p(Ze), _p(p), B, V, M, _stream()))
    else:
        _lib.check(_lib.lib().yt8m_moe_mix_fwd(_p(Zg), _p(Ze), _p(p), B, V, M, _stream()))
    return p


def moe_mix_bwd_(Zg, Ze, dp, V, M):
    _dev(Zg, Ze, dp)
    dp = _f32c(dp)
    _lib.check(_lib.lib().yt8m_moe_mix_bwd(_p(Zg), _p(Ze), _p(dp), Zg.shape[0], V, M, _stream()))
    return Zg, Ze


def _labels_arg(labels):
    if labels.dtype == torch.bool:
        labels = labels.view(torch.uint8)
    if labels.dtype == torch.uint8:
        return labels.contiguous(), 0
    if labels.dtype == torch.float32:
        return labels.contiguous(), 1
    raise TypeError("labels must be bool / uint8 / float32, got %s" % labels.dtype)


def xent_fwd(p, labels, weights=None, want_dp=False, eps=XENT_EPS, upstream=1.0):
    _dev(p, labels, weights)
    p = _f32c(p)
    B, V = p.shape
    lab, ldt = _labels_arg(labels)
    if tuple(lab.shape) != (B, V):
        raise ValueError("labels shape %s != predictions shape %s" % (tuple(lab.shape), (B, V)))
    L = _lib.lib()
    ws = torch.empty((L.yt8m_xent_workspace_bytes(B, V) + 3) // 4, dtype=torch.float32, device=p.device)
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    dp = torch.empty_like(p) if want_dp else None
    w = None if weights is None else _f32c(weights)
    _lib.check(L.yt8m_xent_fwd_bwd(_p(p), _p(lab), ldt, _p(w), _p(loss), _p(dp), B, V, eps, upstream, _p(ws), _stream()))
    return loss, dp


def xent_bwd(p, labels, weights, upstream_dev, eps=XENT_EPS, upstream=1.0):
    _dev(p, labels, weights, upstream_dev)
    p = _f32c(p)
    B, V = p.shape
    lab, ldt = _labels_arg(labels)
    dp = torch.empty_like(p)
    w = None if weights is None else _f32c(weights)
    up = None if upstream_dev is None else _f32c(upstream_dev)
    _lib.check(_lib.lib().yt8m_xent_bwd(_p(p), _p(lab), ldt, _p(w), _p(up), _p(dp), B, V, eps, upstream, _stream()))
    return dp


def topk_rows(p, k=20):
    _dev(p)
    p = _f32c(p)
    B, V = p.shape
    k = min(k, V)
    vals = torch.empty((B, k), dtype=torch.float32, device=p.device)
    idx = torch.empty((B, k), dtype=torch.int32, device=p.device)
    _lib.check(_lib.lib().yt8m_topk_rows(_p(p), B, V, k, _p(vals), _p(idx), _stream()))
    return vals, idx


def sample_frames(x, num_frames, num_samples, mode, seed, return_index=False):
    """SampleRandomFrames (mode 0) / SampleRandomSequence (mode 1) of W/model_utils.py:23-70 on the device: x [B,F,D] uint8 or
    float32 -> [B,S,D] of the same dtype (the frames are data: no gradient)."""
    _dev(x)
    x = x.contiguous()
    B, F, D = x.shape
    nf = None if num_frames is None else num_frames.to(torch.int32).contiguous().view(-1)
    out = torch.empty((B, num_samples, D), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, num_samples), dtype=torch.int32, device=x.device) if return_index else None
    fn = {torch.uint8: "yt8m_sample_frames_u8", torch.float32: "yt8m_sample_frames_f32"}[x.dtype]
    _lib.check(getattr(_lib.lib(), fn)(_p(x), _p(nf), B, F, D, int(num_samples), int(mode), int(seed), _p(out), _p(idx), _stream()))
    return (out, idx) if return_index else out


class _FramePool(torch.autograd.Function):
    """FramePooling max / average over the sampled frames (W/model_utils.py:72-95); the max gradient is split equally between
    tied maxima like tf.reduce_max's."""

    @staticmethod
    def forward(ctx, x, mode):
        x = _f32c(x)
        _dev(x)
        B, S, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().yt8m_frame_pool_fwd(_p(x), B, S, C, mode, _p(out), _stream()))
        ctx.save_for_backward(x, out)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, dy):
        x, out = ctx.saved_tensors
        B, S, C = x.shape
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().yt8m_frame_pool_bwd(_p(x), _p(out), _p(dy), B, S, C, ctx.mode, _p(dx), _stream()))
        return dx, None


def frame_pool(x, method):
    return _FramePool.apply(x, {"max": 0, "average": 1}[method])


class _BatchNorm(torch.autograd.Function):
    """slim.batch_norm(center=True, scale=True) (W/all_frame_models/dbof_model.py:66-71,79-84,103-108) on the
    rows of x [N,C]: yt8m_batchnorm_fwd / _bwd.  Couples the examples of the local batch, like the reference."""

    @staticmethod
    def forward(ctx, x, token, gamma, beta, mm, mv, is_training, eps, decay):
        x = _f32c(x)
        _dev(x)
        N, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().yt8m_batchnorm_fwd(_p(x), N, C, _p(gamma.data), _p(beta.data), _p(mm.data), _p(mv.data),
                                                 int(bool(is_training)), float(eps), float(decay), _p(y), _p(mean), _p(rstd), _stream()))
        ctx.save_for_backward(x, mean, rstd)
        ctx.vars = (gamma, beta, bool(is_training))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta, is_training = ctx.vars
        N, C = x.shape
        dy = _f32c(dy)
        L = _lib.lib()
        ws = torch.empty(L.yt8m_batchnorm_workspace_bytes(C) // 4, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gg, gb = gamma.grad, beta.grad
        bg = gamma.grad_beta() if gg is not None else 0.0
        bb = beta.grad_beta() if gb is not None else 0.0
        _lib.check(L.yt8m_batchnorm_bwd(_p(x), _p(dy), N, C, _p(gamma.data), _p(mean), _p(rstd), int(is_training), _p(dx), _p(gg),
                                        float(bg), _p(gb), float(bb), _p(ws), ws.numel() * 4, _stream()))
        if gg is not None:
            gamma.grad_done()
        if gb is not None:
            beta.grad_done()
        return dx, None, None, None, None, None, None, None, None


def batch_norm(x, gamma, beta, moving_mean, moving_variance, is_training, eps=1e-3, decay=0.999):
    return _BatchNorm.apply(x, _token(gamma._graph), gamma, beta, moving_mean, moving_variance, is_training, eps, decay)


def perr_rows(p, labels):
    """Per-video precision at equal recall rate on the device (W/eval_util.py:74-99); labels bool / uint8 [B,V]."""
    _dev(p)
    p = _f32c(p)
    lab = labels.contiguous()
    if lab.dtype == torch.bool:
        lab = lab.view(torch.uint8)
    assert lab.dtype == torch.uint8 and lab.shape == p.shape
    B, V = p.shape
    out = torch.empty((B,), dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().yt8m_perr_rows(_p(p), _p(lab), B, V, _p(out), _stream()))
    return out


def sqnorm_and_adam(graph, lr_t, gscale=1.0, clip=1.0, beta1=0.9, beta2=0.999, eps=1e-8, tensors=None):
    """Per-tensor clip + TF-Adam over the arena (two multi-tensor passes).  tensors = (lo, hi) restricts the update
    to trainable variables lo..hi-1 (a contiguous slice of the chunk table)."""
    _dev(graph.params)
    L = _lib.lib()
    s = _stream()
    nt = len(graph.trainable_variables())
    lo, hi = (0, nt) if tensors is None else tensors
    if hi <= lo:
        return
    c0, c1 = graph.chunk_start[lo], graph.chunk_start[hi]
    chunks = ctypes.c_void_p(graph.chunks.data_ptr() + 16 * c0)
    partial = ctypes.c_void_p(graph.partial.data_ptr() + 4 * c0)
    if clip > 0:
        _lib.check(L.yt8m_sqnorm_multi(_p(graph.params), _p(graph.grads), chunks, c1 - c0, _p(graph.l2), gscale, partial,
                                       _p(graph.norms), lo, hi - lo, _p(graph.chunk_start_dev), c0, s))
    wi = getattr(graph, "wimg", None)
    tiles = wi is not None and wi.active
    # tensors that own operand images are updated by the tile pass, which rewrites the images where it rewrites the weight
    # (csrc/optim.hip adam_tile_kernel: bitwise the chunk kernel's arithmetic); the chunk pass leaves them alone
    _lib.check(L.yt8m_adam_multi_ex(_p(graph.params), _p(graph.adam_m), _p(graph.adam_v), _p(graph.grads), chunks, c1 - c0,
                                    _p(graph.l2), gscale, _p(graph.norms), clip, lr_t, beta1, beta2, eps,
                                    _p(wi.skip_dev) if tiles else None, s))
    if tiles:
        wi.adam(lo, hi, (gscale, clip, lr_t, beta1, beta2, eps), s)


# ------------------------------------------------------------------------------------------ autograd ops
def _token(graph=None):
    g = graph or get_default_graph()
    if g.token is None:
        g.begin_step()
    return g.token


class _Dropout(torch.autograd.Function):
    """tf.nn.dropout(x, keep_prob) (W/all_video_models/deep_combine_chain_model.py:57-58): x / keep_prob where the Philox
    stream of (seed, offset + element) keeps the element, else 0.  The mask is never stored: backward replays it."""

    @staticmethod
    def forward(ctx, x, keep_prob, seed, offset):
        x = _f32c(x)
        _dev(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().yt8m_dropout_f32(_p(x), _p(y), x.numel(), float(keep_prob), int(seed), int(offset), _stream()))
        ctx.args = (float(keep_prob), int(seed), int(offset))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        keep_prob, seed, offset = ctx.args
        _lib.check(_lib.lib().yt8m_dropout_f32(_p(dy), _p(dx), dy.numel(), keep_prob, seed, offset, _stream()))
        return dx, None, None, None


def dropout(x, keep_prob, seed=None, offset=0, graph=None):
    """seed None: the next key of the graph's random stream (variables.random_seed)."""
    if seed is None:
        seed = (graph or get_default_graph()).next_random_seed()
    return _Dropout.apply(x, keep_prob, seed, offset)


def dropout_(x, keep_prob, seed, offset=0):
    """In place, no autograd (used inside the fused recurrent-stack op)."""
    _dev(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    _lib.check(_lib.lib().yt8m_dropout_f32(_p(x), _p(x), x.numel(), float(keep_prob), int(seed), int(offset), _stream()))
    return x


class _AddNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stddev, seed, offset):
        x = _f32c(x)
        _dev(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().yt8m_add_noise_f32(_p(x), _p(y), x.numel(), float(stddev), int(seed), int(offset), _stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None, None


def add_noise(x, stddev, seed=None, offset=0, graph=None):
    """x + N(0, stddev^2) (W/all_frame_models/lstm_memory_model.py:62-63); the gradient passes through unchanged."""
    if seed is None:
        seed = (graph or get_default_graph()).next_random_seed()
    return _AddNoise.apply(x, stddev, seed, offset)


BF16_MIN_MACS = 1 << 27      # below this the cast passes cost more than the bf16 MFMAs save
BF16_MIN_ROWS = 512          # a weight matrix is re-cast every step (6 B/element): that only pays when >= ~100 activation
                             # rows share it; 512 keeps a margin (measured: B = 128 NetVLAD hidden FC loses, 1024-row chain wins)


def _use_bf16(bf16, M, N, K, weight_operand=True):
    """bf16 operands for C[M,N] = A[M,K].B[K,N]?  Needs an even reduction length (bf16 pairs), enough work, and -- when B
    is a weight matrix that has to be cast for this one product -- enough rows to amortise the cast."""
    return (bool(bf16) and K % 2 == 0 and M * N * K >= BF16_MIN_MACS and (not weight_operand or M >= BF16_MIN_ROWS))


class _Linear(torch.autograd.Function):
    """y = x.W (+ b) for a 2-D x.  slim.fully_connected without activation.
    bf16 = True (compute_dtype=bfloat16): each of the three GEMMs takes bf16 copies of its operands (K-contiguous on both
    sides, fp32 accumulation, fp32 master weights and gradients) when it is large enough to pay for the casts."""

    @staticmethod
    def forward(ctx, x, token, W, b, bf16):
        x2 = _f32c(x)
        M, K = x2.shape
        N = W.data.shape[1]
        if _use_bf16(bf16, M, N, K):
            y, = gemm_bf16_nt_grouped([dict(A=cast_bf16(x2), B=cast_bf16(W.data, transpose=True),
                                            bias=None if b is None else b.data)])
        else:
            # round 6: a fully-connected layer over >= LINEAR_H2_MIN_ROWS rows declares the h2 role for its FORWARD product (activation x
            # weight: one scale per operand matrix serves; the weight's half-plane image is resident, wimg.py) -- not for dx, whose dy
            # rows may differ by decades (the recurrent stack gives those per-row scales)
            role = "h2" if (LINEAR_FWD_H2 and _linear_h2_size(M, N, K) and N % 4 == 0 and K >= 512) else None
            # max |x| is measured ONCE: the forward product's split and the weight gradient's split of x^T both take the word
            ctx.xmax = h2_absmax(x2).view(torch.float32) if role == "h2" else None
            y = gemm_grouped([dict(A=x2, B=W.data, bias=None if b is None else b.data, absmaxA=ctx.xmax)], role=role)[0]
        ctx.save_for_backward(x2)
        ctx.W, ctx.b, ctx.bf16 = W, b, bf16
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        W, b = ctx.W, ctx.b
        dy = _f32c(dy)
        M, K = x.shape
        N = dy.shape[1]
        dyb = None
        if W.trainable and W.grad is not None:
            if _use_bf16(ctx.bf16, K, N, M, weight_operand=False):   # dW[K,N] = x^T dy: reduction over the M rows
                if ctx.needs_input_grad[0] and _use_bf16(ctx.bf16, M, K, N):
                    dyb, dyT = cast_bf16_both(dy)                   # dy feeds dW (transposed) and dx (plain): one pass
                else:
                    dyT = cast_bf16(dy, transpose=True)
                gemm_bf16_nt_grouped([dict(A=cast_bf16(x, transpose=True), B=dyT, out=W.grad, beta=W.grad_beta())])
                del dyT
            else:
                gemm_grouped([dict(A=x, B=dy, out=W.grad, beta=W.grad_beta(), absmaxA=getattr(ctx, "xmax", None))], transA=True, role="dw")
        if b is not None and b.trainable and b.grad is not None:
            colsum(dy, b.grad.view(-1), beta=b.grad_beta())
        dx = None
        if ctx.needs_input_grad[0]:
            if _use_bf16(ctx.bf16, M, K, N):                     # dx[M,K] = dy W^T: reduction over N
                dx, = gemm_bf16_nt_grouped([dict(A=dyb if dyb is not None else cast_bf16(dy), B=cast_bf16(W.data))])
            elif LINEAR_DX_H2 and _linear_h2_size(M, N, K) and N >= 512 and K % 4 == 0 and W.data.is_contiguous():
                dx = _linear_dx_h2_rows(dy, W)
            else:
                dx = gemm(dy, W.data, transB=True)
        return dx, None, None, None, None


def _linear_h2_size(M, N, K):
    """A fully-connected layer takes the h2 forms from LINEAR_H2_MIN_ROWS rows on, or when its product is large whatever the row count
    (LINEAR_H2_MIN_MNK: the NetVLAD / DBoF hidden layers at B = 128 -- K = 73 728 / 8 192 -- gain 12-14 % of their step; small layers do
    not pay the per-call passes over their activations back: cnn_chain +3 % with every layer on it)."""
    return M >= LINEAR_H2_MIN_ROWS or M * N * K >= LINEAR_H2_MIN_MNK


def _hoisted_role(M, N, K, bf16=False):
    """Role of a recurrent layer's hoisted input projection x . W_x over all time steps (GRU / LayerNorm-LSTM layers outside the native
    stack): "h2" under the fully-connected layers' size rule -- its input is the l2-normalised frames or a bounded recurrent output."""
    return "h2" if (LINEAR_FWD_H2 and not bf16 and _linear_h2_size(M, N, K) and N % 4 == 0 and K >= 512) else None


class _RowWindow(object):
    """The first rows of a weight matrix as the `W` of _linear_dx_h2_rows (a recurrent layer's input half W[:in] of its [in + H, .] weights)."""
    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data


def hoisted_dx(dy, Wrows, out=None, beta=0.0, bf16=False):
    """dx [M, K] (+)= dy [M, N] . Wrows [K, N]^T of a recurrent layer's hoisted input projection: row-scaled three-f16-product form under the
    fully-connected layers' size rule (time steps whose gradients differ by decades keep their own precision), else the generic product."""
    M, N = dy.shape
    K = Wrows.shape[0]
    if (LINEAR_DX_H2 and not bf16 and _linear_h2_size(M, N, K) and N >= 512 and K % 4 == 0 and Wrows.is_contiguous() and dy.is_contiguous()):
        return _linear_dx_h2_rows(dy, _RowWindow(Wrows), out=out, beta=beta)
    return gemm_any(dy, Wrows, out=out, transB=True, beta=beta, bf16=bf16)


def _linear_dx_h2_rows(dy, W, out=None, beta=0.0, row0=0):
    """dx [M, K] = dy [M, N] . W [K, N]^T as three f16 products with dy split ROW BY ROW -- one power of two per row (yt8m_h2_rowscales /
    _split_rows, undone by the product's rowscale): a row of dy whose gradient is decades below the largest keeps its own 22 bits, which
    one scale per matrix would not give it (the form the recurrent stack's dx takes, csrc/lstm_stack.hip).  W as an [K rows, K' = N]
    half-plane image: resident (wimg.py) or made here under its measured maximum.  row0 (a multiple of 32): only the columns [row0, K) of
    dx are computed -- rows [row0, K) of W are whole 32-row groups of its image --, a fresh dx holds zeros in front of them."""
    L = _lib.lib()
    M, N = dy.shape
    K = W.data.shape[0]
    assert row0 % 32 == 0 and 0 <= row0 < K
    dev = dy.device
    nb = lambda rows, kk: max(L.yt8m_x3_image_bytes(rows, kk) // 3 * 2, 16)
    S = torch.empty(M, dtype=torch.float32, device=dev)
    inv = torch.empty(M, dtype=torch.float32, device=dev)
    _lib.check(L.yt8m_h2_rowscales(_p(dy), M, N, N, _p(S), _p(inv), _stream()))
    dyi = torch.empty(nb(M, N), dtype=torch.uint8, device=dev)
    _lib.check(L.yt8m_h2_split_rows(_p(dy), M, N, N, _p(S), _p(dyi), _stream()))
    wp = ctypes.c_void_p(W.data.data_ptr())
    ptr = L.yt8m_wimg_lookup(wp, K, N, N, 0, 2, 0.0)
    keep = None
    if ptr:
        wimg_p, word_p = ctypes.c_void_p(ptr), ctypes.c_void_p(ptr - 256)          # (the scale word sits in front of a resident image)
    else:
        L.yt8m_wimg_note_demand(wp, K, N, N, 0, 2, 0.0)                            # its owner may keep it resident from the next step on
        word = h2_absmax(W.data)
        wi = torch.empty(nb(K, N), dtype=torch.uint8, device=dev)
        _lib.check(L.yt8m_h2_split(wp, K, N, N, 1.0, _p(word), _p(wi), None, None, _stream()))
        keep = (word, wi)
        wimg_p, word_p = _p(wi), _p(word)
    if out is not None:
        dx = out
    else:
        dx = (torch.zeros if row0 else torch.empty)((M, K), dtype=torch.float32, device=dev)
    ws = _workspace(dev)
    if row0:                                                     # (a K block of an h2 image: two 1 KiB half planes)
        wimg_p = ctypes.c_void_p(wimg_p.value + (row0 // 32) * ((N + 15) // 16) * 2048)
    _lib.check(L.yt8m_gemm_h2_nt_ex(M, K - row0, N, _p(dyi), 0, wimg_p, 0, ctypes.c_void_p(dx.data_ptr() + row0 * 4), K, None, 1.0, None,
                                    word_p, _p(inv), float(beta), _p(ws), ws.numel() * 4, _stream()))
    del keep
    return dx


SKINNY_MIN_ROWS = 2048       # below this the padded MFMA tiles are cheap enough


def skinny_ok(M, K, N, *tensors):
    """Layers with <= 16 outputs over many rows (attention logits): vector-ALU streaming kernels (csrc/gemm_skinny.hip)."""
    if not (N <= 16 and M >= SKINNY_MIN_ROWS and K % 4 == 0 and _lib.lib().yt8m_skinny_supported(M, K, N)):
        return False
    return all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in tensors)


def skinny_fwd(x, W, bias, y, beta=0.0):
    """y[M,N] (+)= x[M,K] . W[K,N] (+ bias); W may be a row slice of a wider-K weight (ld = N)."""
    M, K = x.shape
    N = W.shape[1]
    _lib.check(_lib.lib().yt8m_skinny_fwd_f32(_p(x), x.stride(0), _p(W), W.stride(0), _p(bias), _p(y), y.stride(0), M, K, N,
                                              float(beta), _stream()))
    return y


def skinny_dw(x, dy, dW, beta=0.0):
    M, K = x.shape
    N = dy.shape[1]
    ws = _workspace(x.device)
    _lib.check(_lib.lib().yt8m_skinny_dw_f32(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(dW), dW.stride(0), M, K, N, float(beta),
                                             _p(ws), ws.numel() * 4, _stream()))
    return dW


def skinny_dx(dy, W, dx=None, beta=0.0):
    M, N = dy.shape
    K = W.shape[0]
    if dx is None:
        dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.lib().yt8m_skinny_dx_f32(_p(dy), dy.stride(0), _p(W), W.stride(0), _p(dx), dx.stride(0), M, K, N, float(beta),
                                             _stream()))
    return dx


class _LinearCat(torch.autograd.Function):
    """slim.fully_connected on tf.concat(parts, axis=-1) without materialising the concatenation
    (W/all_frame_models/lstm_attention_max_pooling_model.py:51-56): y = sum_i parts_i . W[rows_i] + b.  W rows are taken in
    part order.  The first `nfull` parts are [M, K_i]; the others are PER-GROUP parts [M / rep, K_i] whose rows stand for
    `rep` consecutive rows of the concatenation each (a per-video vector tiled over the frames, e.g. the mean frame of
    lstm_positional_attention_max_pooling_model.py:77-84): their product is computed once per group and broadcast.
    Narrow outputs (<= 16) over many rows take the streaming kernels (csrc/gemm_skinny.hip)."""

    @staticmethod
    def forward(ctx, token, W, b, rep, nfull, *parts):
        parts = [_f32c(p) for p in parts]
        _dev(*parts)
        M = parts[0].shape[0]
        N = W.data.shape[1]
        assert nfull >= 1 and sum(p.shape[1] for p in parts) == W.data.shape[0], "parts do not add up to the weight's input width"
        y = torch.empty((M, N), dtype=torch.float32, device=parts[0].device)
        k0 = 0
        for i, p in enumerate(parts):
            K = p.shape[1]
            Wi = W.data[k0:k0 + K]
            if i >= nfull:
                assert p.shape[0] * rep == M
                t = gemm(p, Wi)                                       # [M / rep, N]: tiny
                y.view(-1, rep, N).add_(t.view(-1, 1, N))             # broadcast over the group (layout glue on [M, N])
            elif skinny_ok(M, K, N, p):
                skinny_fwd(p, Wi, b.data if (b is not None and i == 0) else None, y, beta=0.0 if i == 0 else 1.0)
            else:
                gemm(p, Wi, out=y, bias=b.data if (b is not None and i == 0) else None, beta=0.0 if i == 0 else 1.0)
            k0 += K
        ctx.save_for_backward(*parts)
        ctx.W, ctx.b, ctx.rep, ctx.nfull = W, b, rep, nfull
        return y

    @staticmethod
    def backward(ctx, dy):
        parts = ctx.saved_tensors
        W, b, rep, nfull = ctx.W, ctx.b, ctx.rep, ctx.nfull
        dy = _f32c(dy)
        M, N = dy.shape
        dxs = []
        wbeta = W.grad_beta() if (W.trainable and W.grad is not None) else None
        dyg = None
        k0 = 0
        for i, p in enumerate(parts):
            K = p.shape[1]
            Wi = W.data[k0:k0 + K]
            need_dx = ctx.needs_input_grad[5 + i]
            if i >= nfull:
                if dyg is None:
                    dyg = dy.view(-1, rep, N).sum(dim=1)              # [M / rep, N]
                if wbeta is not None:
                    gemm(p, dyg, out=W.grad[k0:k0 + K], transA=True, beta=wbeta, role="dw")
                dxs.append(gemm(dyg, Wi, transB=True) if need_dx else None)
            else:
                sk = skinny_ok(M, K, N, p, dy)
                if wbeta is not None:
                    if sk:
                        skinny_dw(p, dy, W.grad[k0:k0 + K], beta=wbeta)
                    else:
                        gemm(p, dy, out=W.grad[k0:k0 + K], transA=True, beta=wbeta, role="dw")
                dxs.append((skinny_dx(dy, Wi) if sk else gemm(dy, Wi, transB=True)) if need_dx else None)
            k0 += K
        if wbeta is not None:
            W.grad_done()
        if b is not None and b.trainable and b.grad is not None:
            colsum(dy, b.grad.view(-1), beta=b.grad_beta())
            b.grad_done()
        return (None, None, None, None, None) + tuple(dxs)


def linear_cat(parts, W, b=None, group_parts=()):
    """parts: tensors [..., K_i] with equal leading dims; group_parts: tensors [G, K_j] with G * rep = number of rows, each row
    standing for `rep` consecutive rows (appended after `parts` in the concatenation order)."""
    lead = parts[0].shape[:-1]
    flat = [p.reshape(-1, p.shape[-1]) for p in parts]
    rep = 1
    if group_parts:
        rep = flat[0].shape[0] // group_parts[0].shape[0]
    y = _LinearCat.apply(_token(W._graph), W, b, rep, len(flat), *flat, *group_parts)
    return y.view(*lead, y.shape[-1])


def linear(x, W, b=None, bf16=None):
    """Rank-N input is flattened on the leading dims like slim.fully_connected."""
    if W.data.shape[1] <= 16 and x.numel() // x.shape[-1] >= SKINNY_MIN_ROWS:
        return linear_cat([x], W, b)
    lead = x.shape[:-1]
    if bf16 is None:
        bf16 = FLAGS.compute_dtype == "bfloat16"
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), _token(W._graph), W, b, bool(bf16))
    return y.view(*lead, y.shape[-1])


class _VarTensor(torch.autograd.Function):
    """A Variable used as a plain tensor in layout glue (tile / concat), e.g. the positional embedding of
    W/all_frame_models/lstm_positional_attention_max_pooling_model.py:68-76: forward hands out the parameter, backward
    lands the gradient in the arena slice (beta 0/1 like every other op)."""

    @staticmethod
    def forward(ctx, token, var):
        ctx.var = var
        return var.data.view_as(var.data)

    @staticmethod
    def backward(ctx, g):
        var = ctx.var
        if var.grad is not None and g is not None:
            if var.grad_beta() == 0.0:
                var.grad.copy_(g.view_as(var.grad))
            else:
                var.grad.add_(g.view_as(var.grad))
            var.grad_done()
        return None, None


def as_tensor(var):
    return _VarTensor.apply(_token(var._graph), var)


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        y = act_fwd(kind, x)
        ctx.kind = kind
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return act_bwd(ctx.kind, y, dy), None


def activation(x, kind):
    return _Act.apply(x, kind)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = _f32c(x)
        ctx.save_for_backward(x)
        ctx.eps = eps
        return l2norm_fwd(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return l2norm_bwd(x, dy, ctx.eps), None


def l2_normalize(x, eps=1e-12):
    """tf.nn.l2_normalize on the last axis (differentiable)."""
    return _L2Norm.apply(x, eps)


class _MoeHead(torch.autograd.Function):
    """MoE block of W/all_video_models/moe_model.py:40-64: two GEMMs + mixing kernel; backward per Appendix G."""

    @staticmethod
    def forward(ctx, x, token, Wg, We, be, V, M, bf16, dx_from=0):
        x2 = _f32c(x)
        ctx.images = {} if (bf16 and FUSED_MIX_BF16 and M == 2 and ctx.needs_input_grad[1]) else None
        ctx.words = {}                                  # max |x| measured once: the weight gradient's split of x^T takes the same word
        Zg, Ze = _moe_logits(x2, Wg, We, be, bf16, keep=ctx.images, z16=_z16_ok(x2, Wg, We, V, M, bf16, ctx.images is not None),
                             words=ctx.words)
        ctx.bf16 = bf16
        ctx.dx_from = dx_from
        p = moe_mix_fwd(Zg, Ze, V, M)
        ctx.save_for_backward(x2)
        ctx.Z = (Zg, Ze)
        ctx.vars = (Wg, We, be)
        ctx.VM = (V, M)
        return p

    @staticmethod
    def backward(ctx, dp):
        (x,) = ctx.saved_tensors
        Zg, Ze = ctx.Z
        Wg, We, be = ctx.vars
        V, M = ctx.VM
        ctx.Z = None
        if Zg.dtype == torch.bfloat16 and not (_fused_mix_bf16_ok(ctx, x, Zg, Ze, M) and _b1_ok(Zg.shape[0], Zg.shape[1], x.shape[1])
                                               and _b1_ok(x.shape[1], Zg.shape[1], Zg.shape[0])):
            Zg, Ze = Zg.float(), Ze.float()         # (not reached with _z16_ok's predicates; the fp32 passes below read fp32 logits)
        if _fused_mix_bf16_ok(ctx, x, Zg, Ze, M):
            dx = _moe_head_bwd_bf16_fused(ctx, x, Zg, Ze, Wg, We, be, V, M, dp=_f32c(dp))
            return dx, None, None, None, None, None, None, None, None
        moe_mix_bwd_(Zg, Ze, dp, V, M)            # in place: Zg <- dL/dZg, Ze <- dL/dZe
        dx = _moe_head_param_grads(ctx, x, Zg, Ze, Wg, We, be, xmax=(getattr(ctx, "words", None) or {}).get("x"))
        return dx, None, None, None, None, None, None, None, None


FUSED_MIX_BF16 = True     # compute_dtype=bfloat16, M == 2: mixing backward writes the bf16 GEMM operands itself (csrc/moe_bf16.hip)


def _moe_head_bwd_bf16_fused(ctx, x, Zg, Ze, Wg, We, be, V, M, dp=None, labels=None, ldt=0, dscale=1.0, up=None):
    """bf16 configuration: ONE pass over the fp32 logits produces dL/dZ as bf16 in both layouts + the bias partial sums, then
    the same three bf16 products as _moe_head_param_grads_bf16 (no fp32 dZ, no cast passes, no colsum over dZ_e)."""
    L = _lib.lib()
    B = Zg.shape[0]
    dev = Zg.device
    if _b1_ok(B, Zg.shape[1], x.shape[1]) and _b1_ok(x.shape[1], Zg.shape[1], B):
        return _moe_head_bwd_bf16_images(ctx, x, Zg, Ze, Wg, We, be, V, M, dp, labels, ldt, dscale, up)
    Zgb, ZgT = _bf16_empty(B, Zg.shape[1], dev), _bf16_empty(Zg.shape[1], B, dev)
    Zeb, ZeT = _bf16_empty(B, Ze.shape[1], dev), _bf16_empty(Ze.shape[1], B, dev)
    part = torch.empty((L.yt8m_moe_mix_bwd_bf16_partial_rows(B), Ze.shape[1]), dtype=torch.float32, device=dev) \
        if be.grad is not None else None
    _lib.check(L.yt8m_moe_mix_bwd_bf16(_p(Zg), _p(Ze), _p(dp), _p(labels), ldt, B, V, M, XENT_EPS, float(dscale), _p(up),
                                       _p(Zgb), Zgb.stride(0), _p(ZgT), ZgT.stride(0), _p(Zeb), Zeb.stride(0), _p(ZeT),
                                       ZeT.stride(0), _p(part), _stream()))
    dx = None
    if ctx.needs_input_grad[0]:
        dx, = gemm_bf16_nt_grouped([dict(A=Zgb, B=cast_bf16(Wg.data))])
        gemm_bf16_nt_grouped([dict(A=Zeb, B=cast_bf16(We.data), out=dx, beta=1.0)])
    del Zgb, Zeb
    if Wg.grad is not None and We.grad is not None:
        xT = cast_bf16(x, transpose=True)
        overlap = Wg._graph is not None and Wg._graph.grad_ready_hook is not None
        pg = dict(A=xT, B=ZgT, out=Wg.grad, beta=Wg.grad_beta())
        pe = dict(A=xT, B=ZeT, out=We.grad, beta=We.grad_beta())
        if overlap:
            gemm_bf16_nt_grouped([pg])
            Wg.grad_done()
            gemm_bf16_nt_grouped([pe])
            We.grad_done()
        else:
            gemm_bf16_nt_grouped([pg, pe])
            Wg.grad_done()
            We.grad_done()
    if be.grad is not None:
        colsum(part, be.grad.view(-1), beta=be.grad_beta())
        be.grad_done()
    return dx


def _moe_head_bwd_bf16_images(ctx, x, Zg, Ze, Wg, We, be, V, M, dp, labels, ldt, dscale, up):
    """_moe_head_bwd_bf16_fused on operand images: the mixing backward writes dL/dZ straight into the one-plane images the b1
    kernel reads (both orientations), the weights and x^T are rounded into images by one pass each; three image products."""
    L = _lib.lib()
    B, Ng, Ne = Zg.shape[0], Zg.shape[1], Ze.shape[1]
    dev = Zg.device
    mk = lambda rows, K: X3Image(torch.empty(max(L.yt8m_x3_image_bytes(rows, K) // 3, 16), dtype=torch.uint8, device=dev), rows, K)
    Zgi, ZgTi, Zei, ZeTi = mk(B, Ng), mk(Ng, B), mk(B, Ne), mk(Ne, B)
    part = torch.empty((L.yt8m_moe_mix_bwd_bf16_partial_rows(B), Ne), dtype=torch.float32, device=dev) if be.grad is not None else None
    kb = lambda K: (K + 15) // 16
    mix = L.yt8m_moe_mix_bwd_bf16_images_z16 if Zg.dtype == torch.bfloat16 else L.yt8m_moe_mix_bwd_bf16_images
    _lib.check(mix(_p(Zg), _p(Ze), _p(dp), _p(labels), ldt, B, V, M, XENT_EPS, float(dscale), _p(up),
                   _p(Zgi.buf), kb(Ng), _p(ZgTi.buf), kb(B), _p(Zei.buf), kb(Ne), _p(ZeTi.buf), kb(B), _p(part), _stream()))
    kept = getattr(ctx, "images", None) or {}                                   # written by the forward's image passes
    ctx.images = None
    dx = None
    if ctx.needs_input_grad[0]:
        dx, = gemm_b1_grouped([dict(A=Zgi, B=kept.get("Wg") or bf16_image(Wg.data))])           # W [D, N] as [D rows, K = N]
        gemm_b1_grouped([dict(A=Zei, B=kept.get("We") or bf16_image(We.data), out=dx, beta=1.0)])
    del Zgi, Zei
    if Wg.grad is not None and We.grad is not None:
        xT = kept.get("xT") or bf16_image(x, transpose=True)                    # [D rows, K = B]
        overlap = Wg._graph is not None and Wg._graph.grad_ready_hook is not None
        pg = dict(A=xT, B=ZgTi, out=Wg.grad, beta=Wg.grad_beta())
        pe = dict(A=xT, B=ZeTi, out=We.grad, beta=We.grad_beta())
        if overlap:
            gemm_b1_grouped([pg])
            Wg.grad_done()
            gemm_b1_grouped([pe])
            We.grad_done()
        else:
            gemm_b1_grouped([pg, pe])
            Wg.grad_done()
            We.grad_done()
    if be.grad is not None:
        colsum(part, be.grad.view(-1), beta=be.grad_beta())
        be.grad_done()
    return dx


def _fused_mix_bf16_ok(ctx, x, Zg, Ze, M):
    return (FUSED_MIX_BF16 and getattr(ctx, "bf16", False) and M == 2 and _bf16_ok(x) and Zg.shape[1] % 2 == 0
            and Ze.shape[1] % 2 == 0 and Zg.is_contiguous() and Ze.is_contiguous())


def _bf16_ok(x2):
    """bf16 GEMMs need even reduction lengths (D for the forward, the batch for dW) and enough rows to amortise the
    per-step cast of the weights (see BF16_MIN_ROWS)."""
    return x2.shape[0] % 2 == 0 and x2.shape[1] % 2 == 0 and x2.shape[0] >= BF16_MIN_ROWS


# OPT-IN (YT8M_Z16_LOGITS=1): configs[4] at B = 1024 19.88 / 19.71 -> 19.25 / 19.19 ms/step (-3 %), but the full-size golden replay of the same
# configuration (tests/test_gpu_fullsize_golden.py[c4_composite_bf16]) then sees the attention weights' gradient at 0.67 of its fp64
# magnitude (abs-sum 2.61 against 3.89; tolerance 0.2 of the scale, met with fp32 logits): rounding the logits themselves -- not just the
# products' operands -- to 8 significant bits is more than that gradient takes.  The kernels are exact in kind (bit for bit the fp32-logit
# passes on rounded logits: tests/test_gpu_round6.py::test_bf16_logits_*).
Z16_LOGITS = os.environ.get("YT8M_Z16_LOGITS", "0") != "0"
MOE_LOGITS_H2 = os.environ.get("YT8M_MOE_LOGITS_H2", "1") != "0"
MOE_LOGITS_H2_MIN_ROWS = int(os.environ.get("YT8M_MOE_LOGITS_H2_MIN_ROWS", "512"))   # 512: DeepCombineChain at B = 512 12.95 -> 12.03 ms/step (tools/r6_moe_rows.sh); B = 128 heads stay on the fp32 kernel (+0.3 ms on the headline with them)
LINEAR_FWD_H2 = os.environ.get("YT8M_LINEAR_FWD_H2", "1") != "0"
LINEAR_H2_MIN_ROWS = int(os.environ.get("YT8M_LINEAR_H2_MIN_ROWS", "512"))
LINEAR_DX_H2 = os.environ.get("YT8M_LINEAR_DX_H2", "1") != "0"
LINEAR_H2_MIN_MNK = float(os.environ.get("YT8M_LINEAR_H2_MIN_MNK", "1e9"))
MOE_DX_H2 = os.environ.get("YT8M_MOE_DX_H2", "1") != "0"
MOE_DX_FROM = os.environ.get("YT8M_MOE_DX_FROM", "1") != "0"           # chain models: no dx for the data columns in front of a head's input
MIX_BWD_ABSMAX = os.environ.get("YT8M_MIX_BWD_ABSMAX", "1") != "0"     # the mixing backward measures max |dZ| for the dW products' h2 split


def _z16_ok(x2, Wg, We, V, M, bf16, training):
    """Round 6: in the bf16 configuration the head's logits leave the product as bf16 -- the configuration's own
    precision -- when everything behind them reads bf16: the bf16-logit mixing pass (M == 2, B V % 4 == 0, V % 4 == 0) and, in training,
    the image form of the fused mixing backward (same predicates as _moe_head_bwd_bf16_fused takes).  Halves the 773 MB a [8192, 5 x 4716]
    stage of configs[4] writes and the two passes behind it read.  Opt-in: see Z16_LOGITS."""
    B, D = x2.shape
    Ng = Wg.data.shape[1]
    if not (Z16_LOGITS and bf16 and M == 2 and _bf16_ok(x2) and V % 4 == 0 and Ng == 3 * V and We.data.shape[1] == 2 * V):
        return False
    if not _b1_ok(B, Ng, D):
        return False
    if training and not (FUSED_MIX_BF16 and _b1_ok(D, Ng, B) and Wg.grad is not None and We.grad is not None):
        return False
    return True


def _moe_logits(x2, Wg, We, be, bf16, keep=None, z16=False, words=None):
    """Zg = x.Wg, Ze = x.We + be as ONE persistent launch; bf16: operands are bf16 copies (x, Wg^T, We^T: both sides
    K-contiguous), accumulation and outputs stay fp32.  keep (a dict, training only): the image pass of x / Wg / We writes the
    OTHER orientation too -- what the backward products read (x^T for dW, W for dx) -- so each tensor is read once per step."""
    if bf16 and _bf16_ok(x2) and _b1_ok(x2.shape[0], Wg.data.shape[1], x2.shape[1]):
        odt = torch.bfloat16 if z16 else torch.float32
        if keep is not None and _b1_ok(x2.shape[1], Wg.data.shape[1], x2.shape[0]):
            xi, keep["xT"] = bf16_image(x2, both=True)
            keep["Wg"], WgT = bf16_image(Wg.data, both=True)
            keep["We"], WeT = bf16_image(We.data, both=True)
            return gemm_b1_grouped([dict(A=xi, B=WgT, out_dtype=odt), dict(A=xi, B=WeT, bias=be.data, out_dtype=odt)])
        xi = bf16_image(x2)                                  # [B rows, K = D]; W^T as [N rows, K = D]: the transposing image pass
        return gemm_b1_grouped([dict(A=xi, B=bf16_image(Wg.data, transpose=True), out_dtype=odt),
                                dict(A=xi, B=bf16_image(We.data, transpose=True), bias=be.data, out_dtype=odt)])
    if bf16 and _bf16_ok(x2):
        xb = cast_bf16(x2)
        return gemm_bf16_nt_grouped([dict(A=xb, B=cast_bf16(Wg.data, transpose=True)),
                                     dict(A=xb, B=cast_bf16(We.data, transpose=True), bias=be.data)])
    # fp32 configuration: the logits product declares the h2 role (three f16 products under one scale per operand matrix) -- its input is
    # l2-normalised or a bounded hidden activation, each operand one weight matrix (MOE_LOGITS_H2 / YT8M_MOE_LOGITS_H2=0: six-product form)
    # -- from MOE_LOGITS_H2_MIN_ROWS rows on: the weights' half-plane images are made per call (absmax + split of [D, 5V]), which a
    # B = 128 product does not pay back (NetVLADModel at B = 128: 2.57 -> 2.88 ms/step with it; break-even by the split / product rates
    # ~1 000 rows while the images were made per call; with them resident (wimg.py) 512 rows gain too: DeepCombineChain at B = 512
    # 12.95 -> 12.03 ms/step; measured -6 % at 1 024, -7 % at 8 192)
    h2 = MOE_LOGITS_H2 and x2.shape[0] >= MOE_LOGITS_H2_MIN_ROWS
    xmax = h2_absmax(x2).view(torch.float32) if (h2 and words is not None) else None      # measured once: the weight gradient's split of
    if xmax is not None:                                                                  # x^T takes the same word (words["x"])
        words["x"] = xmax
    return gemm_grouped([dict(A=x2, B=Wg.data, absmaxA=xmax), dict(A=x2, B=We.data, bias=be.data, absmaxA=xmax)], role="h2" if h2 else None)


class _MoeHeadXent(torch.autograd.Function):
    """MoE block + CrossEntropyLoss in one op: returns (p, loss).  The head's GEMMs are the same grouped launches as
    _MoeHead; mixing+loss and (dL/dp -> dL/dZ) are single fused passes (yt8m_moe_mix_xent_fwd/bwd)."""

    @staticmethod
    def forward(ctx, x, token, Wg, We, be, labels, V, M, bf16):
        x2 = _f32c(x)
        ctx.images = {} if (bf16 and FUSED_MIX_BF16 and M == 2 and ctx.needs_input_grad[1]) else None
        ctx.words = {}
        Zg, Ze = _moe_logits(x2, Wg, We, be, bf16, keep=ctx.images, words=ctx.words)
        ctx.bf16 = bf16
        B = x2.shape[0]
        lab, ldt = _labels_arg(labels)
        if tuple(lab.shape) != (B, V):
            raise ValueError("labels shape %s != predictions shape %s" % (tuple(lab.shape), (B, V)))
        L = _lib.lib()
        ws = torch.empty((L.yt8m_moe_mix_xent_workspace_bytes(B, V) + 3) // 4, dtype=torch.float32, device=x2.device)
        p = torch.empty((B, V), dtype=torch.float32, device=x2.device)
        loss = torch.empty((), dtype=torch.float32, device=x2.device)
        _lib.check(L.yt8m_moe_mix_xent_fwd(_p(Zg), _p(Ze), _p(lab), ldt, _p(p), _p(loss), B, V, M, XENT_EPS, _p(ws), _stream()))
        ctx.save_for_backward(x2)
        ctx.Z = (Zg, Ze)
        ctx.vars = (Wg, We, be)
        ctx.lab = (lab, ldt)
        ctx.VM = (V, M)
        ctx.mark_non_differentiable(p)          # predictions leave through the loss only on this path
        ctx.set_materialize_grads(False)        # no [B,V] zero-fill for the unused dL/dp slot
        return p, loss

    @staticmethod
    def backward(ctx, dp_unused, dloss):
        (x,) = ctx.saved_tensors
        Zg, Ze = ctx.Z
        Wg, We, be = ctx.vars
        lab, ldt = ctx.lab
        V, M = ctx.VM
        ctx.Z = None
        if dloss is None:
            return (None,) * 9
        if _fused_mix_bf16_ok(ctx, x, Zg, Ze, M):
            dx = _moe_head_bwd_bf16_fused(ctx, x, Zg, Ze, Wg, We, be, V, M, labels=lab, ldt=ldt, dscale=1.0 / x.shape[0],
                                          up=_f32c(dloss.reshape(1)))
            return dx, None, None, None, None, None, None, None, None
        zmax = None
        if MIX_BWD_ABSMAX:
            zmax = torch.empty(2, dtype=torch.float32, device=Zg.device)  # max |dZg|, max |dZe| as float bits, written by the same pass
            _lib.check(_lib.lib().yt8m_moe_mix_xent_bwd_absmax(_p(Zg), _p(Ze), _p(lab), ldt, _p(_f32c(dloss.reshape(1))), x.shape[0], V, M,
                                                               XENT_EPS, 1.0, _p(zmax), _stream()))
        else:
            _lib.check(_lib.lib().yt8m_moe_mix_xent_bwd(_p(Zg), _p(Ze), _p(lab), ldt, _p(_f32c(dloss.reshape(1))), x.shape[0], V, M,
                                                        XENT_EPS, 1.0, _stream()))
        dx = _moe_head_param_grads(ctx, x, Zg, Ze, Wg, We, be, zmax=zmax, xmax=ctx.words.get("x"))
        return dx, None, None, None, None, None, None, None, None


def _moe_head_param_grads_bf16(ctx, x, Zg, Ze, Wg, We, be):
    """Same gradients on bf16 MFMAs: every product is written as A.B^T with K-contiguous bf16 copies
    (dW = x^T.dZ = (x^T) . (dZ^T)^T, dx = dZ . (W)^T with W [D,N] itself K-contiguous over N)."""
    dx = None
    need_dw = Wg.grad is not None and We.grad is not None
    dx_bf16 = ctx.needs_input_grad[0] and Zg.shape[1] % 2 == 0 and Ze.shape[1] % 2 == 0
    ZgT = ZeT = None
    if ctx.needs_input_grad[0]:
        if dx_bf16:
            # dZ is needed K-contiguous for dx and row-transposed for dW: both layouts from one pass over the fp32 logits
            Zgb, ZgT = cast_bf16_both(Zg) if need_dw else (cast_bf16(Zg), None)
            Zeb, ZeT = cast_bf16_both(Ze) if need_dw else (cast_bf16(Ze), None)
            dx, = gemm_bf16_nt_grouped([dict(A=Zgb, B=cast_bf16(Wg.data))])
            gemm_bf16_nt_grouped([dict(A=Zeb, B=cast_bf16(We.data), out=dx, beta=1.0)])
            del Zgb, Zeb
        else:                                      # odd V*(M+1): the reduction length cannot be packed in bf16 pairs
            dx = gemm(Zg, Wg.data, transB=True)
            gemm(Ze, We.data, out=dx, transB=True, beta=1.0)
    if need_dw:
        xT = cast_bf16(x, transpose=True)
        overlap = Wg._graph is not None and Wg._graph.grad_ready_hook is not None
        pg = dict(A=xT, B=ZgT if ZgT is not None else cast_bf16(Zg, transpose=True), out=Wg.grad, beta=Wg.grad_beta())
        pe = dict(A=xT, B=ZeT if ZeT is not None else cast_bf16(Ze, transpose=True), out=We.grad, beta=We.grad_beta())
        if overlap:
            gemm_bf16_nt_grouped([pg])
            Wg.grad_done()
            gemm_bf16_nt_grouped([pe])
            We.grad_done()
        else:
            gemm_bf16_nt_grouped([pg, pe])
            Wg.grad_done()
            We.grad_done()
    if be.grad is not None:
        colsum(Ze, be.grad.view(-1), beta=be.grad_beta())
        be.grad_done()
    return dx


_SIDE = {}
DEFER_HEAD_DW = os.environ.get("YT8M_DEFER_HEAD_DW", "0") != "0"   # measured: no gain (the backward pass is work-bound), opt-in


def side_stream(graph):
    """The side stream on which an op may leave weight-gradient work that nothing later in the backward pass reads (round 5): the
    classifier head's dW products when a recurrent stack's backward pass follows -- its first recurrence holds half the chip for a
    millisecond and waits only for dx.  None when the step must not defer (no stack follows, data parallel: the reducer starts a
    gradient's all-reduce from the current stream)."""
    if not DEFER_HEAD_DW or graph is None or not getattr(graph, "defer_head_dw", False) or graph.grad_ready_hook is not None:
        return None
    st = _SIDE.get(graph.device)
    if st is None:
        st = torch.cuda.Stream(device=graph.device)
        _SIDE[graph.device] = st
    return st


def join_side_work(graph):
    """Makes the current stream wait for everything ops left on side streams in this backward pass (called by the recurrent
    stack's backward and by TrainGraph.step before the optimiser pass)."""
    if graph is None:
        return
    for st in getattr(graph, "side_pending", None) or ():
        torch.cuda.current_stream(graph.device).wait_stream(st)
    graph.side_pending = []


def _moe_head_param_grads(ctx, x, Zg, Ze, Wg, We, be, zmax=None, xmax=None):
    """dW_g = x^T dZ_g, dW_e = x^T dZ_e, db_e = colsum(dZ_e), dx = dZ_g W_g^T + dZ_e W_e^T.
    zmax (a [2] float tensor, optional): max |dZ_g|, max |dZ_e| as float bits, already measured by the pass that wrote them."""
    mg = zmax[0:1] if zmax is not None else None
    me = zmax[1:2] if zmax is not None else None
    if getattr(ctx, "bf16", False) and _bf16_ok(x):
        return _moe_head_param_grads_bf16(ctx, x, Zg, Ze, Wg, We, be)
    dx = None
    if ctx.needs_input_grad[0]:                    # before the weights' gradient slots are released to an optimiser
        # columns [0, k0) of x are data (moe_head's dx_from): no gradient is computed for them -- k0 on a 32-row group of the weights' images
        k0 = int(getattr(ctx, "dx_from", 0) or 0)
        k0 = k0 if (MOE_DX_FROM and 0 < k0 < Wg.data.shape[0] and k0 % 32 == 0) else 0
        if (MOE_DX_H2 and Zg.shape[0] >= MOE_LOGITS_H2_MIN_ROWS and Wg.data.shape[0] % 4 == 0 and Wg.data.is_contiguous()
                and We.data.is_contiguous() and Zg.is_contiguous() and Ze.is_contiguous()):
            # round 6: from 1 024 rows on, three f16 products with dZ split row by row against the weights' half-plane images (the form of
            # ops._linear_dx_h2_rows) instead of the fp32-MFMA kernel these 16-tile, K ~ 14 000 products fell back to
            dx = _linear_dx_h2_rows(Zg, Wg, row0=k0)
            _linear_dx_h2_rows(Ze, We, out=dx, beta=1.0, row0=k0)
        elif k0:
            dx = torch.zeros((Zg.shape[0], Wg.data.shape[0]), dtype=torch.float32, device=Zg.device)
            gemm(Zg, Wg.data[k0:], transB=True, out=dx[:, k0:])
            gemm(Ze, We.data[k0:], out=dx[:, k0:], transB=True, beta=1.0)
        else:
            dx = gemm(Zg, Wg.data, transB=True)
            gemm(Ze, We.data, out=dx, transB=True, beta=1.0)
    overlap = Wg._graph is not None and Wg._graph.grad_ready_hook is not None
    side = side_stream(Wg._graph) if (dx is not None and Wg.grad is not None and We.grad is not None and be.grad is not None) else None
    if side is not None:
        # dx is on its way to the recurrent stack; the parameter gradients are read by the optimiser only: off the critical chain
        g = Wg._graph
        side.wait_stream(torch.cuda.current_stream(g.device))
        for t in (x, Zg, Ze):
            t.record_stream(side)                  # (their memory must not be handed out again before the side stream is done)
        bw, bwe, bbe = Wg.grad_beta(), We.grad_beta(), be.grad_beta()
        with torch.cuda.stream(side):
            gemm_grouped([dict(A=x, B=Zg, out=Wg.grad, beta=bw, absmaxA=xmax, absmaxB=mg), dict(A=x, B=Ze, out=We.grad, beta=bwe, absmaxA=xmax, absmaxB=me)],
                         transA=True, role="dw")
            colsum(Ze, be.grad.view(-1), beta=bbe)
        if not hasattr(g, "side_pending") or g.side_pending is None:
            g.side_pending = []
        g.side_pending.append(side)
        Wg.grad_done()
        We.grad_done()
        be.grad_done()
        return dx
    if Wg.grad is not None and We.grad is not None and not overlap:
        gemm_grouped([dict(A=x, B=Zg, out=Wg.grad, beta=Wg.grad_beta(), absmaxA=xmax, absmaxB=mg),
                      dict(A=x, B=Ze, out=We.grad, beta=We.grad_beta(), absmaxA=xmax, absmaxB=me)], transA=True, role="dw")
        Wg.grad_done()
        We.grad_done()
    elif Wg.grad is not None and We.grad is not None:
        # data-parallel: finish the big gate gradient first so its all-reduce rides under the expert GEMM
        gemm_grouped([dict(A=x, B=Zg, out=Wg.grad, beta=Wg.grad_beta(), absmaxA=xmax, absmaxB=mg)], transA=True, role="dw")
        Wg.grad_done()
        gemm_grouped([dict(A=x, B=Ze, out=We.grad, beta=We.grad_beta(), absmaxA=xmax, absmaxB=me)], transA=True, role="dw")
        We.grad_done()
    if be.grad is not None:
        colsum(Ze, be.grad.view(-1), beta=be.grad_beta())
        be.grad_done()
    return dx


def moe_head_xent(x, Wg, We, be, labels, vocab_size, num_mixtures, bf16=False):
    return _MoeHeadXent.apply(x, _token(Wg._graph), Wg, We, be, labels, vocab_size, num_mixtures, bool(bf16))


def moe_head(x, Wg, We, be, vocab_size, num_mixtures, bf16=False, dx_from=0):
    """dx_from: the first `dx_from` columns of x are DATA (the chain models concatenate the model input in front of what they learned:
    deep_combine_chain_model.py:66-70) -- their gradient is not computed (fp32 configuration; the returned dx holds zeros there)."""
    return _MoeHead.apply(x, _token(Wg._graph), Wg, We, be, vocab_size, num_mixtures, bool(bf16), int(dx_from))


class _Xent(torch.autograd.Function):
    """CrossEntropyLoss (W/losses.py:110-130); backward recomputes dL/dp from (p, y) with the upstream scalar read
    on the device, so no host sync and no extra elementwise pass."""

    @staticmethod
    def forward(ctx, p, labels, weights, scale):
        loss, _ = xent_fwd(p, labels, weights, want_dp=False, upstream=1.0)
        ctx.save_for_backward(p)
        ctx.labels, ctx.weights, ctx.scale = labels, weights, scale
        return loss * scale if scale != 1.0 else loss

    @staticmethod
    def backward(ctx, dloss):
        (p,) = ctx.saved_tensors
        dp = xent_bwd(p, ctx.labels, ctx.weights, dloss.reshape(1), upstream=ctx.scale)
        return dp, None, None, None


def cross_entropy(p, labels, weights=None, scale=1.0):
    return _Xent.apply(p, labels, weights, float(scale))
