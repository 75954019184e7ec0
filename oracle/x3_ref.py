"""ORACLE (test infrastructure only -- never imported by the product path).

The arithmetic of the "x3" GEMM (youtube-8m_amd/csrc/gemm_x3.hip) restated in numpy: an fp32 product on the bf16 matrix pipe.
The reference multiplies in fp32 (tf.matmul on float32 tensors: W/all_frame_models/lstm_model.py:44-47 through BasicLSTMCell,
W/all_video_models/moe_model.py:43-55 through slim.fully_connected); this build computes those products as

    a = a1 + a2 + a3  (exactly; a_i = bfloat16 values, a1 = rne(a), a2 = rne(a - a1), a3 = rne(a - a1 - a2))
    a . b ~= a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1          (fp32 accumulation; dropped: a2 b3 + a3 b2 + a3 b3)

so the only deviation from an exact product is the dropped terms (<= 2^-24 |a| |b|, below one fp32 rounding of the product) plus
the fp32 accumulation the reference's GEMM has as well.  What is restated here:

  bf16_rne_bits / split3   -- the integer rounding the device uses (round to nearest even on the upper 16 bits; finite inputs)
  image                    -- the operand image the split pass writes: [ceil(rows/32)][ceil(K/16)][3 planes][32 rows][2][8] bf16,
                              the 8-element half h of row r stored in slot h ^ ((r >> 3) & 1); padding rows / columns are zero
  six_products             -- the six-term sum above with fp64 accumulation (the device accumulates in fp32 inside the MFMA, in an
                              order this restatement does not fix: device results are compared with a tolerance, the split planes
                              and images bit for bit)

Pinning: nothing in the reference pins this (it has no such kernel); the restatement is checked against its own defining
properties in tests/test_oracle_x3.py (exact reconstruction, bf16 representability, error bound against an fp64 product) and the
device's images are compared with it bit for bit in tests/test_gpu_x3.py."""
import numpy as np

TERMS = ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0))


def bf16_rne_bits(x):
    """float32 array -> uint32 array holding the 16 bits of the bfloat16 nearest to x (ties to even); finite x."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.uint32)


def bits_to_f32(h):
    return (np.asarray(h, dtype=np.uint32) << np.uint32(16)).view(np.float32)


def split3(x):
    """-> (h1, h2, h3) bit patterns (uint32, 16 significant bits) with f(h1) + f(h2) + f(h3) == x exactly."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    h1 = bf16_rne_bits(x)
    r1 = (x - bits_to_f32(h1)).astype(np.float32)               # exact: the difference fits 16 mantissa bits
    h2 = bf16_rne_bits(r1)
    r2 = (r1 - bits_to_f32(h2)).astype(np.float32)               # exact
    h3 = bf16_rne_bits(r2)
    return h1, h2, h3


def planes(x):
    """-> [3, ...] float32: the three bf16 terms as values."""
    return np.stack([bits_to_f32(h) for h in split3(x)])


def image(x, scale=1.0):
    """x: fp32 [rows, K] operand (used K-contiguous) -> uint16 [ceil(rows/32), ceil(K/16), 3, 32, 2, 8], the bytes yt8m_x3_split
    writes for its `plain` output (pass x.T for the `trans` output).  scale multiplies in fp32 before the split."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if scale != 1.0:
        x = (x * np.float32(scale)).astype(np.float32)
    R, K = x.shape
    RG, KB = (R + 31) // 32, (K + 15) // 16
    pad = np.zeros((RG * 32, KB * 16), dtype=np.float32)
    pad[:R, :K] = x
    h = np.stack(split3(pad)).astype(np.uint16)                  # [3, RG*32, KB*16]
    h = h.reshape(3, RG, 32, KB, 2, 8).transpose(1, 3, 0, 2, 4, 5)   # [RG, KB, plane, row, half, 8]
    out = np.empty_like(h)
    sw = (np.arange(32) >> 3) & 1
    for r in range(32):
        out[:, :, :, r, sw[r], :] = h[:, :, :, r, 0, :]
        out[:, :, :, r, sw[r] ^ 1, :] = h[:, :, :, r, 1, :]
    return np.ascontiguousarray(out)


def six_products(A, B):
    """A [M, K], B [N, K] fp32 -> fp64 [M, N]: the six kept partial products of the split operands, accumulated in fp64."""
    pa = planes(A).astype(np.float64)
    pb = planes(B).astype(np.float64)
    C = np.zeros((A.shape[0], B.shape[0]), dtype=np.float64)
    for p, q in TERMS:
        C += pa[p] @ pb[q].T
    return C


# ---- "h2": two IEEE-half planes, three products (csrc/x3_image.h split_h2, csrc/gemm_x3.hip gemm_h2q_kernel; round 5) -----------------
def split_h2(x, scale=1.0):
    """fp32 array -> (hi, lo) float16 arrays of scale * x: hi = half(clamp(scale x)), lo = half(scale x - hi); numpy's float32 ->
    float16 cast rounds to nearest even like the device's v_cvt_f16_f32."""
    v = (np.asarray(x, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    v = np.clip(v, np.float32(-65504.0), np.float32(65504.0))
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def image_h2(x, scale=1.0):
    """x: fp32 [rows, K] -> uint16 [ceil(rows/32), ceil(K/16), 2, 32, 2, 8]: the bytes yt8m_h2_split writes for its `plain` output."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    R, K = x.shape
    RG, KB = (R + 31) // 32, (K + 15) // 16
    pad = np.zeros((RG * 32, KB * 16), dtype=np.float32)
    pad[:R, :K] = x
    hi, lo = split_h2(pad, scale)
    h = np.stack([hi.view(np.uint16), lo.view(np.uint16)])        # [2, RG*32, KB*16]
    h = h.reshape(2, RG, 32, KB, 2, 8).transpose(1, 3, 0, 2, 4, 5)
    out = np.empty_like(h)
    sw = (np.arange(32) >> 3) & 1
    for r in range(32):
        out[:, :, :, r, sw[r], :] = h[:, :, :, r, 0, :]
        out[:, :, :, r, sw[r] ^ 1, :] = h[:, :, :, r, 1, :]
    return np.ascontiguousarray(out)


def three_products(A, B, sa=1.0, sb=1.0):
    """A [M, K], B [N, K] fp32 -> fp64 [M, N]: (hi hi + hi lo + lo hi) / (sa sb) of the h2 splits, accumulated in fp64."""
    ah, al = (t.astype(np.float64) for t in split_h2(A, sa))
    bh, bl = (t.astype(np.float64) for t in split_h2(B, sb))
    return (ah @ bh.T + ah @ bl.T + al @ bh.T) / (float(sa) * float(sb))
