// x3_image.h -- the operand-image block format of csrc/gemm_x3.hip, shared by everything that WRITES images: the split pass
// (gemm_x3.hip), the optimiser pass that keeps the images of the weight matrices current (optim.hip, round 5).
//   image: [rows / 32][K / 16][plane 0..NP-1][32 rows][2 halves][8] bf16 -- one 1 KiB block per 32 rows, 16-wide K block and plane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace yt8m_x3 {

constexpr int RG_F = 256;                             // floats of one image block: 32 rows x 32 B (1 KiB)

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {        // round to nearest even, finite x
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  h1 = bf16_rn_bits(x);
  const float r1 = x - __uint_as_float(h1 << 16);                   // exact
  h2 = bf16_rn_bits(r1);
  const float r2 = r1 - __uint_as_float(h2 << 16);                  // exact
  h3 = bf16_rn_bits(r2);
  if ((__float_as_uint(x) & 0x7F800000u) == 0x7F800000u) {          // inf / nan stay in the leading term only
    h1 = __float_as_uint(x) >> 16;
    h2 = h3 = 0;
  }
}
// 16 values of one K block of image row `row` -> its two 16-byte halves in each of the three plane blocks at dst (the block of
// plane 0; half h sits in slot h ^ ((row >> 3) & 1))
template <int NP = 3>
__device__ __forceinline__ void store_block(const float (&v)[16], float* __restrict__ dst, int row) {
  const int r = row & 31, sw = (r >> 3) & 1;
  dst += r * 8;
  unsigned h[3][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) split3(v[j], h[0][j], h[1][j], h[2][j]);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    uint4 lo, hi;
    lo.x = h[p][0] | (h[p][1] << 16);  lo.y = h[p][2] | (h[p][3] << 16);
    lo.z = h[p][4] | (h[p][5] << 16);  lo.w = h[p][6] | (h[p][7] << 16);
    hi.x = h[p][8] | (h[p][9] << 16);  hi.y = h[p][10] | (h[p][11] << 16);
    hi.z = h[p][12] | (h[p][13] << 16); hi.w = h[p][14] | (h[p][15] << 16);
    *reinterpret_cast<uint4*>(dst + p * RG_F + 4 * sw) = lo;
    *reinterpret_cast<uint4*>(dst + p * RG_F + 4 * (sw ^ 1)) = hi;
  }
}


// ---- "h2" images (round 5): TWO fp16 planes, a S = hi + lo ---------------------------------------------------------------------------
// The same block layout with NP = 2 planes of IEEE half instead of bfloat16: hi = f16(a S), lo = f16(a S - hi) for a power-of-two
// scale S that brings the operand's magnitude into the half range (|a S| is clamped to the largest half: an operand that outgrew its
// scale degrades, it does not turn into inf; NaN stays NaN).  11 + 11 significand bits: a S = hi + lo + r, |r| <= 2^-23 |a S| while lo is a normal
// half (|a S| >= 2^-3), an absolute 2^-25 below that.  Three products (hi hi, hi lo, lo hi) then carry a b to 2^-21 relative --
// the size of three fp32 roundings -- at HALF the matrix-pipe time of the six-product bf16 split (csrc/gemm_x3.hip gemm_h2q_kernel).
__device__ __forceinline__ void split_h2(float x, unsigned& h1, unsigned& h2) {
  x = (x != x) ? x : fminf(fmaxf(x, -65504.f), 65504.f);            // finite and infinite values clamp; a NaN stays a NaN (hi = NaN, lo = NaN):
  const _Float16 hi = (_Float16)x;                                  // a diverged step shows up as NaN downstream, as it would in fp32
  const _Float16 lo = (_Float16)(x - (float)hi);                    // exact difference, rounded once
  h1 = (unsigned)__builtin_bit_cast(unsigned short, hi);
  h2 = (unsigned)__builtin_bit_cast(unsigned short, lo);
}
__device__ __forceinline__ void store_block_h2(const float (&v)[16], float* __restrict__ dst, int row) {
  const int r = row & 31, sw = (r >> 3) & 1;
  dst += r * 8;
  unsigned h[2][16];
#pragma unroll
  for (int j = 0; j < 16; ++j) split_h2(v[j], h[0][j], h[1][j]);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    uint4 lo, hi;
    lo.x = h[p][0] | (h[p][1] << 16);  lo.y = h[p][2] | (h[p][3] << 16);
    lo.z = h[p][4] | (h[p][5] << 16);  lo.w = h[p][6] | (h[p][7] << 16);
    hi.x = h[p][8] | (h[p][9] << 16);  hi.y = h[p][10] | (h[p][11] << 16);
    hi.z = h[p][12] | (h[p][13] << 16); hi.w = h[p][14] | (h[p][15] << 16);
    *reinterpret_cast<uint4*>(dst + p * RG_F + 4 * sw) = lo;
    *reinterpret_cast<uint4*>(dst + p * RG_F + 4 * (sw ^ 1)) = hi;
  }
}
// power-of-two scale that brings a magnitude m just below 2^target (vanishing / non-finite operands keep scale 1)
__device__ __forceinline__ float pow2_scale_for(float m, int target) {
  if (!(m > 7.9e-31f) || !(m < 3.0e38f)) return 1.f;
  int e;
  frexpf(m, &e);
  return ldexpf(1.f, target - e);
}

}  // namespace yt8m_x3
