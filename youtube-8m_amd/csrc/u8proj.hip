// u8proj.hip -- "readers.py uint8 -> float dequantise folded into the first GEMM" for the hoisted input projection of the
// recurrent models (W/readers.py:178-187, W/utils.py:23-38, W/all_feature_transform/default_transformer.py:4-8 ->
// W/all_frame_models/lstm_model.py:34-47).
//
//   x_f = r_f (alpha q_f + c0),  alpha = 4/255, c0 = 4/512 - 2,  r_f = 1 / max(||alpha q_f + c0||, sqrt(eps))  (0 for padding frames)
//       = r_f (alpha (q_f - 128) + beta),  beta = 128 alpha + c0
//   x . W = r (.) ((q - 128) . (alpha W) + beta colsum(W))
// (q - 128) in [-128, 127] is EXACT in bf16; alpha W (one fp32 rounding) is split into three bf16 terms W1 + W2 + W3 (24
// mantissa bits: exact up to 2^-26), so every product of the bf16 MFMA is exact and the fp32 accumulation differs from the
// fp32 GEMM only by summation order -- fp32-class results at 16/3 of the fp32 matrix rate, from 1 byte instead of 4 per input
// element.  The three terms ride one NT product with the reduction concatenated: A' = [Q Q Q] ([rows, 3 D] bf16, written by the
// conversion kernel), B' = [W1; W2; W3]^T ([4H, 3 D] bf16).  The affine remainder is a row-scale + rank-1 epilogue.
#include "common.h"

namespace {

__device__ __forceinline__ uint16_t bf16_rn(float v) {          // round to nearest even (no NaN inputs here)
  uint32_t u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_val(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// One wave per frame row of raw uint8 [B,F,D] (same arithmetic and summation order as dequant_l2norm_kernel, so r and x are
// bit-identical to the float path); output rows are TIME-major (f * B + b): what the recurrence and the hoisted GEMMs use.
// HALF: the one-plane image holds (q - 128) as IEEE half instead of bfloat16 (exact either way; the h1x2 product of round 5 reads half)
__device__ __forceinline__ uint32_t small_int_bits(int v, bool half) {
  return half ? (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)(float)v) : (__float_as_uint((float)v) >> 16);
}
template <bool HALF>
__global__ __launch_bounds__(256) void u8_frames_tm_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nf,
                                                           uint16_t* __restrict__ Qb, long long ldq, int copies, float* __restrict__ xtm,
                                                           float* __restrict__ rout, int B, int F, int D, float eps,
                                                           uint16_t* __restrict__ img) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * F) return;
  const int b = (int)(row / F), f = (int)(row - (long long)b * F);
  const long long orow = (long long)f * B + b;
  const bool live = nf ? (f < nf[b]) : true;
  const uint8_t* qr = q + row * D;
  const float s = 4.0f / 255.0f, bias = 4.0f / 512.0f - 2.0f;
  const int nd = D >> 2;                                 // D % 4 == 0, D <= 2048 (checked by the host)
  uint32_t w[8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c4 = lane + 64 * it;
    w[it] = (live && c4 < nd) ? reinterpret_cast<const uint32_t*>(qr)[c4] : 0x80808080u;     // q - 128 = 0 for padding rows
    if (live && c4 < nd) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float v = fmaf((float)((w[it] >> (8 * k)) & 255u), s, bias); ss += v * v; }
    }
  }
  ss = wave_sum(ss);
  const float r = live ? rsqrtf(fmaxf(ss, eps)) : 0.f;
  if (lane == 0) rout[orow] = r;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int c4 = lane + 64 * it;
    if (c4 >= nd) continue;
    uint16_t h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = (uint16_t)small_int_bits((int)((w[it] >> (8 * k)) & 255u) - 128, HALF);   // exact
    const uint2 pk = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    if (img) {
      // one-plane x3 image (csrc/gemm_x3.hip): [orow / 32][D / 16][32 rows][2 halves][8] bf16, half h of row r in slot h ^ ((r >> 3) & 1)
      const int kb = c4 >> 2, r32 = (int)(orow & 31);
      const int slot = ((c4 >> 1) & 1) ^ ((r32 >> 3) & 1);
      uint16_t* blk = img + ((orow >> 5) * (long long)(D >> 4) + kb) * 512 + r32 * 16 + slot * 8 + (c4 & 1) * 4;
      *reinterpret_cast<uint2*>(blk) = pk;
    }
    for (int j = 0; j < copies; ++j) reinterpret_cast<uint2*>(Qb + orow * ldq + (long long)j * D)[c4] = pk;
    if (xtm) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        o.x = fmaf((float)(w[it] & 255u), s, bias) * r;
        o.y = fmaf((float)((w[it] >> 8) & 255u), s, bias) * r;
        o.z = fmaf((float)((w[it] >> 16) & 255u), s, bias) * r;
        o.w = fmaf((float)(w[it] >> 24), s, bias) * r;
      }
      reinterpret_cast<float4*>(xtm + orow * D)[c4] = o;
    }
  }
}

// (q - 128)^T as a ONE-plane operand image of the x3 kernel: rows = features d (D of them), K = time-major frame rows
// m = f * B + b.  [d / 32][m / 16][32 rows][2 halves][8] bf16, half h of row r in slot h ^ ((r >> 3) & 1) -- the A operand of the
// layer-0 weight-gradient product (yt8m_gemm_x1x3_nt_ex).  One workgroup per 16-wide K block: its 16 source rows (16 videos of one
// frame index when B % 16 == 0) are staged in LDS and leave as 16-byte pieces (8 consecutive m of one feature).
template <bool HALF>
__global__ __launch_bounds__(256) void u8_frames_image_t_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nf,
                                                                uint16_t* __restrict__ img, int B, int F, int D, int KB) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tile[];       // [16][D]
  const int kb = blockIdx.x;
  const long long MB = (long long)B * F;
  const int nd = D >> 2;
  for (int i = threadIdx.x; i < 16 * nd; i += 256) {
    const int j = i / nd, c4 = i - j * nd;
    const long long m = (long long)kb * 16 + j;
    uint32_t w = 0x80808080u;                                          // q - 128 = 0: padding frames and the K tail
    if (m < MB) {
      const int f = (int)(m / B), b = (int)(m - (long long)f * B);
      if (!nf || f < nf[b]) w = reinterpret_cast<const uint32_t*>(q + ((long long)b * F + f) * D)[c4];
    }
    reinterpret_cast<uint32_t*>(tile + (size_t)j * D)[c4] = w;
  }
  __syncthreads();
  const int Dp = (D + 31) & ~31;
  for (int i = threadIdx.x; i < 2 * Dp; i += 256) {
    const int d = i >> 1, half = i & 1;
    uint32_t pk[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t h2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int v = d < D ? (int)tile[(size_t)(half * 8 + 2 * e + u) * D + d] - 128 : 0;
        h2[u] = small_int_bits(v, HALF);                               // exact: |v| <= 128
      }
      pk[e] = h2[0] | (h2[1] << 16);
    }
    const int r32 = d & 31, slot = half ^ ((r32 >> 3) & 1);
    uint16_t* dst = img + ((long long)(d >> 5) * KB + kb) * 512 + r32 * 16 + slot * 8;
    *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
}

// W [K, ldw] fp32 -> out [N, ldo] bf16 with out[n][j K + k] = term j of the 3-way bf16 split of scale * W[k][n]  (64 x 64 tiles
// through LDS: reads coalesced along n, writes along k)
__global__ __launch_bounds__(256) void split3_bf16_t_kernel(const float* __restrict__ W, long long ldw, int K, int N, float scale,
                                                            uint16_t* __restrict__ out, long long ldo) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int k = k0 + i, n = n0 + tx;
    tile[i][tx] = (k < K && n < N) ? W[(long long)k * ldw + n] * scale : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int n = n0 + i, k = k0 + tx;
    if (n >= N || k >= K) continue;
    const float t = tile[tx][i];
    const uint16_t h1 = bf16_rn(t);
    const float r1 = t - bf16_val(h1);
    const uint16_t h2 = bf16_rn(r1);
    const uint16_t h3 = bf16_rn(r1 - bf16_val(h2));
    uint16_t* o = out + (long long)n * ldo + k;
    o[0] = h1; o[K] = h2; o[2 * (long long)K] = h3;
  }
}

// z[m][n] = r[m] * (z[m][n] + beta * cs[n]) + bias[n], in place, float4 along n
__global__ __launch_bounds__(256) void rowscale_bias_kernel(float* __restrict__ z, long long M, int N4, long long ldz, const float* __restrict__ r,
                                                            const float* __restrict__ cs, float beta, const float* __restrict__ bias) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= M * N4) return;
  const long long m = e / N4;
  const int c = (int)(e - m * N4) * 4;
  float4* p = reinterpret_cast<float4*>(z + m * ldz + c);
  const float4 c4 = *reinterpret_cast<const float4*>(cs + c);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float rr = r[m];
  float4 v = *p;
  v.x = rr * (v.x + beta * c4.x) + b4.x;
  v.y = rr * (v.y + beta * c4.y) + b4.y;
  v.z = rr * (v.z + beta * c4.z) + b4.z;
  v.w = rr * (v.w + beta * c4.w) + b4.w;
  *p = v;
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_u8_proj_supported(int64_t D) { return (D >= 4 && D <= 2048 && (D % 8) == 0) ? 1 : 0; }

extern "C" int yt8m_u8_frames_to_bf16_tm(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps,
                                         int copies, void* Qb, int64_t ldq, float* x_tm, float* r_out, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_u8_proj_supported(D), YT8M_E_SHAPE, "D must be a multiple of 8 and <= 2048");
  YT8M_REQUIRE(copies >= 1 && copies <= 4 && ldq >= (int64_t)copies * D && (ldq % 4) == 0, YT8M_E_SHAPE, "bad copies / ldq");
  YT8M_REQUIRE(q && Qb && r_out, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(q) & 3) | (reinterpret_cast<uintptr_t>(Qb) & 7) |
                (x_tm ? reinterpret_cast<uintptr_t>(x_tm) & 15 : 0)) == 0, YT8M_E_BADARG, "misaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frames_tm_kernel<false>, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, q, num_frames, static_cast<uint16_t*>(Qb),
                     (long long)ldq, copies, x_tm, r_out, (int)B, (int)F, (int)D, eps, (uint16_t*)nullptr);
  return launch_status("u8_frames_tm_kernel");
}

// Same pass, but (q - 128) goes into the ONE-plane operand image of yt8m_gemm_x1x3_nt (rows time-major: f * B + b;
// ceil(B F / 32) * (D / 16) KiB, 16-byte aligned; D % 16 == 0) instead of the K-concatenated bf16 copies.
extern "C" int yt8m_u8_frames_image(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, void* image,
                                    float* x_tm, float* r_out, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_u8_proj_supported(D) && (D % 16) == 0, YT8M_E_SHAPE, "D must be a multiple of 16 and <= 2048");
  YT8M_REQUIRE(q && image && r_out, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(q) & 3) | (reinterpret_cast<uintptr_t>(image) & 15) |
                (x_tm ? reinterpret_cast<uintptr_t>(x_tm) & 15 : 0)) == 0, YT8M_E_BADARG, "misaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frames_tm_kernel<false>, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, q, num_frames, (uint16_t*)nullptr, 0LL, 0,
                     x_tm, r_out, (int)B, (int)F, (int)D, eps, static_cast<uint16_t*>(image));
  return launch_status("u8_frames_tm_kernel");
}

// (q - 128)^T as a one-plane image: ceil(D / 32) * ceil(B F / 16) KiB, 16-byte aligned; K runs over the time-major rows f * B + b.
extern "C" int yt8m_u8_frames_image_t(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, void* image_t,
                                      yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(D <= 2048 && (D % 4) == 0, YT8M_E_SHAPE, "D must be a multiple of 4 and <= 2048");
  YT8M_REQUIRE(q && image_t, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(q) & 3) | (reinterpret_cast<uintptr_t>(image_t) & 15)) == 0, YT8M_E_BADARG, "misaligned operand");
  YT8M_REQUIRE(B * F < (1LL << 31) - 16, YT8M_E_SHAPE, "too many frame rows");
  hipStream_t s = as_stream(stream);
  const int KB = (int)((B * F + 15) / 16);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frames_image_t_kernel<false>, dim3((unsigned)KB), dim3(256), (size_t)16 * D, s, q, num_frames,
                     static_cast<uint16_t*>(image_t), (int)B, (int)F, (int)D, KB);
  return launch_status("u8_frames_image_t_kernel");
}

// The same two images with (q - 128) as IEEE half (operands of yt8m_gemm_h1x2_nt_ex).
extern "C" int yt8m_u8_frames_image_f16(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, void* image,
                                    float* x_tm, float* r_out, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_u8_proj_supported(D) && (D % 16) == 0, YT8M_E_SHAPE, "D must be a multiple of 16 and <= 2048");
  YT8M_REQUIRE(q && image && r_out, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(q) & 3) | (reinterpret_cast<uintptr_t>(image) & 15) |
                (x_tm ? reinterpret_cast<uintptr_t>(x_tm) & 15 : 0)) == 0, YT8M_E_BADARG, "misaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frames_tm_kernel<true>, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, q, num_frames, (uint16_t*)nullptr, 0LL, 0,
                     x_tm, r_out, (int)B, (int)F, (int)D, eps, static_cast<uint16_t*>(image));
  return launch_status("u8_frames_tm_kernel");
}

// (q - 128)^T as a one-plane image: ceil(D / 32) * ceil(B F / 16) KiB, 16-byte aligned; K runs over the time-major rows f * B + b.
extern "C" int yt8m_u8_frames_image_t_f16(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, void* image_t,
                                      yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(D <= 2048 && (D % 4) == 0, YT8M_E_SHAPE, "D must be a multiple of 4 and <= 2048");
  YT8M_REQUIRE(q && image_t, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(q) & 3) | (reinterpret_cast<uintptr_t>(image_t) & 15)) == 0, YT8M_E_BADARG, "misaligned operand");
  YT8M_REQUIRE(B * F < (1LL << 31) - 16, YT8M_E_SHAPE, "too many frame rows");
  hipStream_t s = as_stream(stream);
  const int KB = (int)((B * F + 15) / 16);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frames_image_t_kernel<true>, dim3((unsigned)KB), dim3(256), (size_t)16 * D, s, q, num_frames,
                     static_cast<uint16_t*>(image_t), (int)B, (int)F, (int)D, KB);
  return launch_status("u8_frames_image_t_kernel");
}

extern "C" int yt8m_split3_bf16_t(const float* W, int64_t ldw, int64_t K, int64_t N, float scale, void* out, int64_t ldo,
                                  yt8m_stream_t stream) {
  YT8M_REQUIRE(K >= 0 && N >= 0 && K < (1 << 24) && N < (1 << 24), YT8M_E_SHAPE, "bad dimension");
  if (K * N == 0) return YT8M_OK;
  YT8M_REQUIRE(W && out && ldw >= N && ldo >= 3 * K, YT8M_E_BADARG, "bad operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(split3_bf16_t_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, s, W, (long long)ldw,
                     (int)K, (int)N, scale, static_cast<uint16_t*>(out), (long long)ldo);
  return launch_status("split3_bf16_t_kernel");
}

extern "C" int yt8m_rowscale_bias_f32(float* z, int64_t M, int64_t N, int64_t ldz, const float* r, const float* cs, float beta,
                                      const float* bias, yt8m_stream_t stream) {
  YT8M_REQUIRE(M >= 0 && N >= 0, YT8M_E_SHAPE, "negative dimension");
  if (M * N == 0) return YT8M_OK;
  YT8M_REQUIRE(z && r && cs, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((N % 4) == 0 && (ldz % 4) == 0 && ldz >= N, YT8M_E_SHAPE, "N and ldz must be multiples of 4");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const long long n = M * (N / 4);
  hipLaunchKernelGGL(rowscale_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, (long long)M, (int)(N / 4), (long long)ldz,
                     r, cs, beta, bias);
  return launch_status("rowscale_bias_kernel");
}
