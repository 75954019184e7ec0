// cnn_pool.hip -- the max-pooled "einsum CNN" of W/all_frame_models/cnn_deep_combine_chain_model.py:60-82,100-106 on the reader's raw
// uint8 frames (gfx950): what is left once the dense products are yt8m_gemm_h1x2_nt_ex launches on the byte image (youtube-8m_amd/seq_ops.py
// _PooledCnnU8).
//   yt8m_timepool_max_f32 : cnn_output in TIME-major rows y[t B + b, n] -> tf.reduce_max over the frames, and WHICH frame it was.
//   yt8m_u8_cnn_pool_dw   : the filters' gradient.  d(loss)/d(cnn_output) is non-zero at ONE frame per (video, column) -- the argmax --
//                           so x^T dy is not a [D, F B] x [F B, N] product (102 GFLOP per CNN at BASELINE shapes) but B gathered frame rows
//                           per output column (0.34 GFLOP): dW[i D + d, n] (+)= sum_b g[b, n] x[t*(b, n) - i, b, d], x dequantised and
//                           l2-normalised on the fly from the bytes (W/utils.py:23-38, default_transformer.py:4-8).
// Bound: both are streams -- the pooling reads y once (4 F B N bytes at the HBM rate), the gradient gathers B N fs rows of D bytes from L2.
#include "common.h"

namespace {

constexpr float DQ_ALPHA = 4.0f / 255.0f;                              // x = r (alpha (q - 128) + beta')
constexpr float DQ_BETA = 128.0f * 4.0f / 255.0f + (4.0f / 512.0f - 2.0f);

// The pooling over the per-shift partial outputs z[t B + b, zbase_k + i N_k + n] = x[t, b] . W_k[i D : (i + 1) D][:, n] of ONE product
// for the whole CNN: cnn_output[t, b, k, n] = sum_i z[(t - i) B + b, .] (i ascending: the order the in-place accumulation of the
// per-shift products uses), maximum over t and its first frame.  Every element of z is read once.  (A column group of four never
// straddles two filters: ncol[k] % 4 == 0.)
struct PoolDesc {
  int nfilt;
  int fs[8], ncol[8], zbase[8], obase[8];
};
// block = (64 output columns, video): 16 column groups of four x 16 frame classes (t mod 16); a class keeps the first frame of its own
// maximum, the classes are merged through LDS (larger value, the earlier frame on a tie) -- 300 frames on one thread per column group
// left three quarters of the chip idle.
__global__ __launch_bounds__(256) void timepool_shiftmax_kernel(const float* __restrict__ z, int F, int B, int64_t ldz, PoolDesc d, int Ntot,
                                                                float* __restrict__ out, int32_t* __restrict__ idx, int64_t ldo) {
  __shared__ float4 s_m[16][16];
  __shared__ int4 s_t[16][16];
  const int cg = threadIdx.x & 15, tg = threadIdx.x >> 4;
  const int b = blockIdx.y, c = blockIdx.x * 64 + 4 * cg;
  const bool live = c < Ntot;
  int k = 0;
  while (k + 1 < d.nfilt && c >= d.obase[k + 1]) ++k;
  const int fs = d.fs[k], nk = d.ncol[k];
  const float* p = z + (int64_t)b * ldz + d.zbase[k] + (c - d.obase[k]);
  const float NEG = -3.402823466e38f;
  float4 m = make_float4(NEG, NEG, NEG, NEG);
  int4 at = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
  if (live) {
    for (int t = tg; t < F; t += 16) {
      float4 v = *reinterpret_cast<const float4*>(p + (int64_t)t * B * ldz);
      for (int i = 1; i < fs && i <= t; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(p + (int64_t)(t - i) * B * ldz + (int64_t)i * nk);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (v.x > m.x || at.x == 0x7fffffff) { m.x = v.x; at.x = t; }   // (the first frame of the class always enters: NaN-free data or not)
      if (v.y > m.y || at.y == 0x7fffffff) { m.y = v.y; at.y = t; }
      if (v.z > m.z || at.z == 0x7fffffff) { m.z = v.z; at.z = t; }
      if (v.w > m.w || at.w == 0x7fffffff) { m.w = v.w; at.w = t; }
    }
  }
  s_m[tg][cg] = m;
  s_t[tg][cg] = at;
  __syncthreads();
  if (tg == 0 && live) {
#pragma unroll
    for (int j = 1; j < 16; ++j) {
      const float4 v = s_m[j][cg];
      const int4 w = s_t[j][cg];
      if (w.x != 0x7fffffff && (v.x > m.x || (v.x == m.x && w.x < at.x))) { m.x = v.x; at.x = w.x; }
      if (w.y != 0x7fffffff && (v.y > m.y || (v.y == m.y && w.y < at.y))) { m.y = v.y; at.y = w.y; }
      if (w.z != 0x7fffffff && (v.z > m.z || (v.z == m.z && w.z < at.z))) { m.z = v.z; at.z = w.z; }
      if (w.w != 0x7fffffff && (v.w > m.w || (v.w == m.w && w.w < at.w))) { m.w = v.w; at.w = w.w; }
    }
    *reinterpret_cast<float4*>(out + (int64_t)b * ldo + c) = m;
    *reinterpret_cast<int4*>(idx + (int64_t)b * ldo + c) = at;
  }
}

// block = (column n, shift i); thread = four consecutive features (uchar4 of a frame row; D / 4 threads rounded up to whole waves).
// The B (coefficient, frame row) pairs of the column go through LDS first: the gather loop then has no dependent address chain.
__global__ __launch_bounds__(512) void u8_cnn_pool_dw_kernel(const uint8_t* __restrict__ q, const float* __restrict__ r_tm,
                                                             const int32_t* __restrict__ idx, const float* __restrict__ g, int64_t ldg,
                                                             int B, int F, int D, int N, float* __restrict__ dW, float beta) {
  __shared__ float s_coef[512];
  __shared__ int s_row[512];
  const int n = blockIdx.x, i = blockIdx.y;
  const int d4 = D >> 2, nt = (int)blockDim.x;
  const int t0 = threadIdx.x;
  const bool h0 = t0 < d4;
  float a0[4] = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                                                   // sum of the coefficients: the affine remainder
  for (int b0 = 0; b0 < B; b0 += nt) {
    const int b = b0 + t0;
    float coef = 0.f;
    int row = 0;
    if (b < B) {
      const int t = idx[(int64_t)b * ldg + n] - i;                    // the frame this shift read at the video's argmax
      if (t >= 0) {                                                   // (r = 0 on a padding frame: its bytes count for nothing)
        coef = g[(int64_t)b * ldg + n] * r_tm[(int64_t)t * B + b];
        row = b * F + t;
      }
    }
    s_coef[t0] = coef;
    s_row[t0] = row;
    __syncthreads();
    const int nb = min(nt, B - b0);
    int j = 0;
    for (; j + 8 <= nb; j += 8) {                                     // eight independent row requests in flight per thread
      uint32_t u[8];
      float c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        c[k] = s_coef[j + k];
        u[k] = h0 ? reinterpret_cast<const uint32_t*>(q + (int64_t)s_row[j + k] * D)[t0] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        csum += c[k];
        a0[0] += c[k] * (float)(u[k] & 255u); a0[1] += c[k] * (float)((u[k] >> 8) & 255u);
        a0[2] += c[k] * (float)((u[k] >> 16) & 255u); a0[3] += c[k] * (float)(u[k] >> 24);
      }
    }
    for (; j < nb; ++j) {
      const float c = s_coef[j];
      const uint32_t u = h0 ? reinterpret_cast<const uint32_t*>(q + (int64_t)s_row[j] * D)[t0] : 0u;
      csum += c;
      a0[0] += c * (float)(u & 255u); a0[1] += c * (float)((u >> 8) & 255u);
      a0[2] += c * (float)((u >> 16) & 255u); a0[3] += c * (float)(u >> 24);
    }
    __syncthreads();
  }
  // sum coef (alpha (q - 128) + beta') = alpha sum coef q + (beta' - 128 alpha) sum coef
  const float rem = (DQ_BETA - 128.0f * DQ_ALPHA) * csum;
  float* col = dW + (int64_t)i * D * N + n;
  if (h0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* o = col + (int64_t)(4 * t0 + k) * N;
      const float v = DQ_ALPHA * a0[k] + rem;
      *o = beta != 0.f ? beta * *o + v : v;
    }
  }
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_timepool_max_f32(const float* y, int64_t F, int64_t B, int64_t N, int64_t ldy, float* out, int32_t* idx, int64_t ldo,
                                     yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 1 && B >= 0 && N >= 0, YT8M_E_SHAPE, "bad dimension");
  if (B * N == 0) return YT8M_OK;
  YT8M_REQUIRE((N % 4) == 0 && (ldy % 4) == 0 && (ldo % 4) == 0 && ldy >= N && ldo >= N, YT8M_E_SHAPE, "N, ldy, ldo must be multiples of 4");
  YT8M_REQUIRE(y && out && idx, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0,
               YT8M_E_BADARG, "operands must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  PoolDesc d;                                                         // one "filter" of length 1: the plain maximum over the frames
  d.nfilt = 1;
  for (int k = 0; k < 8; ++k) { d.fs[k] = 1; d.ncol[k] = (int)N; d.zbase[k] = 0; d.obase[k] = 0; }
  hipLaunchKernelGGL(timepool_shiftmax_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)B), dim3(256), 0, s, y, (int)F, (int)B, ldy, d, (int)N, out,
                     idx, ldo);
  return launch_status("timepool_shiftmax_kernel");
}

extern "C" int yt8m_timepool_shiftmax_f32(const float* z, int64_t F, int64_t B, int64_t ldz, int nfilt, const int32_t* fs, const int32_t* ncol,
                                          float* out, int32_t* idx, int64_t ldo, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 1 && B >= 0 && nfilt >= 1 && nfilt <= 8 && fs && ncol, YT8M_E_SHAPE, "1..8 filters");
  PoolDesc d;
  d.nfilt = nfilt;
  int zb = 0, ob = 0;
  for (int k = 0; k < nfilt; ++k) {
    YT8M_REQUIRE(fs[k] >= 1 && ncol[k] >= 4 && (ncol[k] % 4) == 0, YT8M_E_SHAPE, "filter columns must be multiples of 4");
    d.fs[k] = fs[k]; d.ncol[k] = ncol[k]; d.zbase[k] = zb; d.obase[k] = ob;
    zb += fs[k] * ncol[k];
    ob += ncol[k];
  }
  for (int k = nfilt; k < 8; ++k) { d.fs[k] = 1; d.ncol[k] = 4; d.zbase[k] = zb; d.obase[k] = ob; }
  if (B == 0) return YT8M_OK;
  YT8M_REQUIRE((ldz % 4) == 0 && (ldo % 4) == 0 && ldz >= zb && ldo >= ob, YT8M_E_SHAPE, "ldz / ldo too small or not multiples of 4");
  YT8M_REQUIRE(z && out && idx, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0,
               YT8M_E_BADARG, "operands must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(timepool_shiftmax_kernel, dim3((unsigned)((ob + 63) / 64), (unsigned)B), dim3(256), 0, s, z, (int)F, (int)B, ldz, d, ob, out,
                     idx, ldo);
  return launch_status("timepool_shiftmax_kernel");
}

extern "C" int yt8m_u8_cnn_pool_dw(const uint8_t* q, const float* r_tm, const int32_t* idx, const float* g, int64_t ldg, int64_t B, int64_t F,
                                   int64_t D, int64_t N, int64_t fs, float* dW, float beta, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 1 && D >= 4 && N >= 0 && fs >= 1 && fs <= 16, YT8M_E_SHAPE, "bad dimension");
  if (B * N == 0) return YT8M_OK;
  YT8M_REQUIRE((D % 4) == 0 && D <= 2048 && ldg >= N, YT8M_E_SHAPE, "D must be a multiple of 4 and <= 2048");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  YT8M_REQUIRE(q && r_tm && idx && g && dW, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(q) & 3) == 0, YT8M_E_BADARG, "frames must be 4-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const unsigned nt = (unsigned)(((D / 4) + 63) / 64 * 64);
  hipLaunchKernelGGL(u8_cnn_pool_dw_kernel, dim3((unsigned)N, (unsigned)fs), dim3(nt), 0, s, q, r_tm, idx, g, ldg, (int)B, (int)F, (int)D,
                     (int)N, dW, beta);
  return launch_status("u8_cnn_pool_dw_kernel");
}
