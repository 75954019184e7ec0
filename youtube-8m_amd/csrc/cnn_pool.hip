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

// one thread = four consecutive columns of one video; rows t B + b are N floats apart per video and B N per frame: coalesced along n
__global__ __launch_bounds__(256) void timepool_max_kernel(const float* __restrict__ y, int F, int B, int N, int64_t ldy,
                                                           float* __restrict__ out, int32_t* __restrict__ idx, int64_t ldo) {
  const int n4 = N >> 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * n4) return;
  const int b = (int)(e / n4), c = (int)(e - (int64_t)b * n4) * 4;
  const float* p = y + (int64_t)b * ldy + c;
  float4 m = *reinterpret_cast<const float4*>(p);
  int4 at = make_int4(0, 0, 0, 0);
  for (int t = 1; t < F; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)t * B * ldy);
    if (v.x > m.x) { m.x = v.x; at.x = t; }                           // strict: the FIRST frame that attains the maximum
    if (v.y > m.y) { m.y = v.y; at.y = t; }
    if (v.z > m.z) { m.z = v.z; at.z = t; }
    if (v.w > m.w) { m.w = v.w; at.w = t; }
  }
  *reinterpret_cast<float4*>(out + (int64_t)b * ldo + c) = m;
  *reinterpret_cast<int4*>(idx + (int64_t)b * ldo + c) = at;
}

// block = (column n, shift i); thread = four consecutive features (uchar4 of a frame row), up to two groups per thread (D <= 2048).
// The B (coefficient, frame row) pairs of the column go through LDS first: the gather loop then has no dependent address chain.
__global__ __launch_bounds__(256) void u8_cnn_pool_dw_kernel(const uint8_t* __restrict__ q, const float* __restrict__ r_tm,
                                                             const int32_t* __restrict__ idx, const float* __restrict__ g, int64_t ldg,
                                                             int B, int F, int D, int N, float* __restrict__ dW, float beta) {
  __shared__ float s_coef[256];
  __shared__ int s_row[256];
  const int n = blockIdx.x, i = blockIdx.y;
  const int d4 = D >> 2;
  const int t0 = threadIdx.x, t1 = threadIdx.x + 256;
  const bool h0 = t0 < d4, h1 = t1 < d4;
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                                                   // sum of the coefficients: the affine remainder
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + (int)threadIdx.x;
    float coef = 0.f;
    int row = 0;
    if (b < B) {
      const int t = idx[(int64_t)b * ldg + n] - i;                    // the frame this shift read at the video's argmax
      if (t >= 0) {                                                   // (r = 0 on a padding frame: its bytes count for nothing)
        coef = g[(int64_t)b * ldg + n] * r_tm[(int64_t)t * B + b];
        row = b * F + t;
      }
    }
    s_coef[threadIdx.x] = coef;
    s_row[threadIdx.x] = row;
    __syncthreads();
    const int nb = min(256, B - b0);
#pragma unroll 4
    for (int j = 0; j < nb; ++j) {
      const float c = s_coef[j];
      const uint32_t* rp = reinterpret_cast<const uint32_t*>(q + (int64_t)s_row[j] * D);
      csum += c;
      if (h0) {
        const uint32_t u = rp[t0];
        a0[0] += c * (float)(u & 255u); a0[1] += c * (float)((u >> 8) & 255u);
        a0[2] += c * (float)((u >> 16) & 255u); a0[3] += c * (float)(u >> 24);
      }
      if (h1) {
        const uint32_t u = rp[t1];
        a1[0] += c * (float)(u & 255u); a1[1] += c * (float)((u >> 8) & 255u);
        a1[2] += c * (float)((u >> 16) & 255u); a1[3] += c * (float)(u >> 24);
      }
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
  if (h1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* o = col + (int64_t)(4 * t1 + k) * N;
      const float v = DQ_ALPHA * a1[k] + rem;
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
  const int64_t n = B * (N / 4);
  hipLaunchKernelGGL(timepool_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, (int)F, (int)B, (int)N, ldy, out, idx, ldo);
  return launch_status("timepool_max_kernel");
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
  hipLaunchKernelGGL(u8_cnn_pool_dw_kernel, dim3((unsigned)N, (unsigned)fs), dim3(256), 0, s, q, r_tm, idx, g, ldg, (int)B, (int)F, (int)D,
                     (int)N, dW, beta);
  return launch_status("u8_cnn_pool_dw_kernel");
}
