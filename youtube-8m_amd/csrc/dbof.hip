// dbof.hip -- the non-GEMM pieces of DbofModel on the device (W/all_frame_models/dbof_model.py:36-124, W/model_utils.py:23-95):
//   * frame sampling + gather (SampleRandomFrames / SampleRandomSequence): index bookkeeping in integers from a Philox uniform
//     per (video, sample) / per video, rows copied with 16-byte accesses; raw uint8 or float frames.  Only the `iterations`
//     (30) sampled frames of the 300 are ever dequantised / normalised.
//   * FramePooling over the sampled frames: max (gradient split equally between tied maxima, as tf.reduce_max's _MinOrMaxGrad
//     does -- after relu6 ties at 0 and 6 are the common case) and average.
//   * slim.batch_norm(center, scale): batch moments (two passes, fixed summation order: bitwise reproducible), moving-average
//     update with `decay`, normalise; backward with the batch-statistics terms (training) or the frozen form (inference).
// All of it is HBM-bound column / row streaming: a wave reads 64 consecutive floats per row.
#include "common.h"
#include "philox.h"

namespace {

// ---- frame sampling --------------------------------------------------------------------------------------------------------
// mode 0: SampleRandomFrames  (model_utils.py:51-70): index = int(u[b,s] * num_frames[b]),  u element b*S + s
// mode 1: SampleRandomSequence (:23-48): start = int(u[b] * (max(num_frames - S, 0) + 1)), index = min(start + s, num_frames - 1)
// Indices are clamped to [0, F-1] (a video without frames reads frame 0; TF's gather_nd would fail on -1).
template <typename T>
__global__ __launch_bounds__(256) void sample_gather_kernel(const T* __restrict__ x, const int32_t* __restrict__ nf, T* __restrict__ out,
                                                            int32_t* __restrict__ idx_out, int F, int D, int S, int mode,
                                                            unsigned long long seed) {
  const int b = blockIdx.x / S, s = blockIdx.x % S;
  const int n = nf ? nf[b] : F;
  int idx;
  if (mode == 0) {
    idx = (int)(yt8m_rng::uniform_at((unsigned long long)b * S + s, seed) * (float)n);
  } else {
    const int max_start = n - S > 0 ? n - S : 0;
    const int start = (int)(yt8m_rng::uniform_at((unsigned long long)b, seed) * (float)(max_start + 1));
    idx = start + s < n - 1 ? start + s : n - 1;
  }
  idx = idx < 0 ? 0 : (idx > F - 1 ? F - 1 : idx);
  if (threadIdx.x == 0 && idx_out) idx_out[blockIdx.x] = idx;
  const T* src = x + ((long long)b * F + idx) * D;
  T* dst = out + (long long)blockIdx.x * D;
  const long long bytes = (long long)D * sizeof(T);
  if ((bytes & 15) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < (int)(bytes >> 4); i += 256) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < D; i += 256) dst[i] = src[i];
  }
}

// ---- frame pooling -----------------------------------------------------------------------------------------------------------
// x [B,S,C] -> out [B,C]; a thread owns VEC consecutive c of one b (lanes along c: coalesced 16-byte accesses for VEC = 4); the S
// values of a column are fetched eight at a time before they are reduced (one 4-byte load in flight per thread ran the max over
// the 8 attention copies of BASELINE configs[4] at 1.6 TB/s), and the backward keeps up to 16 of them in registers so that x is
// read once (ties share the gradient, as tf.reduce_max's gradient does).
template <int VEC> struct PoolVec;
template <> struct PoolVec<4> { using T = float4; };
template <> struct PoolVec<1> { using T = float; };
__device__ __forceinline__ float4 vmax(float4 a, float4 b) { return float4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vscale(float4 a, float k) { return float4{a.x * k, a.y * k, a.z * k, a.w * k}; }
__device__ __forceinline__ float vscale(float a, float k) { return a * k; }
__device__ __forceinline__ float4 vdiv(float4 a, float k) { return float4{a.x / k, a.y / k, a.z / k, a.w / k}; }
__device__ __forceinline__ float vdiv(float a, float k) { return a / k; }
__device__ __forceinline__ float4 veq(float4 a, float4 m) { return float4{a.x == m.x ? 1.f : 0.f, a.y == m.y ? 1.f : 0.f, a.z == m.z ? 1.f : 0.f, a.w == m.w ? 1.f : 0.f}; }
__device__ __forceinline__ float veq(float a, float m) { return a == m ? 1.f : 0.f; }
__device__ __forceinline__ float4 vshare(float4 g, float4 t) { return float4{g.x / fmaxf(t.x, 1.f), g.y / fmaxf(t.y, 1.f), g.z / fmaxf(t.z, 1.f), g.w / fmaxf(t.w, 1.f)}; }
__device__ __forceinline__ float vshare(float g, float t) { return g / fmaxf(t, 1.f); }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) { return float4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ void vzero(float4& a) { a = float4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }

template <int VEC>
__global__ __launch_bounds__(256) void frame_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int S, int C, int mode) {
  using T = typename PoolVec<VEC>::T;
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC, b = blockIdx.y;
  if (c >= C) return;
  const T* p = reinterpret_cast<const T*>(x + (long long)b * S * C + c);
  const long long cs = C / VEC;                                       // elements of T between frames
  T acc = p[0];                                                       // max: seed; average: first term (added in frame order)
  for (int s0 = 1; s0 < S; s0 += 8) {
    T v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s0 + i < S ? p[(long long)(s0 + i) * cs] : acc;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (s0 + i < S) acc = mode == 0 ? vmax(acc, v[i]) : vadd(acc, v[i]);
  }
  if (mode != 0) acc = vdiv(acc, (float)S);
  *reinterpret_cast<T*>(out + (long long)b * C + c) = acc;
}

template <int VEC>
__global__ __launch_bounds__(256) void frame_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                                             const float* __restrict__ dy, float* __restrict__ dx, int S, int C, int mode) {
  using T = typename PoolVec<VEC>::T;
  const int c = (blockIdx.x * 256 + threadIdx.x) * VEC, b = blockIdx.y;
  if (c >= C) return;
  const long long base = (long long)b * S * C + c, cs = C / VEC;
  const T g = *reinterpret_cast<const T*>(dy + (long long)b * C + c);
  const T* xp = reinterpret_cast<const T*>(x + base);
  T* dp = reinterpret_cast<T*>(dx + base);
  if (mode != 0) {
    const T share = vdiv(g, (float)S);
    for (int s = 0; s < S; ++s) dp[(long long)s * cs] = share;
    return;
  }
  const T m = *reinterpret_cast<const T*>(out + (long long)b * C + c);
  T ties;
  vzero(ties);
  if (S <= 16) {
    T v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < S) v[i] = xp[(long long)i * cs];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < S) { v[i] = veq(v[i], m); ties = vadd(ties, v[i]); }
    const T share = vshare(g, ties);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < S) dp[(long long)i * cs] = vmul(v[i], share);
    return;
  }
  for (int s = 0; s < S; ++s) ties = vadd(ties, veq(xp[(long long)s * cs], m));
  const T share = vshare(g, ties);
  for (int s = 0; s < S; ++s) dp[(long long)s * cs] = vmul(veq(xp[(long long)s * cs], m), share);
}

// ---- batch norm -----------------------------------------------------------------------------------------------------------
// column statistics over N rows: workgroup = 64 columns x 4 row lanes; two passes (mean, then centred second moment)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int N, int C, float eps, float decay,
                                                       float* __restrict__ mean, float* __restrict__ rstd,
                                                       float* __restrict__ moving_mean, float* __restrict__ moving_var) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  float acc = 0.f;
  if (ok) for (int r = rl; r < N; r += 4) acc += x[(long long)r * C + c];
  red[rl][cl] = acc;
  __syncthreads();
  const float mu = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)N;
  __syncthreads();
  acc = 0.f;
  if (ok) for (int r = rl; r < N; r += 4) { const float d = x[(long long)r * C + c] - mu; acc += d * d; }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && ok) {
    const float var = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)N;    // biased, as tf.nn.moments
    mean[c] = mu;
    rstd[c] = 1.0f / sqrtf(var + eps);
    if (moving_mean) moving_mean[c] = decay * moving_mean[c] + (1.0f - decay) * mu;
    if (moving_var) moving_var[c] = decay * moving_var[c] + (1.0f - decay) * var;
  }
}

__global__ __launch_bounds__(256) void bn_frozen_stats_kernel(const float* __restrict__ moving_mean, const float* __restrict__ moving_var,
                                                              int C, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = moving_mean[c];
  rstd[c] = 1.0f / sqrtf(moving_var[c] + eps);
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, long long n, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int c = (int)(e % C);
  y[e] = (x[e] - mean[c]) * rstd[c] * gamma[c] + beta[c];
}

// per column: sum dy and sum dy * xhat (fixed order)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int C,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ sum_dy, float* __restrict__ sum_dyx) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool ok = c < C;
  float a = 0.f, b = 0.f;
  if (ok) {
    const float mu = mean[c], rs = rstd[c];
    for (int r = rl; r < N; r += 4) {
      const float g = dy[(long long)r * C + c];
      a += g;
      b += g * ((x[(long long)r * C + c] - mu) * rs);
    }
  }
  red[0][rl][cl] = a;
  red[1][rl][cl] = b;
  __syncthreads();
  if (rl == 0 && ok) {
    sum_dy[c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    sum_dyx[c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}

// training: dx = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)); frozen statistics: dx = gamma rstd dy.
// Also writes dgamma / dbeta (beta_* = 1: accumulate) from the column sums -- by the first row block only.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, long long n, int N, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                                           const float* __restrict__ sum_dyx, int training, float* __restrict__ dx) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int c = (int)(e % C);
  const float rs = rstd[c], g = dy[e];
  if (training) {
    const float xhat = (x[e] - mean[c]) * rs;
    dx[e] = gamma[c] * rs * (g - sum_dy[c] / (float)N - xhat * (sum_dyx[c] / (float)N));
  } else {
    dx[e] = gamma[c] * rs * g;
  }
}

__global__ __launch_bounds__(256) void bn_param_grads_kernel(const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx, int C,
                                                             float* __restrict__ dgamma, float bg, float* __restrict__ dbeta, float bb) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (dgamma) dgamma[c] = (bg != 0.f ? dgamma[c] : 0.f) + sum_dyx[c];
  if (dbeta) dbeta[c] = (bb != 0.f ? dbeta[c] : 0.f) + sum_dy[c];
}

}  // namespace

using namespace yt8m;

static int sample_check(const void* x, void* out, int64_t B, int64_t F, int64_t D, int64_t S, int mode) {
  YT8M_REQUIRE(B >= 0 && F >= 1 && D >= 1 && S >= 1, YT8M_E_SHAPE, "bad dimension");
  YT8M_REQUIRE(mode == 0 || mode == 1, YT8M_E_BADARG, "mode: 0 = random frames, 1 = random sequence");
  YT8M_REQUIRE(B * S < (1LL << 31) && F < (1LL << 31) && D < (1LL << 31), YT8M_E_SHAPE, "dimension too large");
  if (B > 0) YT8M_REQUIRE(x && out, YT8M_E_BADARG, "null operand");
  return YT8M_OK;
}

extern "C" int yt8m_sample_frames_f32(const float* x, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, int64_t S, int mode,
                                      uint64_t seed, float* out, int32_t* idx_out, yt8m_stream_t stream) {
  int rc = sample_check(x, out, B, F, D, S, mode);
  if (rc != YT8M_OK || B == 0) return rc;
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(sample_gather_kernel<float>, dim3((unsigned)(B * S)), dim3(256), 0, s, x, num_frames, out, idx_out, (int)F, (int)D,
                     (int)S, mode, (unsigned long long)seed);
  return launch_status("sample_gather_kernel<float>");
}

extern "C" int yt8m_sample_frames_u8(const uint8_t* x, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, int64_t S, int mode,
                                     uint64_t seed, uint8_t* out, int32_t* idx_out, yt8m_stream_t stream) {
  int rc = sample_check(x, out, B, F, D, S, mode);
  if (rc != YT8M_OK || B == 0) return rc;
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(sample_gather_kernel<uint8_t>, dim3((unsigned)(B * S)), dim3(256), 0, s, x, num_frames, out, idx_out, (int)F,
                     (int)D, (int)S, mode, (unsigned long long)seed);
  return launch_status("sample_gather_kernel<uint8_t>");
}

extern "C" int yt8m_frame_pool_fwd(const float* x, int64_t B, int64_t S, int64_t C, int mode, float* out, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && S >= 1 && C >= 0 && B < 65536, YT8M_E_SHAPE, "bad dimension");
  YT8M_REQUIRE(mode == 0 || mode == 1, YT8M_E_BADARG, "mode: 0 = max, 1 = average");
  if (B * C == 0) return YT8M_OK;
  YT8M_REQUIRE(x && out, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0)
    hipLaunchKernelGGL(frame_pool_fwd_kernel<4>, dim3((unsigned)((C / 4 + 255) / 256), (unsigned)B), dim3(256), 0, s, x, out, (int)S, (int)C, mode);
  else
    hipLaunchKernelGGL(frame_pool_fwd_kernel<1>, dim3((unsigned)((C + 255) / 256), (unsigned)B), dim3(256), 0, s, x, out, (int)S, (int)C, mode);
  return launch_status("frame_pool_fwd_kernel");
}

extern "C" int yt8m_frame_pool_bwd(const float* x, const float* out, const float* dy, int64_t B, int64_t S, int64_t C, int mode,
                                   float* dx, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && S >= 1 && C >= 0 && B < 65536, YT8M_E_SHAPE, "bad dimension");
  YT8M_REQUIRE(mode == 0 || mode == 1, YT8M_E_BADARG, "mode: 0 = max, 1 = average");
  if (B * C == 0) return YT8M_OK;
  YT8M_REQUIRE(x && out && dy && dx, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(dy) |
                         reinterpret_cast<uintptr_t>(dx)) & 15) == 0)
    hipLaunchKernelGGL(frame_pool_bwd_kernel<4>, dim3((unsigned)((C / 4 + 255) / 256), (unsigned)B), dim3(256), 0, s, x, out, dy, dx, (int)S,
                       (int)C, mode);
  else
    hipLaunchKernelGGL(frame_pool_bwd_kernel<1>, dim3((unsigned)((C + 255) / 256), (unsigned)B), dim3(256), 0, s, x, out, dy, dx, (int)S,
                       (int)C, mode);
  return launch_status("frame_pool_bwd_kernel");
}

extern "C" int yt8m_batchnorm_fwd(const float* x, int64_t N, int64_t C, const float* gamma, const float* beta, float* moving_mean,
                                  float* moving_var, int training, float eps, float decay, float* y, float* save_mean,
                                  float* save_rstd, yt8m_stream_t stream) {
  YT8M_REQUIRE(N >= 0 && C >= 0 && N < (1LL << 31) && C < (1LL << 31), YT8M_E_SHAPE, "bad dimension");
  if (N * C == 0) return YT8M_OK;
  YT8M_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && moving_mean && moving_var, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  if (training)
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, s, x, (int)N, (int)C, eps, decay, save_mean,
                       save_rstd, moving_mean, moving_var);
  else
    hipLaunchKernelGGL(bn_frozen_stats_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, moving_mean, moving_var, (int)C, eps,
                       save_mean, save_rstd);
  const long long n = (long long)N * C;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, (int)C, save_mean, save_rstd, gamma, beta, y);
  return launch_status("bn_apply_kernel");
}

extern "C" int64_t yt8m_batchnorm_workspace_bytes(int64_t C) { return 2 * C * (int64_t)sizeof(float); }

extern "C" int yt8m_batchnorm_bwd(const float* x, const float* dy, int64_t N, int64_t C, const float* gamma, const float* save_mean,
                                  const float* save_rstd, int training, float* dx, float* dgamma, float dgamma_beta, float* dbeta,
                                  float dbeta_beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(N >= 0 && C >= 0 && N < (1LL << 31) && C < (1LL << 31), YT8M_E_SHAPE, "bad dimension");
  if (N * C == 0) return YT8M_OK;
  YT8M_REQUIRE(x && dy && gamma && save_mean && save_rstd && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(workspace_bytes >= yt8m_batchnorm_workspace_bytes(C), YT8M_E_SHAPE, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  float* sum_dy = static_cast<float*>(workspace);
  float* sum_dyx = sum_dy + C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, s, x, dy, (int)N, (int)C, save_mean, save_rstd,
                     sum_dy, sum_dyx);
  if (dgamma || dbeta)
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, s, sum_dy, sum_dyx, (int)C, dgamma,
                       dgamma_beta, dbeta, dbeta_beta);
  if (dx) {
    const long long n = (long long)N * C;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, dy, n, (int)N, (int)C, save_mean,
                       save_rstd, gamma, sum_dy, sum_dyx, training, dx);
  }
  return launch_status("bn_bwd kernels");
}
