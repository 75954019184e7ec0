// netvlad.hip -- NetVLAD residual aggregation + intra-normalisation (NOT in the reference).
//   n[b,k]      = sum_f a[b,f,k]
//   pre[b,k,:]  = agg[b,k,:] - n[b,k] * c[k,:]            agg = a^T x (batched GEMM), c = cluster centres
//   vlad[b,k,:] = pre * rsqrt(max(sum_d pre^2, eps))      (intra-normalisation)
// One wave per (b,k) row of D floats: the row makes ONE trip through HBM (read agg, write vlad).  HBM-bound:
// 8 B per (b,k,d) forward, 16 B backward.  Backward per Appendix G (l2-normalise) + product rule:
//   dpre = r*(dy - y*(y.dy))  [or r*dy when the row norm was clamped];  dagg = dpre;
//   dn[b,k] = -sum_d dpre*c;  dc[k,:] = -sum_b n[b,k]*dpre[b,k,:]  (second kernel, fixed order => deterministic).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void vlad_finish_fwd_kernel(const float* __restrict__ agg, const float* __restrict__ a,
                                                              const float* __restrict__ c, float* __restrict__ vlad,
                                                              float* __restrict__ n_out, int64_t B, int64_t F, int64_t K,
                                                              int64_t D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b,k)
  if (row >= B * K) return;
  const int64_t b = row / K, k = row - b * K;
  float n;
  if (a) {
    n = 0.f;
    for (int64_t f = lane; f < F; f += 64) n += a[(b * F + f) * K + k];
    n = wave_sum(n);
  } else {
    n = n_out[row];                                                       // precomputed by the fused pooling kernels
  }
  const float* ar = agg + row * D;
  const float* cr = c + k * D;
  float ss = 0.f;
  for (int64_t d = lane; d < D; d += 64) { const float v = ar[d] - n * cr[d]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(fmaxf(ss, eps));
  float* vr = vlad + row * D;
  for (int64_t d = lane; d < D; d += 64) vr[d] = (ar[d] - n * cr[d]) * r;
  if (a && lane == 0) n_out[row] = n;
}

__global__ __launch_bounds__(256) void vlad_finish_bwd_kernel(const float* __restrict__ agg, const float* __restrict__ n_in,
                                                              const float* __restrict__ c, const float* __restrict__ dy,
                                                              float* __restrict__ dagg, float* __restrict__ dn, int64_t BK,
                                                              int64_t K, int64_t D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BK) return;
  const int64_t k = row % K;
  const float n = n_in[row];
  const float* ar = agg + row * D;
  const float* cr = c + k * D;
  const float* gr = dy + row * D;
  float ss = 0.f, xd = 0.f;
  for (int64_t d = lane; d < D; d += 64) { const float v = ar[d] - n * cr[d]; ss += v * v; xd += v * gr[d]; }
  ss = wave_sum(ss);
  xd = wave_sum(xd);
  const float r = rsqrtf(fmaxf(ss, eps));
  const float kk = ss > eps ? xd * r * r : 0.f;
  float* dr = dagg + row * D;
  float dnc = 0.f;
  for (int64_t d = lane; d < D; d += 64) {
    const float v = ar[d] - n * cr[d];
    const float dp = r * (gr[d] - v * kk);
    dr[d] = dp;
    dnc += dp * cr[d];
  }
  dnc = wave_sum(dnc);
  if (lane == 0) dn[row] = -dnc;
}

// ---- the same two passes with the row held in registers (D % 4 == 0, D <= 2048: 8 float4 per lane) ---------------------------------
// The loops above walk a row with one 4-byte load in flight per lane and read agg twice (forward) / agg and dy twice (backward):
// 177 / 241 us at [1024, 64, 1152] against 604 / 906 MB.  Here every operand row is fetched once, as float4, all loads issued
// before the first use.
// q_out [B,K] (optional) = ||vlad[b,k,:]||^2 = ss * r^2 -- 1 unless the row norm was clamped.  The caller's l2-normalisation of
// the whole descriptor only needs sum_k q (l2_normalize over [K D] after the intra-normalisation), so that
// pass over [B,K,D] is replaced by a per-video scale of the hidden layer's output; dq (optional, backward) is the gradient that
// flows back through q: non-zero only into clamped rows (q = ss / eps there).
__global__ __launch_bounds__(256) void vlad_finish_fwd_reg_kernel(const float* __restrict__ agg, const float* __restrict__ a,
                                                                  const float* __restrict__ c, float* __restrict__ vlad,
                                                                  float* __restrict__ n_out, float* __restrict__ q_out, int64_t B,
                                                                  int64_t F, int64_t K, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b,k)
  if (row >= B * K) return;
  const int64_t b = row / K, k = row - b * K;
  const float4* ar = reinterpret_cast<const float4*>(agg + row * D);
  const float4* cr = reinterpret_cast<const float4*>(c + k * D);
  const int nd = D >> 2;
  float4 v[8], cv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int d = lane + 64 * i;
    v[i] = d < nd ? ar[d] : float4{0.f, 0.f, 0.f, 0.f};
    cv[i] = d < nd ? cr[d] : float4{0.f, 0.f, 0.f, 0.f};
  }
  float n;
  if (a) {
    n = 0.f;
    for (int64_t f = lane; f < F; f += 64) n += a[(b * F + f) * K + k];
    n = wave_sum(n);
  } else {
    n = n_out[row];                                                       // precomputed by the fused pooling kernels
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x -= n * cv[i].x; v[i].y -= n * cv[i].y; v[i].z -= n * cv[i].z; v[i].w -= n * cv[i].w;
    ss += v[i].x * v[i].x; ss += v[i].y * v[i].y; ss += v[i].z * v[i].z; ss += v[i].w * v[i].w;
  }
  ss = wave_sum(ss);
  const float r = rsqrtf(fmaxf(ss, eps));
  float4* vr = reinterpret_cast<float4*>(vlad + row * D);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int d = lane + 64 * i;
    if (d < nd) vr[d] = float4{v[i].x * r, v[i].y * r, v[i].z * r, v[i].w * r};
  }
  if (lane == 0) {
    if (a) n_out[row] = n;
    if (q_out) q_out[row] = ss * r * r;
  }
}

__global__ __launch_bounds__(256) void vlad_finish_bwd_reg_kernel(const float* __restrict__ agg, const float* __restrict__ n_in,
                                                                  const float* __restrict__ c, const float* __restrict__ dy,
                                                                  const float* __restrict__ dq, float* __restrict__ dagg,
                                                                  float* __restrict__ dn, int64_t BK, int64_t K, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= BK) return;
  const int64_t k = row % K;
  const float4* ar = reinterpret_cast<const float4*>(agg + row * D);
  const float4* cr = reinterpret_cast<const float4*>(c + k * D);
  const float4* gr = reinterpret_cast<const float4*>(dy + row * D);
  const int nd = D >> 2;
  float4 v[8], cv[8], g[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int d = lane + 64 * i;
    v[i] = d < nd ? ar[d] : float4{0.f, 0.f, 0.f, 0.f};
    cv[i] = d < nd ? cr[d] : float4{0.f, 0.f, 0.f, 0.f};
    g[i] = d < nd ? gr[d] : float4{0.f, 0.f, 0.f, 0.f};
  }
  const float n = n_in[row];
  float ss = 0.f, xd = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x -= n * cv[i].x; v[i].y -= n * cv[i].y; v[i].z -= n * cv[i].z; v[i].w -= n * cv[i].w;
    ss += v[i].x * v[i].x; ss += v[i].y * v[i].y; ss += v[i].z * v[i].z; ss += v[i].w * v[i].w;
    xd += v[i].x * g[i].x; xd += v[i].y * g[i].y; xd += v[i].z * g[i].z; xd += v[i].w * g[i].w;
  }
  ss = wave_sum(ss);
  xd = wave_sum(xd);
  const float r = rsqrtf(fmaxf(ss, eps));
  const float kk = ss > eps ? xd * r * r : 0.f;
  const float qq = (dq && !(ss > eps)) ? 2.f * dq[row] * r : 0.f;         // d(ss r^2)/dv = 2 v r^2 on a clamped row; as r * (qq v)
  float4* dr = reinterpret_cast<float4*>(dagg + row * D);
  float dnc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int d = lane + 64 * i;
    float4 o;
    o.x = r * (g[i].x - v[i].x * kk + v[i].x * qq); o.y = r * (g[i].y - v[i].y * kk + v[i].y * qq);
    o.z = r * (g[i].z - v[i].z * kk + v[i].z * qq); o.w = r * (g[i].w - v[i].w * kk + v[i].w * qq);
    if (d < nd) dr[d] = o;
    dnc += o.x * cv[i].x; dnc += o.y * cv[i].y; dnc += o.z * cv[i].z; dnc += o.w * cv[i].w;
  }
  dnc = wave_sum(dnc);
  if (lane == 0) dn[row] = -dnc;
}

bool vlad_reg_ok(int64_t D, const void* p0, const void* p1, const void* p2, const void* p3) {
  return D >= 4 && D <= 2048 && (D & 3) == 0 &&
         ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2) | reinterpret_cast<uintptr_t>(p3)) & 15) == 0;
}

// dc[k,d] (+)= -sum_b n[b,k] * dpre[b,k,d]
__global__ __launch_bounds__(256) void vlad_dcentres_kernel(const float* __restrict__ n_in, const float* __restrict__ dpre,
                                                            float* __restrict__ dc, int64_t B, int64_t K, int64_t D,
                                                            int accumulate) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;           // (k,d)
  if (e >= K * D) return;
  const int64_t k = e / D;
  // eight independent partial sums: eight loads in flight per lane (one (k, d) per lane and a serial loop over the batch ran at
  // 0.75 TB/s: 288 workgroups cannot cover the memory latency with one load each); fixed combination order
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int64_t b = 0;
  for (; b + 8 <= B; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += n_in[(b + u) * K + k] * dpre[(b + u) * K * D + e];
  }
  for (; b < B; ++b) p[0] += n_in[b * K + k] * dpre[b * K * D + e];
  const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  dc[e] = accumulate ? dc[e] - s : -s;
}

}  // namespace

using namespace yt8m;

namespace {
int vlad_finish_fwd_impl(const float* agg, const float* a, const float* centres, float* vlad, float* n_out, float* q_out, int64_t B,
                         int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && K >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * K * D == 0) return YT8M_OK;
  YT8M_REQUIRE(agg && centres && vlad && n_out, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  if (vlad_reg_ok(D, agg, centres, vlad, nullptr)) {
    hipLaunchKernelGGL(vlad_finish_fwd_reg_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, a, centres, vlad, n_out, q_out, B,
                       F, K, (int)D, eps);
    return launch_status("vlad_finish_fwd_reg_kernel");
  }
  YT8M_REQUIRE(!q_out, YT8M_E_SHAPE, "the q output needs D % 4 == 0, D <= 2048 and 16-byte aligned operands");
  hipLaunchKernelGGL(vlad_finish_fwd_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, a, centres, vlad, n_out, B,
                     F, K, D, eps);
  return launch_status("vlad_finish_fwd_kernel");
}

int vlad_finish_bwd_impl(const float* agg, const float* n_in, const float* centres, const float* dvlad, const float* dq, float* dagg,
                         float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D, float eps,
                         yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && K >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(dcentres_beta == 0.f || dcentres_beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (B * K * D == 0) return YT8M_OK;
  YT8M_REQUIRE(agg && n_in && centres && dvlad && dagg && dn, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  if (vlad_reg_ok(D, agg, centres, dvlad, dagg)) {
    hipLaunchKernelGGL(vlad_finish_bwd_reg_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, n_in, centres, dvlad, dq, dagg,
                       dn, B * K, K, (int)D, eps);
  } else {
    YT8M_REQUIRE(!dq, YT8M_E_SHAPE, "the q gradient needs D % 4 == 0, D <= 2048 and 16-byte aligned operands");
    hipLaunchKernelGGL(vlad_finish_bwd_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, n_in, centres, dvlad, dagg,
                       dn, B * K, K, D, eps);
  }
  if (dcentres)
    hipLaunchKernelGGL(vlad_dcentres_kernel, dim3((unsigned)((K * D + 255) / 256)), dim3(256), 0, s, n_in, dagg, dcentres, B, K, D,
                       dcentres_beta != 0.f ? 1 : 0);
  return launch_status("vlad_finish_bwd_kernel");
}
}  // namespace

extern "C" int yt8m_vlad_finish_fwd(const float* agg, const float* a, const float* centres, float* vlad, float* n_out, int64_t B,
                                    int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream) {
  return vlad_finish_fwd_impl(agg, a, centres, vlad, n_out, nullptr, B, F, K, D, eps, stream);
}

// + q_out [B,K] = ||vlad[b,k,:]||^2 (see vlad_finish_fwd_reg_kernel); yt8m_vlad_finish_q_supported(D) tells whether the shape is covered
extern "C" int yt8m_vlad_finish_q_supported(int64_t D) { return (D >= 4 && D <= 2048 && (D & 3) == 0) ? 1 : 0; }
extern "C" int yt8m_vlad_finish_q_fwd(const float* agg, const float* a, const float* centres, float* vlad, float* n_out, float* q_out,
                                      int64_t B, int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(q_out, YT8M_E_BADARG, "null q_out");
  return vlad_finish_fwd_impl(agg, a, centres, vlad, n_out, q_out, B, F, K, D, eps, stream);
}

extern "C" int yt8m_vlad_finish_bwd(const float* agg, const float* n_in, const float* centres, const float* dvlad, float* dagg,
                                    float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D, float eps,
                                    yt8m_stream_t stream) {
  return vlad_finish_bwd_impl(agg, n_in, centres, dvlad, nullptr, dagg, dn, dcentres, dcentres_beta, B, K, D, eps, stream);
}

// + dq [B,K] (may be NULL): the gradient that reaches q_out of yt8m_vlad_finish_q_fwd
extern "C" int yt8m_vlad_finish_q_bwd(const float* agg, const float* n_in, const float* centres, const float* dvlad, const float* dq,
                                      float* dagg, float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D,
                                      float eps, yt8m_stream_t stream) {
  return vlad_finish_bwd_impl(agg, n_in, centres, dvlad, dq, dagg, dn, dcentres, dcentres_beta, B, K, D, eps, stream);
}
