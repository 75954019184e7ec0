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

extern "C" int yt8m_vlad_finish_fwd(const float* agg, const float* a, const float* centres, float* vlad, float* n_out, int64_t B,
                                    int64_t F, int64_t K, int64_t D, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && K >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * K * D == 0) return YT8M_OK;
  YT8M_REQUIRE(agg && centres && vlad && n_out, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  hipLaunchKernelGGL(vlad_finish_fwd_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, a, centres, vlad, n_out, B,
                     F, K, D, eps);
  return launch_status("vlad_finish_fwd_kernel");
}

extern "C" int yt8m_vlad_finish_bwd(const float* agg, const float* n_in, const float* centres, const float* dvlad, float* dagg,
                                    float* dn, float* dcentres, float dcentres_beta, int64_t B, int64_t K, int64_t D, float eps,
                                    yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && K >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(dcentres_beta == 0.f || dcentres_beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (B * K * D == 0) return YT8M_OK;
  YT8M_REQUIRE(agg && n_in && centres && dvlad && dagg && dn, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  hipLaunchKernelGGL(vlad_finish_bwd_kernel, dim3((unsigned)((B * K + 3) / 4)), dim3(256), 0, s, agg, n_in, centres, dvlad, dagg,
                     dn, B * K, K, D, eps);
  if (dcentres)
    hipLaunchKernelGGL(vlad_dcentres_kernel, dim3((unsigned)((K * D + 255) / 256)), dim3(256), 0, s, n_in, dagg, dcentres, B, K, D,
                       dcentres_beta != 0.f ? 1 : 0);
  return launch_status("vlad_finish_bwd_kernel");
}
