// elementwise.hip -- HBM-bound kernels of the head / loss / input-transform slice (gfx950, wave64).
// All of them are one pass over their operands with lane-consecutive (coalesced) addressing and
// wave-shuffle reductions; none is reshaped into a GEMM.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// MoE mixing: p[b,l] = sum_{m<M} softmax(Zg[b,l,0..M])[m] * sigmoid(Ze[b,l,m])
// W/all_video_models/moe_model.py:54-64.  One thread per (b,l); a wave touches (M+1)*64 and M*64
// consecutive floats, so every fetched line is fully used.
constexpr int MAXM = 16;

template <int MT>
__global__ __launch_bounds__(256) void moe_mix_fwd_kernel(const float* __restrict__ Zg, const float* __restrict__ Ze,
                                                          float* __restrict__ p, int64_t BV, int Mrt) {
  const int M = MT > 0 ? MT : Mrt;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BV) return;
  const float* g = Zg + i * (M + 1);
  const float* e = Ze + i * M;
  float gl[MT > 0 ? MT + 1 : MAXM + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m <= M; ++m) { gl[m] = g[m]; mx = fmaxf(mx, gl[m]); }
  float den = 0.f, num = 0.f;
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    const float ex = expf(gl[m] - mx);
    den += ex;
    if (m < M) num += ex * (1.0f / (1.0f + expf(-e[m])));
  }
  p[i] = num / den;
}

// The same mixing on bf16 logits (round 6: the b1 GEMM of the bf16 configuration writes them as bf16 -- yt8m_gemm_b1_nt_grouped_bf16c),
// M = 2: a thread owns four consecutive labels = 12 gate + 8 expert logits = 24 + 16 contiguous bytes in, 16 out.
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__global__ __launch_bounds__(256) void moe_mix_fwd_bf16z_kernel(const unsigned short* __restrict__ Zg, const unsigned short* __restrict__ Ze,
                                                                float* __restrict__ p, int64_t BV4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BV4) return;
  const uint2* gp = reinterpret_cast<const uint2*>(Zg + i * 12);
  const uint2 g0 = gp[0], g1 = gp[1], g2 = gp[2];
  const uint4 ev = *reinterpret_cast<const uint4*>(Ze + i * 8);
  const unsigned gw[6] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y};
  const unsigned ew[4] = {ev.x, ev.y, ev.z, ev.w};
  float g[12], e[8];
#pragma unroll
  for (int k = 0; k < 6; ++k) { g[2 * k] = bf16_lo(gw[k]); g[2 * k + 1] = bf16_hi(gw[k]); }
#pragma unroll
  for (int k = 0; k < 4; ++k) { e[2 * k] = bf16_lo(ew[k]); e[2 * k + 1] = bf16_hi(ew[k]); }
  float o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a0 = g[3 * j], a1 = g[3 * j + 1], a2 = g[3 * j + 2];
    const float mx = fmaxf(a0, fmaxf(a1, a2));
    const float x0 = expf(a0 - mx), x1 = expf(a1 - mx), x2 = expf(a2 - mx);
    const float num = x0 * (1.0f / (1.0f + expf(-e[2 * j]))) + x1 * (1.0f / (1.0f + expf(-e[2 * j + 1])));
    o[j] = num / (x0 + x1 + x2);
  }
  *reinterpret_cast<float4*>(p + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
}

// in-place backward: Zg <- dL/dZg, Ze <- dL/dZe  
template <int MT>
__global__ __launch_bounds__(256) void moe_mix_bwd_kernel(float* __restrict__ Zg, float* __restrict__ Ze,
                                                          const float* __restrict__ dp, int64_t BV, int Mrt) {
  const int M = MT > 0 ? MT : Mrt;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= BV) return;
  float* g = Zg + i * (M + 1);
  float* e = Ze + i * M;
  float gs[MT > 0 ? MT + 1 : MAXM + 1], es[MT > 0 ? MT + 1 : MAXM + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m <= M; ++m) { gs[m] = g[m]; mx = fmaxf(mx, gs[m]); }
  float den = 0.f;
#pragma unroll
  for (int m = 0; m <= M; ++m) { gs[m] = expf(gs[m] - mx); den += gs[m]; }
  const float inv = 1.0f / den;
  float pv = 0.f;
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    gs[m] *= inv;
    es[m] = (m < M) ? 1.0f / (1.0f + expf(-e[m])) : 0.f;
    pv += gs[m] * es[m];
  }
  const float d = dp[i];
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    g[m] = d * gs[m] * (es[m] - pv);
    if (m < M) e[m] = d * gs[m] * es[m] * (1.0f - es[m]);
  }
}

// ---- MoE mixing fused with CrossEntropyLoss (W/losses.py:110-130): the head's probabilities never make an extra
// HBM round trip.  fwd: p and the per-workgroup loss partials in one pass over Z; bwd: dL/dp is formed from the
// recomputed p and the labels in registers and folded straight into dL/dZ (in place).
template <int MT, typename LT>
__global__ __launch_bounds__(256) void moe_mix_xent_fwd_kernel(const float* __restrict__ Zg, const float* __restrict__ Ze,
                                                               const LT* __restrict__ y, float* __restrict__ p,
                                                               float* __restrict__ partial, int64_t BV, int Mrt, float eps) {
  __shared__ float red[4];
  const int M = MT > 0 ? MT : Mrt;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float ce = 0.f;
  if (i < BV) {
    const float* g = Zg + i * (M + 1);
    const float* e = Ze + i * M;
    float gl[MT > 0 ? MT + 1 : MAXM + 1];
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m <= M; ++m) { gl[m] = g[m]; mx = fmaxf(mx, gl[m]); }
    float den = 0.f, num = 0.f;
#pragma unroll
    for (int m = 0; m <= M; ++m) {
      const float ex = expf(gl[m] - mx);
      den += ex;
      if (m < M) num += ex * (1.0f / (1.0f + expf(-e[m])));
    }
    const float pv = num / den;
    p[i] = pv;
    const float yv = (float)y[i];
    ce = -(yv * logf(pv + eps) + (1.0f - yv) * logf(1.0f - pv + eps));
  }
  ce = block_sum_256(ce, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = ce;
}

constexpr int ZMAX_IT = 8;
// zmax (may be NULL; round 6): two words, max |dL/dZg| and max |dL/dZe| as float bits (atomic max; zeroed by the caller) -- what the
// weight-gradient products' h2 split would otherwise measure in a pass of its own over each of the two gradients.
template <int MT, typename LT>
__global__ __launch_bounds__(256) void moe_mix_xent_bwd_kernel(float* __restrict__ Zg, float* __restrict__ Ze,
                                                               const LT* __restrict__ y, int64_t BV, int Mrt, float eps,
                                                               float dscale, const float* __restrict__ up_dev, unsigned* __restrict__ zmax) {
  __shared__ float redm[8];
  const int M = MT > 0 ? MT : Mrt;
  if (up_dev) dscale *= up_dev[0];
  float mg = 0.f, me = 0.f;
  // with zmax a workgroup walks ZMAX_IT x 256 labels: one pair of block reductions and one look at the words per 2 048 labels (per 256
  // labels they made this 31 us HBM-bound pass a 72 us one)
  const int iters = zmax ? ZMAX_IT : 1;
  for (int it = 0; it < iters; ++it) {
  const int64_t i = ((int64_t)blockIdx.x * iters + it) * 256 + threadIdx.x;
  if (i < BV) {
  float* g = Zg + i * (M + 1);
  float* e = Ze + i * M;
  float gs[MT > 0 ? MT + 1 : MAXM + 1], es[MT > 0 ? MT + 1 : MAXM + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int m = 0; m <= M; ++m) { gs[m] = g[m]; mx = fmaxf(mx, gs[m]); }
  float den = 0.f;
#pragma unroll
  for (int m = 0; m <= M; ++m) { gs[m] = expf(gs[m] - mx); den += gs[m]; }
  const float inv = 1.0f / den;
  float pv = 0.f;
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    gs[m] *= inv;
    es[m] = (m < M) ? 1.0f / (1.0f + expf(-e[m])) : 0.f;
    pv += gs[m] * es[m];
  }
  const float yv = (float)y[i];
  const float d = -(yv / (pv + eps) - (1.0f - yv) / (1.0f - pv + eps)) * dscale;
#pragma unroll
  for (int m = 0; m <= M; ++m) {
    const float vg = d * gs[m] * (es[m] - pv);
    g[m] = vg;
    mg = fmaxf(mg, fabsf(vg));
    if (m < M) {
      const float ve = d * gs[m] * es[m] * (1.0f - es[m]);
      e[m] = ve;
      me = fmaxf(me, fabsf(ve));
    }
  }
  }
  }
  if (zmax) {                                                        // (block-uniform)
    mg = block_max_256(mg, redm);
    me = block_max_256(me, redm + 4);
    if (threadIdx.x == 0) {
      // non-negative floats order like their bit patterns.  Look before the atomic: 19 k workgroups x 2 atomics on one line serialised
      // (+0.34 ms on a 0.9 ms step when every workgroup issued them); the running maximum settles within the first few hundred
      // workgroups, after which a relaxed read says "not larger" and nothing is issued.
      const unsigned bg = __float_as_uint(mg), be = __float_as_uint(me);
      if (bg > __hip_atomic_load(zmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(zmax, bg);
      if (be > __hip_atomic_load(zmax + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(zmax + 1, be);
    }
  }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case YT8M_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case YT8M_ACT_RELU: return fmaxf(x, 0.f);
    case YT8M_ACT_RELU6: return fminf(fmaxf(x, 0.f), 6.f);
    case YT8M_ACT_TANH: return tanhf(x);
    default: return x > 0.f ? x : expf(x) - 1.0f;  // ELU
  }
}
__device__ __forceinline__ float act_grad_from_out(int act, float y) {
  switch (act) {
    case YT8M_ACT_SIGMOID: return y * (1.0f - y);
    case YT8M_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case YT8M_ACT_RELU6: return (y > 0.f && y < 6.f) ? 1.f : 0.f;
    case YT8M_ACT_TANH: return 1.0f - y * y;
    default: return y > 0.f ? 1.f : y + 1.0f;  // ELU
  }
}
__global__ __launch_bounds__(256) void act_fwd_kernel(int act, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = act_apply(act, x[i]);
}
__global__ __launch_bounds__(256) void act_bwd_kernel(int act, const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dx[i] = dy[i] * act_grad_from_out(act, y[i]);
}

// column sums (bias gradients): 1024 threads = 64 columns x 16 row groups; fixed-order LDS tree => deterministic
// gridDim.y > 1: tall-and-narrow inputs (attention / cluster logits: [B*F, 8..64]) -- block (x, y) sums the row range
// [y*rows_per, (y+1)*rows_per) into partial[y][col]; colsum_finish_kernel adds the partials in fixed order.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ X, int64_t rows_total, int64_t cols, int64_t ldx,
                                                      float* __restrict__ out, int accumulate, int64_t rows_per) {
  __shared__ float red[16][65];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  if (gridDim.y > 1) {
    X += (int64_t)blockIdx.y * rows_per * ldx;
    out += (int64_t)blockIdx.y * cols;
  }
  const int64_t rows = gridDim.y > 1 ? min(rows_per, rows_total - (int64_t)blockIdx.y * rows_per) : rows_total;
  float s = 0.f;
  if (col < cols) {
    int64_t r = rg;
    for (; r + 48 < rows; r += 64) {
      const float a = X[r * ldx + col], b = X[(r + 16) * ldx + col], d = X[(r + 32) * ldx + col], e = X[(r + 48) * ldx + col];
      s += (a + b) + (d + e);
    }
    for (; r < rows; r += 16) s += X[r * ldx + col];
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][c];
    out[col] = accumulate ? out[col] + t : t;
  }
}

// Two column sums from one pass: out[col] = sum_r X[r][col] and outw[col] = sum_r w[r] X[r][col] (the bias gradient and the
// rank-1 term of the layer-0 weight gradient on uint8 frames).  Same decomposition and fixed summation order as colsum_kernel.
__global__ __launch_bounds__(1024) void colsum2_kernel(const float* __restrict__ X, const float* __restrict__ w, int64_t rows_total,
                                                       int64_t cols, int64_t ldx, float* __restrict__ out, float* __restrict__ outw,
                                                       int accumulate, int64_t rows_per) {
  __shared__ float red[2][16][65];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + c;
  if (gridDim.y > 1) {
    X += (int64_t)blockIdx.y * rows_per * ldx;
    w += (int64_t)blockIdx.y * rows_per;
    out += (int64_t)blockIdx.y * cols;
    outw += (int64_t)blockIdx.y * cols;
  }
  const int64_t rows = gridDim.y > 1 ? min(rows_per, rows_total - (int64_t)blockIdx.y * rows_per) : rows_total;
  float s = 0.f, sw = 0.f;
  if (col < cols) {
    int64_t r = rg;
    for (; r + 48 < rows; r += 64) {
      const float a = X[r * ldx + col], b = X[(r + 16) * ldx + col], d = X[(r + 32) * ldx + col], e = X[(r + 48) * ldx + col];
      s += (a + b) + (d + e);
      sw += (a * w[r] + b * w[r + 16]) + (d * w[r + 32] + e * w[r + 48]);
    }
    for (; r < rows; r += 16) { const float a = X[r * ldx + col]; s += a; sw += a * w[r]; }
  }
  red[0][rg][c] = s;
  red[1][rg][c] = sw;
  __syncthreads();
  if (rg < 2 && col < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[rg][k][c];
    float* o = rg == 0 ? out : outw;
    o[col] = (accumulate && rg == 0) ? o[col] + t : t;             // only the plain sum accumulates (bias gradient over chunks)
  }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, int nsplit, int64_t cols,
                                                            float* __restrict__ out, int accumulate) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  float t = 0.f;
  for (int k = 0; k < nsplit; ++k) t += partial[(int64_t)k * cols + col];
  out[col] = accumulate ? out[col] + t : t;
}

// ------------------------------------------------------------------------------------------------
// CrossEntropyLoss, W/losses.py:110-130 (probability space, eps = 10e-6), fused forward + dL/dp.
// grid = (ceil(V/1024), B); per-block partial sums -> fixed-order final reduction (deterministic).
template <typename LT>
__global__ __launch_bounds__(256) void xent_kernel(const float* __restrict__ p, const LT* __restrict__ y,
                                                   const float* __restrict__ w, float* __restrict__ dp,
                                                   float* __restrict__ partial, int64_t V, float eps, float dscale,
                                                   const float* __restrict__ up_dev) {
  __shared__ float red[4];
  const int64_t b = blockIdx.y;
  const float wb = w ? w[b] : 1.0f;
  if (up_dev) dscale *= up_dev[0];
  const int64_t base = b * V;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t l = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
    if (l < V) {
      const float pv = p[base + l];
      const float yv = (float)y[base + l];
      const float a = pv + eps, c = 1.0f - pv + eps;
      s -= yv * logf(a) + (1.0f - yv) * logf(c);
      if (dp) dp[base + l] = -(yv / a - (1.0f - yv) / c) * (wb * dscale);
    }
  }
  if (partial) {
    s = block_sum_256(s * wb, red);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

// fixed-order final reduction of the per-workgroup partials (any blockDim that is a multiple of 64, <= 1024)
__global__ __launch_bounds__(1024) void final_sum_kernel(const float* __restrict__ partial, int64_t n, float scale,
                                                         float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    out[0] = t * scale;
  }
}

// ------------------------------------------------------------------------------------------------
// tf.nn.l2_normalize on the last axis: one wave per row, 4 rows per workgroup.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows,
                                                         int64_t cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  float ss = 0.f;
  for (int64_t c = lane; c < cols; c += 64) { const float v = xr[c]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(fmaxf(ss, eps));
  float* yr = y + row * cols;
  for (int64_t c = lane; c < cols; c += 64) yr[c] = xr[c] * r;
}

// long rows (>= 512 columns, few rows: the video-level [B,1152] batch): one 256-thread workgroup per row
__global__ __launch_bounds__(256) void l2norm_fwd_row_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t cols,
                                                             float eps) {
  __shared__ float red[4];
  const float* xr = x + (int64_t)blockIdx.x * cols;
  float* yr = y + (int64_t)blockIdx.x * cols;
  float ss = 0.f;
  for (int64_t c = threadIdx.x; c < cols; c += 256) { const float v = xr[c]; ss += v * v; }
  ss = block_sum_256(ss, red);
  const float r = rsqrtf(fmaxf(ss, eps));
  for (int64_t c = threadIdx.x; c < cols; c += 256) yr[c] = xr[c] * r;
}

// very long rows (the [B, 64*1152] VLAD descriptor): one 1024-thread workgroup per row, float4 traffic; BWD adds the
// x.dy reduction and applies dx = r*(dy - x*k).  cols % 4 == 0 and 16-byte aligned rows (checked by the caller).
__device__ __forceinline__ float block_sum_1024(float v, float* red /* >= 16 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}

template <bool BWD>
__global__ __launch_bounds__(1024) void l2norm_long_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ out, int64_t cols, float eps) {
  __shared__ float red[16];
  const int64_t n4 = cols >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)blockIdx.x * cols);
  const float4* gr = BWD ? reinterpret_cast<const float4*>(dy + (int64_t)blockIdx.x * cols) : nullptr;
  float4* yr = reinterpret_cast<float4*>(out + (int64_t)blockIdx.x * cols);
  float ss = 0.f, xd = 0.f;
  for (int64_t c = threadIdx.x; c < n4; c += 1024) {
    const float4 v = xr[c];
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    if (BWD) { const float4 g = gr[c]; xd += (v.x * g.x + v.y * g.y) + (v.z * g.z + v.w * g.w); }
  }
  ss = block_sum_1024(ss, red);
  const float r = rsqrtf(fmaxf(ss, eps));
  if (!BWD) {
    for (int64_t c = threadIdx.x; c < n4; c += 1024) {
      const float4 v = xr[c];
      yr[c] = make_float4(v.x * r, v.y * r, v.z * r, v.w * r);
    }
  } else {
    xd = block_sum_1024(xd, red);
    const float k = ss > eps ? xd * r * r : 0.f;
    for (int64_t c = threadIdx.x; c < n4; c += 1024) {
      const float4 v = xr[c], g = gr[c];
      yr[c] = make_float4(r * (g.x - v.x * k), r * (g.y - v.y * k), r * (g.z - v.z * k), r * (g.w - v.w * k));
    }
  }
}

// dx = r*(dy - y*(y.dy)) if ss > eps else r*dy      
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int64_t rows, int64_t cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  const float* gr = dy + row * cols;
  float ss = 0.f, xd = 0.f;
  for (int64_t c = lane; c < cols; c += 64) { const float v = xr[c]; ss += v * v; xd += v * gr[c]; }
  ss = wave_sum(ss);
  xd = wave_sum(xd);
  const float r = rsqrtf(fmaxf(ss, eps));
  const float k = ss > eps ? xd * r * r : 0.f;  // y.dy * r = (x.dy) r^2 ; applied to x below
  float* dr = dx + row * cols;
  for (int64_t c = lane; c < cols; c += 64) dr[c] = r * (gr[c] - xr[c] * k);
}

// readers.py:178-187 + utils.py:23-38 + default_transformer.py:7: one wave per frame row of raw uint8.
__global__ __launch_bounds__(256) void dequant_l2norm_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nf,
                                                             float* __restrict__ x, int64_t B, int64_t F, int64_t D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * F) return;
  const int64_t b = row / F, f = row - b * F;
  float* xr = x + row * D;
  const bool live = nf ? (f < (int64_t)nf[b]) : true;
  if (!live) {
    for (int64_t c = lane; c < D; c += 64) xr[c] = 0.f;
    return;
  }
  const uint8_t* qr = q + row * D;
  const float s = 4.0f / 255.0f, bias = 4.0f / 512.0f - 2.0f;
  float ss = 0.f;
  if ((D & 3) == 0 && D <= 2048 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
    // 4 bytes per lane and load, the row held in registers between the two passes, 16-byte stores (1 KiB per wave instruction)
    const int nd = (int)(D >> 2);
    uint32_t w[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c4 = lane + 64 * it;
      w[it] = c4 < nd ? reinterpret_cast<const uint32_t*>(qr)[c4] : 0u;
      if (c4 < nd) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float v = fmaf((float)((w[it] >> (8 * k)) & 255u), s, bias); ss += v * v; }
      }
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(fmaxf(ss, eps));
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int c4 = lane + 64 * it;
      if (c4 < nd) {
        float4 o;
        o.x = fmaf((float)(w[it] & 255u), s, bias) * r;
        o.y = fmaf((float)((w[it] >> 8) & 255u), s, bias) * r;
        o.z = fmaf((float)((w[it] >> 16) & 255u), s, bias) * r;
        o.w = fmaf((float)(w[it] >> 24), s, bias) * r;
        reinterpret_cast<float4*>(xr)[c4] = o;
      }
    }
    return;
  }
  for (int64_t c = lane; c < D; c += 64) { const float v = fmaf((float)qr[c], s, bias); ss += v * v; }
  ss = wave_sum(ss);
  const float r = rsqrtf(fmaxf(ss, eps));
  for (int64_t c = lane; c < D; c += 64) xr[c] = fmaf((float)qr[c], s, bias) * r;
}

// video-level features: mean over valid frames of the dequantised rows, then L2-normalise.  One workgroup per video.
__global__ __launch_bounds__(256) void dequant_mean_l2norm_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nf,
                                                                  float* __restrict__ x, int64_t F, int64_t D, float eps) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const int64_t n = nf ? (int64_t)min((int64_t)nf[b], F) : F;
  const uint8_t* qb = q + b * F * D;
  const float s = 4.0f / 255.0f, bias = 4.0f / 512.0f - 2.0f;
  float ss = 0.f;
  for (int64_t c = threadIdx.x; c < D; c += 256) {
    unsigned int acc = 0;
    for (int64_t f = 0; f < n; ++f) acc += qb[f * D + c];
    const float mean = n > 0 ? fmaf((float)acc / (float)n, s, bias) : 0.f;
    x[b * D + c] = mean;
    ss += mean * mean;
  }
  ss = block_sum_256(ss, red);
  const float r = rsqrtf(fmaxf(ss, eps));
  for (int64_t c = threadIdx.x; c < D; c += 256) x[b * D + c] *= r;
}

// ---- fp32 -> bf16 casts (round to nearest even), optionally transposing through a 64x64 LDS tile -----------------
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld,
                                                        unsigned short* __restrict__ dst, int64_t dld) {
  const int64_t n = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols, c = e - r * cols;
    dst[r * dld + c] = f2bf(src[r * ld + c]);
  }
}

__global__ __launch_bounds__(256) void cast_bf16_transpose_kernel(const float* __restrict__ src, int64_t rows, int64_t cols,
                                                                  int64_t ld, unsigned short* __restrict__ dst, int64_t dld) {
  __shared__ unsigned short tile[64][66];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? f2bf(src[r * ld + c]) : (unsigned short)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;          // dst[c][r]
    if (c < cols && r < rows) dst[c * dld + r] = tile[tx][i];
  }
}

__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) { return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16); }

// Vector forms for 16-byte aligned sources with cols % 4 == 0 (every tensor of the bf16 training path):
// plain: 8 elements per thread (two float4 in, one 16-byte store).
__global__ __launch_bounds__(256) void cast_bf16_vec_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld,
                                                            unsigned short* __restrict__ dst, int64_t dld) {
  const int64_t cg = cols >> 2;                                        // float4 groups per row
  const int64_t n = rows * cg;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cg, c = (e - r * cg) << 2;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + c);
    *reinterpret_cast<uint2*>(dst + r * dld + c) = uint2{pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
  }
}

// 64 x 64 tile: float4 loads (a row of the tile = 16 lanes), optional plain bf16 store (8 bytes per lane, 128 B per row),
// transposed store through an LDS tile of bf16 PAIRS [64 rows][32 column pairs + 1]: a lane gathers rows 4q..4q+3 of one
// column pair (4 ds_read_b32, 2-way conflicts at most) and writes 8 bytes to each of the two transposed rows, so a 16-lane
// group writes 128 contiguous bytes of dst[c][r0..r0+63].  PLAIN && TRANS: both layouts from ONE read of the fp32 source.
template <bool PLAIN, bool TRANS>
__global__ __launch_bounds__(256) void cast_bf16_tile_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, int64_t ld,
                                                             unsigned short* __restrict__ dplain, int64_t pld,
                                                             unsigned short* __restrict__ dtrans, int64_t tld) {
  __shared__ unsigned int tile[64][33];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int lr = tid >> 4, lc = (tid & 15) << 2;                     // 16 rows per pass
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int64_t r = r0 + lr + 16 * ps, c = c0 + lc;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && c < cols) v = *reinterpret_cast<const float4*>(src + r * ld + c);   // cols % 4 == 0: all or nothing
      const unsigned int p0 = pack_bf2(v.x, v.y), p1 = pack_bf2(v.z, v.w);
      if (PLAIN && r < rows && c < cols) *reinterpret_cast<uint2*>(dplain + r * pld + c) = uint2{p0, p1};
      if (TRANS) { tile[lr + 16 * ps][lc >> 1] = p0; tile[lr + 16 * ps][(lc >> 1) + 1] = p1; }
    }
  }
  if (!TRANS) return;
  __syncthreads();
  const int rq = tid & 15;                                              // rows 4rq .. 4rq+3 of the tile
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int cp = (tid >> 4) + 16 * ps;                                // column pair
    const unsigned int a = tile[4 * rq + 0][cp], b = tile[4 * rq + 1][cp], c = tile[4 * rq + 2][cp], d = tile[4 * rq + 3][cp];
    const int64_t cc = c0 + 2 * cp, rr = r0 + 4 * rq;
    if (rr + 3 < rows) {                                                // rows % 4 == 0 on this path
      if (cc < cols)
        *reinterpret_cast<uint2*>(dtrans + cc * tld + rr) = uint2{(a & 0xffffu) | (b << 16), (c & 0xffffu) | (d << 16)};
      if (cc + 1 < cols)
        *reinterpret_cast<uint2*>(dtrans + (cc + 1) * tld + rr) = uint2{(a >> 16) | (b & 0xffff0000u), (c >> 16) | (d & 0xffff0000u)};
    }
  }
}

inline unsigned grid_for(int64_t n, int per_block, int64_t cap = 1 << 20) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_moe_mix_fwd(const float* Zg, const float* Ze, float* p, int64_t B, int64_t V, int M,
                                yt8m_stream_t stream) {
  YT8M_REQUIRE(M >= 1 && M <= MAXM, YT8M_E_BADARG, "num_mixtures must be in [1,16]");
  YT8M_REQUIRE(B >= 0 && V >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(Zg && Ze && p, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t BV = B * V;
  dim3 grid((unsigned)((BV + 255) / 256)), block(256);
  switch (M) {
    case 1: hipLaunchKernelGGL((moe_mix_fwd_kernel<1>), grid, block, 0, s, Zg, Ze, p, BV, M); break;
    case 2: hipLaunchKernelGGL((moe_mix_fwd_kernel<2>), grid, block, 0, s, Zg, Ze, p, BV, M); break;
    case 4: hipLaunchKernelGGL((moe_mix_fwd_kernel<4>), grid, block, 0, s, Zg, Ze, p, BV, M); break;
    case 8: hipLaunchKernelGGL((moe_mix_fwd_kernel<8>), grid, block, 0, s, Zg, Ze, p, BV, M); break;
    default: hipLaunchKernelGGL((moe_mix_fwd_kernel<0>), grid, block, 0, s, Zg, Ze, p, BV, M); break;
  }
  return launch_status("moe_mix_fwd_kernel");
}

// p = mix(Zg, Ze) with bf16 logits Zg [B, 3V], Ze [B, 2V] (M = 2 only; B V % 4 == 0, 16-byte aligned operands)
extern "C" int yt8m_moe_mix_fwd_bf16z(const void* Zg, const void* Ze, float* p, int64_t B, int64_t V, int M, yt8m_stream_t stream) {
  YT8M_REQUIRE(M == 2, YT8M_E_BADARG, "the bf16-logit mixing pass is built for num_mixtures == 2");
  YT8M_REQUIRE(B >= 0 && V >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(Zg && Ze && p, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(((B * V) & 3) == 0, YT8M_E_SHAPE, "B V must be a multiple of 4");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(Zg) | reinterpret_cast<uintptr_t>(Ze) | reinterpret_cast<uintptr_t>(p)) & 15) == 0, YT8M_E_BADARG,
               "operands must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t n4 = B * V / 4;
  hipLaunchKernelGGL(moe_mix_fwd_bf16z_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, static_cast<const unsigned short*>(Zg),
                     static_cast<const unsigned short*>(Ze), p, n4);
  return launch_status("moe_mix_fwd_bf16z_kernel");
}

extern "C" int yt8m_moe_mix_bwd(float* Zg, float* Ze, const float* dp, int64_t B, int64_t V, int M,
                                yt8m_stream_t stream) {
  YT8M_REQUIRE(M >= 1 && M <= MAXM, YT8M_E_BADARG, "num_mixtures must be in [1,16]");
  YT8M_REQUIRE(B >= 0 && V >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(Zg && Ze && dp, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t BV = B * V;
  dim3 grid((unsigned)((BV + 255) / 256)), block(256);
  switch (M) {
    case 1: hipLaunchKernelGGL((moe_mix_bwd_kernel<1>), grid, block, 0, s, Zg, Ze, dp, BV, M); break;
    case 2: hipLaunchKernelGGL((moe_mix_bwd_kernel<2>), grid, block, 0, s, Zg, Ze, dp, BV, M); break;
    case 4: hipLaunchKernelGGL((moe_mix_bwd_kernel<4>), grid, block, 0, s, Zg, Ze, dp, BV, M); break;
    case 8: hipLaunchKernelGGL((moe_mix_bwd_kernel<8>), grid, block, 0, s, Zg, Ze, dp, BV, M); break;
    default: hipLaunchKernelGGL((moe_mix_bwd_kernel<0>), grid, block, 0, s, Zg, Ze, dp, BV, M); break;
  }
  return launch_status("moe_mix_bwd_kernel");
}

#define YT8M_DISPATCH_M(KERNEL, LT, ...)                                                                 \
  switch (M) {                                                                                          \
    case 1: hipLaunchKernelGGL((KERNEL<1, LT>), grid, block, 0, s, __VA_ARGS__); break;                  \
    case 2: hipLaunchKernelGGL((KERNEL<2, LT>), grid, block, 0, s, __VA_ARGS__); break;                  \
    case 4: hipLaunchKernelGGL((KERNEL<4, LT>), grid, block, 0, s, __VA_ARGS__); break;                  \
    case 8: hipLaunchKernelGGL((KERNEL<8, LT>), grid, block, 0, s, __VA_ARGS__); break;                  \
    default: hipLaunchKernelGGL((KERNEL<0, LT>), grid, block, 0, s, __VA_ARGS__); break;                 \
  }

extern "C" int64_t yt8m_moe_mix_xent_workspace_bytes(int64_t B, int64_t V) {
  if (B < 0 || V < 0) return 0;
  return (int64_t)sizeof(float) * ((B * V + 255) / 256 + 1);
}

extern "C" int yt8m_moe_mix_xent_fwd(const float* Zg, const float* Ze, const void* labels, int label_dtype, float* p,
                                     float* loss_out, int64_t B, int64_t V, int M, float eps, void* workspace,
                                     yt8m_stream_t stream) {
  YT8M_REQUIRE(M >= 1 && M <= MAXM, YT8M_E_BADARG, "num_mixtures must be in [1,16]");
  YT8M_REQUIRE(B > 0 && V > 0, YT8M_E_SHAPE, "empty batch: reduce_mean over 0 rows is undefined");
  YT8M_REQUIRE(Zg && Ze && labels && p && loss_out && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(label_dtype == YT8M_LABEL_U8 || label_dtype == YT8M_LABEL_F32, YT8M_E_BADARG, "label dtype");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t BV = B * V;
  dim3 grid((unsigned)((BV + 255) / 256)), block(256);
  float* partial = static_cast<float*>(workspace);
  if (label_dtype == YT8M_LABEL_U8) {
    YT8M_DISPATCH_M(moe_mix_xent_fwd_kernel, uint8_t, Zg, Ze, static_cast<const uint8_t*>(labels), p, partial, BV, M, eps)
  } else {
    YT8M_DISPATCH_M(moe_mix_xent_fwd_kernel, float, Zg, Ze, static_cast<const float*>(labels), p, partial, BV, M, eps)
  }
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(1024), 0, s, partial, (int64_t)grid.x, 1.0f / (float)B, loss_out);
  return launch_status("moe_mix_xent_fwd_kernel");
}

namespace {
int mix_xent_bwd_impl(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev, int64_t B, int64_t V, int M,
                      float eps, float upstream, unsigned* zmax, yt8m_stream_t stream);
}
extern "C" int yt8m_moe_mix_xent_bwd(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev,
                                     int64_t B, int64_t V, int M, float eps, float upstream, yt8m_stream_t stream) {
  return mix_xent_bwd_impl(Zg, Ze, labels, label_dtype, upstream_dev, B, V, M, eps, upstream, nullptr, stream);
}
// The same pass that also leaves max |dL/dZg| and max |dL/dZe| as float bits in absmax2[0..1] (zeroed here): the scale words the h2 form
// of the weight-gradient products x^T . dZ takes (yt8m_gemm_auto_grouped_ex) -- two memsets and two passes over 58 + 39 MB at B = 1024 less.
extern "C" int yt8m_moe_mix_xent_bwd_absmax(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev,
                                            int64_t B, int64_t V, int M, float eps, float upstream, void* absmax2, yt8m_stream_t stream) {
  YT8M_REQUIRE(absmax2, YT8M_E_BADARG, "null absmax words");
  YT8M_HIP_CHECK(hipMemsetAsync(absmax2, 0, 8, as_stream(stream)));
  return mix_xent_bwd_impl(Zg, Ze, labels, label_dtype, upstream_dev, B, V, M, eps, upstream, static_cast<unsigned*>(absmax2), stream);
}
namespace {
int mix_xent_bwd_impl(float* Zg, float* Ze, const void* labels, int label_dtype, const float* upstream_dev, int64_t B, int64_t V, int M,
                      float eps, float upstream, unsigned* zmax, yt8m_stream_t stream) {
  YT8M_REQUIRE(M >= 1 && M <= MAXM, YT8M_E_BADARG, "num_mixtures must be in [1,16]");
  YT8M_REQUIRE(B > 0 && V > 0, YT8M_E_SHAPE, "empty batch");
  YT8M_REQUIRE(Zg && Ze && labels, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(label_dtype == YT8M_LABEL_U8 || label_dtype == YT8M_LABEL_F32, YT8M_E_BADARG, "label dtype");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t BV = B * V;
  const int64_t per_wg = 256 * (zmax ? ZMAX_IT : 1);
  dim3 grid((unsigned)((BV + per_wg - 1) / per_wg)), block(256);
  const float dscale = upstream / (float)B;
  if (label_dtype == YT8M_LABEL_U8) {
    YT8M_DISPATCH_M(moe_mix_xent_bwd_kernel, uint8_t, Zg, Ze, static_cast<const uint8_t*>(labels), BV, M, eps, dscale, upstream_dev, zmax)
  } else {
    YT8M_DISPATCH_M(moe_mix_xent_bwd_kernel, float, Zg, Ze, static_cast<const float*>(labels), BV, M, eps, dscale, upstream_dev, zmax)
  }
  return launch_status("moe_mix_xent_bwd_kernel");
}
}  // namespace

extern "C" int yt8m_act_fwd_f32(int act, const float* x, float* y, int64_t n, yt8m_stream_t stream) {
  YT8M_REQUIRE(act >= 0 && act <= YT8M_ACT_ELU, YT8M_E_BADARG, "unknown activation");
  if (n <= 0) return n == 0 ? YT8M_OK : fail(YT8M_E_SHAPE, "yt8m_act_fwd_f32: negative n%s", "");
  YT8M_REQUIRE(x && y, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, act, x, y, n);
  return launch_status("act_fwd_kernel");
}

extern "C" int yt8m_act_bwd_f32(int act, const float* y, const float* dy, float* dx, int64_t n, yt8m_stream_t stream) {
  YT8M_REQUIRE(act >= 0 && act <= YT8M_ACT_ELU, YT8M_E_BADARG, "unknown activation");
  if (n <= 0) return n == 0 ? YT8M_OK : fail(YT8M_E_SHAPE, "yt8m_act_bwd_f32: negative n%s", "");
  YT8M_REQUIRE(y && dy && dx, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, act, y, dy, dx, n);
  return launch_status("act_bwd_kernel");
}

static bool cast_vec_ok(const float* src, int64_t rows, int64_t cols, int64_t ld, bool transposed) {
  return cols % 4 == 0 && ld % 4 == 0 && ((uintptr_t)src & 15) == 0 && (!transposed || rows % 4 == 0);
}

extern "C" int yt8m_cast_f32_bf16(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst, int64_t dst_ld,
                                  int transpose, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, YT8M_E_SHAPE, "bad shape");
  if (dst_ld == 0) dst_ld = transpose ? rows : cols;
  YT8M_REQUIRE(dst_ld >= (transpose ? rows : cols), YT8M_E_SHAPE, "dst_ld too small");
  if (rows * cols == 0) return YT8M_OK;
  YT8M_REQUIRE(src && dst, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  unsigned short* d = static_cast<unsigned short*>(dst);
  const bool dal = ((uintptr_t)dst & 7) == 0 && dst_ld % 4 == 0;
  if (transpose) {
    YT8M_REQUIRE((rows + 63) / 64 <= 65535, YT8M_E_SHAPE, "too many rows for the transposing cast");
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
    if (cast_vec_ok(src, rows, cols, ld, true) && dal)
      hipLaunchKernelGGL((cast_bf16_tile_kernel<false, true>), grid, dim3(256), 0, s, src, rows, cols, ld, (unsigned short*)nullptr,
                         (int64_t)0, d, dst_ld);
    else
      hipLaunchKernelGGL(cast_bf16_transpose_kernel, grid, dim3(256), 0, s, src, rows, cols, ld, d, dst_ld);
  } else if (cast_vec_ok(src, rows, cols, ld, false) && dal) {
    hipLaunchKernelGGL(cast_bf16_vec_kernel, dim3(grid_for(rows * (cols / 4), 256, 16384)), dim3(256), 0, s, src, rows, cols, ld, d,
                       dst_ld);
  } else {
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(rows * cols, 256, 16384)), dim3(256), 0, s, src, rows, cols, ld, d, dst_ld);
  }
  return launch_status("cast_bf16_kernel");
}

extern "C" int yt8m_cast_f32_bf16_dual(const float* src, int64_t rows, int64_t cols, int64_t ld, void* dst_plain,
                                       int64_t plain_ld, void* dst_trans, int64_t trans_ld, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, YT8M_E_SHAPE, "bad shape");
  if (plain_ld == 0) plain_ld = cols;
  if (trans_ld == 0) trans_ld = rows;
  YT8M_REQUIRE(plain_ld >= cols && trans_ld >= rows, YT8M_E_SHAPE, "destination leading dimension too small");
  if (rows * cols == 0) return YT8M_OK;
  YT8M_REQUIRE(src && dst_plain && dst_trans, YT8M_E_BADARG, "null operand");
  if (!(cast_vec_ok(src, rows, cols, ld, true) && (((uintptr_t)dst_plain | (uintptr_t)dst_trans) & 7) == 0 && plain_ld % 4 == 0 &&
        trans_ld % 4 == 0)) {
    int rc = yt8m_cast_f32_bf16(src, rows, cols, ld, dst_plain, plain_ld, 0, stream);    // unaligned / odd shapes: two passes
    return rc != YT8M_OK ? rc : yt8m_cast_f32_bf16(src, rows, cols, ld, dst_trans, trans_ld, 1, stream);
  }
  YT8M_REQUIRE((rows + 63) / 64 <= 65535, YT8M_E_SHAPE, "too many rows for the transposing cast");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL((cast_bf16_tile_kernel<true, true>), dim3((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), dim3(256), 0,
                     s, src, rows, cols, ld, static_cast<unsigned short*>(dst_plain), plain_ld, static_cast<unsigned short*>(dst_trans),
                     trans_ld);
  return launch_status("cast_bf16_tile_kernel");
}

extern "C" int64_t yt8m_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  (void)rows;
  return cols > 0 ? 256 * cols * (int64_t)sizeof(float) : 0;
}

// C[r][c] += scale * v[c] for every row r: the rank-1 remainder of a product whose operand is affine in an exact integer matrix
// (the layer-0 weight gradient on uint8 frames: beta * 1 (x) colsum(r (.) dz), applied once per step)
__global__ __launch_bounds__(256) void rank1_rows_kernel(float* __restrict__ C, int64_t rows, int64_t cols4, int64_t ldc,
                                                         const float* __restrict__ v, float scale) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * cols4) return;
  const int64_t r = e / cols4, c = (e - r * cols4) * 4;
  const float4 a = *reinterpret_cast<const float4*>(v + c);
  float4* p = reinterpret_cast<float4*>(C + r * ldc + c);
  float4 o = *p;
  o.x += scale * a.x; o.y += scale * a.y; o.z += scale * a.z; o.w += scale * a.w;
  *p = o;
}
extern "C" int yt8m_rank1_add_rows_f32(float* C, int64_t rows, int64_t cols, int64_t ldc, const float* v, float scale, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0 && ldc >= cols, YT8M_E_SHAPE, "bad shape");
  if (rows * cols == 0) return YT8M_OK;
  YT8M_REQUIRE(C && v, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((cols % 4) == 0 && (ldc % 4) == 0 && ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(v)) & 15) == 0, YT8M_E_SHAPE,
               "cols and ldc must be multiples of 4, operands 16-byte aligned");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t n = rows * (cols / 4);
  hipLaunchKernelGGL(rank1_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, rows, cols / 4, ldc, v, scale);
  return launch_status("rank1_rows_kernel");
}

// out[col] (+)= sum_r X[r][col] (beta 0 / 1) and out_weighted[col] = sum_r row_weights[r] X[r][col] (always overwritten) from one
// pass over X.  workspace: 2 * yt8m_colsum_workspace_bytes(rows, cols).
extern "C" int yt8m_colsum_weighted_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, const float* row_weights, float* out,
                                        float beta, float* out_weighted, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols, YT8M_E_SHAPE, "bad shape");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (cols == 0) return YT8M_OK;
  YT8M_REQUIRE(out && out_weighted && ((X && row_weights) || rows == 0), YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t colblocks = (cols + 63) / 64;
  int64_t nsplit = 1;
  if (workspace && colblocks < 512 && rows >= 4096) {
    nsplit = std::min<int64_t>(std::min<int64_t>(256, (1024 + colblocks - 1) / colblocks), rows / 1024);
    if (2 * nsplit * cols * (int64_t)sizeof(float) > workspace_bytes) nsplit = 1;
  }
  if (nsplit <= 1) {
    hipLaunchKernelGGL(colsum2_kernel, dim3((unsigned)colblocks), dim3(1024), 0, s, X, row_weights, rows, cols, ldx, out, out_weighted,
                       beta != 0.f ? 1 : 0, rows);
  } else {
    const int64_t rows_per = (rows + nsplit - 1) / nsplit;
    nsplit = (rows + rows_per - 1) / rows_per;
    float* p0 = static_cast<float*>(workspace);
    float* p1 = p0 + nsplit * cols;
    hipLaunchKernelGGL(colsum2_kernel, dim3((unsigned)colblocks, (unsigned)nsplit), dim3(1024), 0, s, X, row_weights, rows, cols, ldx, p0, p1,
                       0, rows_per);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, p0, (int)nsplit, cols, out,
                       beta != 0.f ? 1 : 0);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, p1, (int)nsplit, cols, out_weighted, 0);
  }
  return launch_status("colsum2_kernel");
}

extern "C" int yt8m_colsum_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, float* out, float beta, void* workspace,
                               int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols, YT8M_E_SHAPE, "bad shape");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (cols == 0) return YT8M_OK;
  YT8M_REQUIRE(out && (X || rows == 0), YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t colblocks = (cols + 63) / 64;
  // row split when the column blocks alone cannot fill the chip and there is enough work per block
  int64_t nsplit = 1;
  if (workspace && colblocks < 512 && rows >= 4096) {
    nsplit = std::min<int64_t>(std::min<int64_t>(256, (1024 + colblocks - 1) / colblocks), rows / 1024);
    if (nsplit * cols * (int64_t)sizeof(float) > workspace_bytes) nsplit = 1;
  }
  if (nsplit <= 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)colblocks), dim3(1024), 0, s, X, rows, cols, ldx, out, beta != 0.f ? 1 : 0,
                       rows);
  } else {
    const int64_t rows_per = (rows + nsplit - 1) / nsplit;
    nsplit = (rows + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)colblocks, (unsigned)nsplit), dim3(1024), 0, s, X, rows, cols, ldx,
                       static_cast<float*>(workspace), 0, rows_per);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), (int)nsplit, cols, out, beta != 0.f ? 1 : 0);
  }
  return launch_status("colsum_kernel");
}

extern "C" int64_t yt8m_xent_workspace_bytes(int64_t B, int64_t V) {
  if (B < 0 || V < 0) return 0;
  return (int64_t)sizeof(float) * (B * ((V + 1023) / 1024) + 1);
}

extern "C" int yt8m_xent_fwd_bwd(const float* p, const void* labels, int label_dtype, const float* weights,
                                 float* loss_out, float* dp, int64_t B, int64_t V, float eps, float upstream,
                                 void* workspace, yt8m_stream_t stream) {
  YT8M_REQUIRE(B > 0 && V > 0, YT8M_E_SHAPE, "empty batch: reduce_mean over 0 rows is undefined");
  YT8M_REQUIRE(B <= 65535, YT8M_E_SHAPE, "B > 65535");
  YT8M_REQUIRE(p && labels && loss_out && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(label_dtype == YT8M_LABEL_U8 || label_dtype == YT8M_LABEL_F32, YT8M_E_BADARG, "label dtype");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  float* partial = static_cast<float*>(workspace);
  dim3 grid((unsigned)((V + 1023) / 1024), (unsigned)B), block(256);
  const float dscale = upstream / (float)B;
  if (label_dtype == YT8M_LABEL_U8)
    hipLaunchKernelGGL((xent_kernel<uint8_t>), grid, block, 0, s, p, static_cast<const uint8_t*>(labels), weights, dp,
                       partial, V, eps, dscale, (const float*)nullptr);
  else
    hipLaunchKernelGGL((xent_kernel<float>), grid, block, 0, s, p, static_cast<const float*>(labels), weights, dp, partial,
                       V, eps, dscale, (const float*)nullptr);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, partial, (int64_t)grid.x * grid.y, 1.0f / (float)B, loss_out);
  return launch_status("xent_kernel");
}

extern "C" int yt8m_xent_bwd(const float* p, const void* labels, int label_dtype, const float* weights,
                             const float* upstream_dev, float* dp, int64_t B, int64_t V, float eps, float upstream,
                             yt8m_stream_t stream) {
  YT8M_REQUIRE(B > 0 && V > 0, YT8M_E_SHAPE, "empty batch");
  YT8M_REQUIRE(B <= 65535, YT8M_E_SHAPE, "B > 65535");
  YT8M_REQUIRE(p && labels && dp, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(label_dtype == YT8M_LABEL_U8 || label_dtype == YT8M_LABEL_F32, YT8M_E_BADARG, "label dtype");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  dim3 grid((unsigned)((V + 1023) / 1024), (unsigned)B), block(256);
  const float dscale = upstream / (float)B;
  if (label_dtype == YT8M_LABEL_U8)
    hipLaunchKernelGGL((xent_kernel<uint8_t>), grid, block, 0, s, p, static_cast<const uint8_t*>(labels), weights, dp,
                       (float*)nullptr, V, eps, dscale, upstream_dev);
  else
    hipLaunchKernelGGL((xent_kernel<float>), grid, block, 0, s, p, static_cast<const float*>(labels), weights, dp,
                       (float*)nullptr, V, eps, dscale, upstream_dev);
  return launch_status("xent_kernel(bwd)");
}

extern "C" int yt8m_l2norm_fwd_f32(const float* x, float* y, int64_t rows, int64_t cols, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0, YT8M_E_SHAPE, "negative dimension");
  if (rows * cols == 0) return YT8M_OK;
  YT8M_REQUIRE(x && y, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  if (cols >= 8192 && (cols & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && rows <= 65536)
    hipLaunchKernelGGL(l2norm_long_kernel<false>, dim3((unsigned)rows), dim3(1024), 0, s, x, (const float*)nullptr, y, cols, eps);
  else if (cols >= 512 && rows <= 16384)
    hipLaunchKernelGGL(l2norm_fwd_row_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, y, cols, eps);
  else
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, y, rows, cols, eps);
  return launch_status("l2norm_fwd_kernel");
}

extern "C" int yt8m_l2norm_bwd_f32(const float* x, const float* dy, float* dx, int64_t rows, int64_t cols, float eps,
                                   yt8m_stream_t stream) {
  YT8M_REQUIRE(rows >= 0 && cols >= 0, YT8M_E_SHAPE, "negative dimension");
  if (rows * cols == 0) return YT8M_OK;
  YT8M_REQUIRE(x && dy && dx, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  if (cols >= 8192 && (cols & 3) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0 && rows <= 65536)
    hipLaunchKernelGGL(l2norm_long_kernel<true>, dim3((unsigned)rows), dim3(1024), 0, s, x, dy, dx, cols, eps);
  else
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, dy, dx, rows, cols, eps);
  return launch_status("l2norm_bwd_kernel");
}

extern "C" int yt8m_dequant_l2norm_u8(const uint8_t* q, const int32_t* num_frames, float* x, int64_t B, int64_t F,
                                      int64_t D, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * D == 0) return YT8M_OK;
  YT8M_REQUIRE(q && x, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(dequant_l2norm_kernel, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, q, num_frames, x, B, F, D, eps);
  return launch_status("dequant_l2norm_kernel");
}

extern "C" int yt8m_dequant_mean_l2norm_u8(const uint8_t* q, const int32_t* num_frames, float* x, int64_t B, int64_t F,
                                           int64_t D, float eps, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * D == 0) return YT8M_OK;
  YT8M_REQUIRE(q && x, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(dequant_mean_l2norm_kernel, dim3((unsigned)B), dim3(256), 0, s, q, num_frames, x, F, D, eps);
  return launch_status("dequant_mean_l2norm_kernel");
}
