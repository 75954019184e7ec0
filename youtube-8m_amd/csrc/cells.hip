// cells.hip -- the two remaining recurrent cells of the reference's frame-level plugins, in the generic per-step form
// (one grouped-GEMM launch for the recurrent product + one pointwise kernel per phase; gfx950, wave64):
//   * tf.contrib.rnn.GRUCell                (W/all_frame_models/gru_pooling_model.py:34-38, gru_with_pooling_model.py:34-38)
//       [r|u] = sigmoid([x|h].W_gates + b_gates)   (b_gates initialised to 1)
//       c     = tanh([x | r*h].W_cand + b_cand)
//       h'    = u*h + (1-u)*c
//   * tf.contrib.rnn.LayerNormBasicLSTMCell (W/all_frame_models/layernorm_lstm_memory_model.py:37-50)
//       [i|j|f|o] = [x|h].W  (no bias);  each gate layer-normalised over its H units (gamma, beta; eps 1e-12 inside the rsqrt)
//       g = tanh(LN_j) [optionally tf.nn.dropout(g, keep_prob): "recurrent dropout without memory loss"]
//       c' = LN_state(c*sigmoid(LN_f + forget_bias) + sigmoid(LN_i)*g);   h' = tanh(c')*sigmoid(LN_o)
// both under tf.nn.dynamic_rnn: a row with t >= num_frames copies its state through and emits zeros.
// The input halves of the projections are hoisted out of the time loop by the caller (one MFMA GEMM over all steps);
// the time loop lives here.  These cells are not in any BASELINE configuration: this is the first correct path
// with two forms of the recurrent product: the packed-weight MFMA step kernels of lstm_fused.hip with the cell's pointwise
// block as their epilogue (H % 256 == 0: GRU 2 launches per forward step / 3 per backward step, LN-LSTM 2 / 2), else one
// grouped-GEMM launch per product.
#include "common.h"

namespace {

struct U4 { uint32_t x, y, z, w; };
// Philox4x32-10, identical to random.hip (element e -> word e & 3 of block e >> 2, key = seed)
__device__ __forceinline__ uint32_t philox_word(uint64_t e, uint64_t key) {
  const uint64_t ctr = e >> 2;
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  const int w = (int)(e & 3);
  return w == 0 ? c0 : w == 1 ? c1 : w == 2 ? c2 : c3;
}
__device__ __forceinline__ float keep_scale(uint64_t e, uint64_t seed, float keep) {     // 1/keep or 0
  if (keep >= 1.0f) return 1.0f;
  const float u = (float)(philox_word(e, seed) >> 8) * 5.9604644775390625e-8f;
  return (keep + u) >= 1.0f ? __fdiv_rn(1.0f, keep) : 0.0f;
}

// ================================================ GRU ======================================================================
__global__ __launch_bounds__(256) void gru_gates_kernel(float* __restrict__ zg, const float* __restrict__ h_prev,
                                                        float* __restrict__ rh, int64_t B, int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  float* zr = zg + b * 2 * H;
  const float r = sigmoidf_(zr[h]), u = sigmoidf_(zr[H + h]);
  zr[h] = r; zr[H + h] = u;
  rh[idx] = r * h_prev[idx];
}

__global__ __launch_bounds__(256) void gru_out_kernel(float* __restrict__ zc, const float* __restrict__ zg,
                                                      const float* __restrict__ h_prev, float* __restrict__ h_new,
                                                      float* __restrict__ out, const int32_t* __restrict__ nf, int t, int64_t B,
                                                      int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  const float c = tanhf(zc[idx]);
  zc[idx] = c;
  const bool live = nf ? (t < nf[b]) : true;
  const float u = zg[b * 2 * H + H + h], hp = h_prev[idx];
  const float hn = live ? u * hp + (1.0f - u) * c : hp;
  h_new[idx] = hn;
  if (out) out[idx] = live ? hn : 0.f;
}

// phase 1 of the step backward: everything that does not need d(r*h)
__global__ __launch_bounds__(256) void gru_bwd1_kernel(const float* __restrict__ zg, const float* __restrict__ zc,
                                                       const float* __restrict__ h_prev, const float* __restrict__ dh_cur,
                                                       const float* __restrict__ dout, float* __restrict__ dzg,
                                                       float* __restrict__ dzc, float* __restrict__ dh_prev,
                                                       const int32_t* __restrict__ nf, int t, int64_t B, int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  const bool live = nf ? (t < nf[b]) : true;
  if (!live) {
    dzg[b * 2 * H + H + h] = 0.f;
    dzc[idx] = 0.f;
    dh_prev[idx] = dh_cur[idx];
    return;
  }
  const float u = zg[b * 2 * H + H + h], c = zc[idx];
  const float dh = dh_cur[idx] + (dout ? dout[idx] : 0.f);
  dzg[b * 2 * H + H + h] = dh * (h_prev[idx] - c) * u * (1.0f - u);
  dzc[idx] = dh * (1.0f - u) * (1.0f - c * c);
  dh_prev[idx] = dh * u;
}

// phase 2: drh = dzc . Wc_h^T is known
__global__ __launch_bounds__(256) void gru_bwd2_kernel(const float* __restrict__ zg, const float* __restrict__ h_prev,
                                                       const float* __restrict__ drh, float* __restrict__ dzg,
                                                       float* __restrict__ dh_prev, const int32_t* __restrict__ nf, int t,
                                                       int64_t B, int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  const bool live = nf ? (t < nf[b]) : true;
  if (!live) { dzg[b * 2 * H + h] = 0.f; return; }
  const float r = zg[b * 2 * H + h], d = drh[idx];
  dzg[b * 2 * H + h] = d * h_prev[idx] * r * (1.0f - r);
  dh_prev[idx] += d * r;
}

// ========================================= LayerNormBasicLSTMCell ==========================================================
// Gate non-linearities on v_exp_f32 / v_rcp_f32, as the persistent recurrences use them (<= ~1.5e-7 absolute per value against libm's
// expf / tanhf, whose range checks and branches were most of these kernels' ~2 400 instructions on ONE wave per SIMD: round 6).
__device__ __forceinline__ float ln_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float ln_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
constexpr int LN_PT = 8;                       // units per thread: H <= 256 * LN_PT
constexpr float LN_EPS = 1e-12f;               // tf.contrib.layers.layer_norm variance_epsilon

struct F4 { float v[4]; };
__device__ __forceinline__ F4 block_sum4_256(F4 a, float* red /* 16 floats */) {
#pragma unroll
  for (int k = 0; k < 4; ++k) a.v[k] = wave_sum(a.v[k]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red[w * 4 + k] = a.v[k];
  }
  __syncthreads();
  F4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o.v[k] = red[k] + red[4 + k] + red[8 + k] + red[12 + k];
  return o;
}

// one workgroup per batch row.  z row [4H] stays RAW (pre-normalisation): backward recomputes from it and `stats`.
// Round 6: the kernel is a chain of dependent round trips on one wave per SIMD (128 workgroups at B = 128), so EVERYTHING it reads --
// z, gamma / beta, c_prev, h_prev, num_frames -- is requested before the first reduction (it used to fetch gamma / beta / c_prev
// behind the second one and the row's liveness in front of everything: 24 us per step against a 5 us launch floor), PT is the
// compile-time units per thread (H <= 256 PT; the H = 1024 plugin shape runs PT = 4, not 8 half-masked slots), and the two
// single-value reductions of the state normalisation ride on one barrier pair each.
template <int PT>
__global__ __launch_bounds__(256) void lnlstm_fwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ stats,
                                                         const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                                         float* __restrict__ c_new, float* __restrict__ h_new,
                                                         float* __restrict__ out, const int32_t* __restrict__ nf, int t,
                                                         int64_t B, int64_t H, float fb, float keep, uint64_t seed) {
  __shared__ float red[16];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* zr = z + b * 4 * H;
  const float invH = 1.0f / (float)H;
  float zv[4][PT], gm[5][PT], bt[5][PT], cpv[PT], hpv[PT];
  const int nfb = nf ? nf[b] : 0x7fffffff;
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    const bool in = h < H;
#pragma unroll
    for (int g = 0; g < 4; ++g) zv[g][p] = in ? zr[g * H + h] : 0.f;
#pragma unroll
    for (int g = 0; g < 5; ++g) { gm[g][p] = in ? gamma[g * H + h] : 0.f; bt[g][p] = in ? beta[g * H + h] : 0.f; }
    cpv[p] = in ? c_prev[b * H + h] : 0.f;
    hpv[p] = in ? h_prev[b * H + h] : 0.f;
  }
  if (t >= nfb) {                                                       // dynamic_rnn copy-through (block-uniform)
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int64_t h = tid + p * 256;
      if (h < H) {
        c_new[b * H + h] = cpv[p];
        h_new[b * H + h] = hpv[p];
        if (out) out[b * H + h] = 0.f;
      }
    }
    return;
  }
  F4 s = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g) s.v[g] += zv[g][p];
  F4 mean = block_sum4_256(s, red);
#pragma unroll
  for (int g = 0; g < 4; ++g) mean.v[g] *= invH;
  F4 q = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) { const float d = zv[g][p] - mean.v[g]; if (h < H) q.v[g] += d * d; }
  }
  F4 rstd = block_sum4_256(q, red);
#pragma unroll
  for (int g = 0; g < 4; ++g) rstd.v[g] = rsqrtf(rstd.v[g] * invH + LN_EPS);
  float cp[PT], og[PT];
  float sc = 0.f;
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    cp[p] = 0.f; og[p] = 0.f;
    if (h < H) {
      float y[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) y[g] = (zv[g][p] - mean.v[g]) * rstd.v[g] * gm[g][p] + bt[g][p];
      const float i = ln_sigmoid(y[0]);
      const float gg = ln_tanh(y[1]) * keep_scale((uint64_t)(((int64_t)t * B + b) * H + h), seed, keep);
      const float f = ln_sigmoid(y[2] + fb);
      og[p] = ln_sigmoid(y[3]);
      cp[p] = cpv[p] * f + i * gg;
      sc += cp[p];
    }
  }
  const float mean_s = block_sum_256(sc, red) * invH;
  float qs = 0.f;
#pragma unroll
  for (int p = 0; p < PT; ++p) { const int64_t h = tid + p * 256; const float d = cp[p] - mean_s; if (h < H) qs += d * d; }
  const float rstd_s = rsqrtf(block_sum_256(qs, red) * invH + LN_EPS);
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    if (h < H) {
      const float cn = (cp[p] - mean_s) * rstd_s * gm[4][p] + bt[4][p];
      const float hn = ln_tanh(cn) * og[p];
      c_new[b * H + h] = cn;
      h_new[b * H + h] = hn;
      if (out) out[b * H + h] = hn;
    }
  }
  if (tid < 4) { stats[b * 10 + tid * 2] = mean.v[tid]; stats[b * 10 + tid * 2 + 1] = rstd.v[tid]; }
  if (tid == 4) { stats[b * 10 + 8] = mean_s; stats[b * 10 + 9] = rstd_s; }
}

// dyb[b, 5H] = gradient w.r.t. the five layer-norm OUTPUTS (-> dbeta by column sum), dyg = dyb * normalised input
// (-> dgamma by column sum), dz[b,4H] = gradient w.r.t. the raw pre-activations.
template <int PT>
__global__ __launch_bounds__(256) void lnlstm_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ stats,
                                                         const float* __restrict__ c_prev, const float* __restrict__ c_new,
                                                         const float* __restrict__ dh_cur, const float* __restrict__ dc_cur,
                                                         const float* __restrict__ dout, float* __restrict__ dz,
                                                         float* __restrict__ dyb, float* __restrict__ dyg,
                                                         float* __restrict__ dc_prev, float* __restrict__ dh_prev,
                                                         const int32_t* __restrict__ nf, int t, int64_t B, int64_t H, float fb,
                                                         float keep, uint64_t seed) {
  __shared__ float red[16];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const float* zr = z + b * 4 * H;
  const float invH = 1.0f / (float)H;
  // every operand is requested up front (one round trip, not a chain of them behind the liveness word)
  const int nfb = nf ? nf[b] : 0x7fffffff;
  float mean[5], rstd[5];
#pragma unroll
  for (int g = 0; g < 5; ++g) { mean[g] = stats[b * 10 + 2 * g]; rstd[g] = stats[b * 10 + 2 * g + 1]; }
  float zv[4][PT], gm[5][PT], bt[4][PT], cpv[PT], cnv[PT], dhv[PT], dcv[PT];
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    const bool in = h < H;
#pragma unroll
    for (int g = 0; g < 4; ++g) { zv[g][p] = in ? zr[g * H + h] : 0.f; bt[g][p] = in ? beta[g * H + h] : 0.f; }
#pragma unroll
    for (int g = 0; g < 5; ++g) gm[g][p] = in ? gamma[g * H + h] : 0.f;
    cpv[p] = in ? c_prev[b * H + h] : 0.f;
    cnv[p] = in ? c_new[b * H + h] : 0.f;
    dhv[p] = in ? dh_cur[b * H + h] : 0.f;
    dcv[p] = in ? dc_cur[b * H + h] : 0.f;
    if (dout && in && t < nfb) dhv[p] += dout[b * H + h];
  }
  if (t >= nfb) {
#pragma unroll
    for (int p = 0; p < PT; ++p) {
      const int64_t h = tid + p * 256;
      if (h < H) {
#pragma unroll
        for (int g = 0; g < 4; ++g) dz[b * 4 * H + g * H + h] = 0.f;
#pragma unroll
        for (int g = 0; g < 5; ++g) { dyb[b * 5 * H + g * H + h] = 0.f; dyg[b * 5 * H + g * H + h] = 0.f; }
        dc_prev[b * H + h] = dcv[p];
        dh_prev[b * H + h] = dhv[p];
      }
    }
    return;
  }
  float n[4][PT], dy[4][PT], ns[PT], dns[PT], fv[PT], iv[PT], gv[PT], tj[PT], ks[PT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    ns[p] = dns[p] = 0.f;
    fv[p] = iv[p] = gv[p] = tj[p] = ks[p] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { n[g][p] = 0.f; dy[g][p] = 0.f; }
    if (h < H) {
      float y[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        n[g][p] = (zv[g][p] - mean[g]) * rstd[g];
        y[g] = n[g][p] * gm[g][p] + bt[g][p];
      }
      iv[p] = ln_sigmoid(y[0]);
      tj[p] = ln_tanh(y[1]);
      ks[p] = keep_scale((uint64_t)(((int64_t)t * B + b) * H + h), seed, keep);
      gv[p] = tj[p] * ks[p];
      fv[p] = ln_sigmoid(y[2] + fb);
      const float o = ln_sigmoid(y[3]);
      const float cpre = cpv[p] * fv[p] + iv[p] * gv[p];
      ns[p] = (cpre - mean[4]) * rstd[4];
      const float tc = ln_tanh(cnv[p]);
      const float dh = dhv[p];
      dy[3][p] = dh * tc * o * (1.0f - o);
      const float dcn = dcv[p] + dh * o * (1.0f - tc * tc);
      dyb[b * 5 * H + 4 * H + h] = dcn;
      dyg[b * 5 * H + 4 * H + h] = dcn * ns[p];
      dns[p] = dcn * gm[4][p];
      s1 += dns[p];
      s2 += dns[p] * ns[p];
    }
  }
  F4 a = {{s1, s2, 0.f, 0.f}};
  a = block_sum4_256(a, red);
  const float m1 = a.v[0] * invH, m2 = a.v[1] * invH;
  F4 t1 = {{0.f, 0.f, 0.f, 0.f}}, t2 = {{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    if (h < H) {
      const float dcp = rstd[4] * (dns[p] - m1 - ns[p] * m2);
      dc_prev[b * H + h] = dcp * fv[p];
      dh_prev[b * H + h] = 0.f;
      dy[0][p] = dcp * gv[p] * iv[p] * (1.0f - iv[p]);
      dy[1][p] = dcp * iv[p] * ks[p] * (1.0f - tj[p] * tj[p]);
      dy[2][p] = dcp * cpv[p] * fv[p] * (1.0f - fv[p]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        dyb[b * 5 * H + g * H + h] = dy[g][p];
        dyg[b * 5 * H + g * H + h] = dy[g][p] * n[g][p];
        dy[g][p] *= gm[g][p];                         // now d(normalised)
        t1.v[g] += dy[g][p];
        t2.v[g] += dy[g][p] * n[g][p];
      }
    }
  }
  t1 = block_sum4_256(t1, red);
  t2 = block_sum4_256(t2, red);
#pragma unroll
  for (int p = 0; p < PT; ++p) {
    const int64_t h = tid + p * 256;
    if (h < H) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        dz[b * 4 * H + g * H + h] = rstd[g] * (dy[g][p] - t1.v[g] * invH - n[g][p] * t2.v[g] * invH);
    }
  }
}

}  // namespace

using namespace yt8m;

namespace yt8m {  // lstm_fused.hip: packed-weight step products (one launch per product, epilogue fused)
bool cell_packed_supported(int64_t B, int64_t H);
int cell_pack(const float* Wh, int64_t ldw, float* Wp, float* Wq, int64_t H, int G, hipStream_t s);
int cell_step_add4(float* z, const float* Wp, const float* h_prev, int64_t B, int64_t H, hipStream_t s);
int gru_step_gates(float* zg, const float* Wp_g, const float* h_prev, float* rh, int64_t B, int64_t H, hipStream_t s);
int gru_step_cand(float* zc, const float* Wp_c, const float* rh, const float* zg, const float* h_prev, float* h_new, float* out,
                  const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s);
int cell_step_bwd(const float* dz, const float* Wq, float* dh_prev, int64_t B, int64_t H, int G, hipStream_t s);
int gru_step_bwd_cand(const float* dzc, const float* Wq_c, float* dh_prev, const float* zg, const float* h_prev, float* dzg,
                      const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s);
}  // namespace yt8m

static bool packed_ok(int64_t B, int64_t H, int G, void* ws, int64_t ws_bytes) {
  static const bool off = getenv("YT8M_NO_PACKED_CELLS") != nullptr;      // A/B switch for tools/ and tests
  return !off && ws && cell_packed_supported(B, H) && ws_bytes >= (int64_t)sizeof(float) * H * G * H;
}

extern "C" int yt8m_gru_layer_fwd(float* zg, float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc, float* hs,
                                  float* rh, float* out, const int32_t* num_frames, int64_t F, int64_t B, int64_t H,
                                  void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(zg && zc && Wg_h && Wc_h && hs && rh, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldg >= 2 * H && ldc >= H, YT8M_E_SHAPE, "leading dimension too small");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H;
  const dim3 grid((unsigned)((BH + 255) / 256));
  if (packed_ok(B, H, 3, gemm_workspace, gemm_workspace_bytes)) {
    // packed path: 2 launches per step (r|u product + sigmoid + r*h;  candidate product + tanh + blend + copy-through)
    ProfScope prof(F_LSTM, s);
    float* Wp_g = static_cast<float*>(gemm_workspace);
    float* Wp_c = Wp_g + H * 2 * H;
    int rc = cell_pack(Wg_h, ldg, Wp_g, nullptr, H, 2, s);
    if (rc == YT8M_OK) rc = cell_pack(Wc_h, ldc, Wp_c, nullptr, H, 1, s);
    for (int64_t t = 0; t < F && rc == YT8M_OK; ++t) {
      rc = gru_step_gates(zg + t * B * 2 * H, Wp_g, hs + t * BH, rh + t * BH, B, H, s);
      if (rc == YT8M_OK)
        rc = gru_step_cand(zc + t * BH, Wp_c, rh + t * BH, zg + t * B * 2 * H, hs + t * BH, hs + (t + 1) * BH,
                           out ? out + t * BH : nullptr, num_frames, (int)t, B, H, s);
    }
    return rc;
  }
  for (int64_t t = 0; t < F; ++t) {
    float* zgt = zg + t * B * 2 * H;
    float* zct = zc + t * BH;
    float* rht = rh + t * BH;
    yt8m_gemm_problem pg = {B, 2 * H, H, hs + t * BH, H, Wg_h, ldg, zgt, 2 * H, nullptr, 1.0f};      // zg_t += h . Wg_h
    int rc = yt8m_gemm_f32_grouped(0, 0, 1, &pg, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    {
      ProfScope prof(F_LSTM, s);
      hipLaunchKernelGGL(gru_gates_kernel, grid, dim3(256), 0, s, zgt, hs + t * BH, rht, B, H);
    }
    yt8m_gemm_problem pc = {B, H, H, rht, H, Wc_h, ldc, zct, H, nullptr, 1.0f};                       // zc_t += (r*h) . Wc_h
    rc = yt8m_gemm_f32_grouped(0, 0, 1, &pc, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    {
      ProfScope prof(F_LSTM, s);
      hipLaunchKernelGGL(gru_out_kernel, grid, dim3(256), 0, s, zct, zgt, hs + t * BH, hs + (t + 1) * BH,
                         out ? out + t * BH : nullptr, num_frames, (int)t, B, H);
    }
  }
  return launch_status("gru step kernels");
}

extern "C" int yt8m_gru_layer_bwd(const float* zg, const float* zc, const float* Wg_h, int64_t ldg, const float* Wc_h, int64_t ldc,
                                  const float* hs, const float* dout, const float* dh_final, float* dzg, float* dzc, float* work,
                                  const int32_t* num_frames, int64_t F, int64_t B, int64_t H, void* gemm_workspace,
                                  int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(zg && zc && Wg_h && Wc_h && hs && dzg && dzc && work, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldg >= 2 * H && ldc >= H, YT8M_E_SHAPE, "leading dimension too small");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H;
  float* dh_cur = work;
  float* dh_prev = work + BH;
  float* drh = work + 2 * BH;
  if (dh_final) YT8M_HIP_CHECK(hipMemcpyAsync(dh_cur, dh_final, BH * sizeof(float), hipMemcpyDeviceToDevice, s));
  else YT8M_HIP_CHECK(hipMemsetAsync(dh_cur, 0, BH * sizeof(float), s));
  const dim3 grid((unsigned)((BH + 255) / 256));
  const bool packed = packed_ok(B, H, 3, gemm_workspace, gemm_workspace_bytes);
  float* Wq_g = static_cast<float*>(gemm_workspace);
  float* Wq_c = packed ? Wq_g + H * 2 * H : nullptr;
  if (packed) {
    ProfScope prof(F_LSTM, s);
    int rc = cell_pack(Wg_h, ldg, nullptr, Wq_g, H, 2, s);
    if (rc == YT8M_OK) rc = cell_pack(Wc_h, ldc, nullptr, Wq_c, H, 1, s);
    if (rc != YT8M_OK) return rc;
  }
  for (int64_t t = F - 1; t >= 0; --t) {
    const float* zgt = zg + t * B * 2 * H;
    float* dzgt = dzg + t * B * 2 * H;
    float* dzct = dzc + t * BH;
    if (packed) {                                  // 3 launches per step
      ProfScope prof(F_LSTM, s);
      hipLaunchKernelGGL(gru_bwd1_kernel, grid, dim3(256), 0, s, zgt, zc + t * BH, hs + t * BH, dh_cur,
                         dout ? dout + t * BH : nullptr, dzgt, dzct, dh_prev, num_frames, (int)t, B, H);
      int rc = gru_step_bwd_cand(dzct, Wq_c, dh_prev, zgt, hs + t * BH, dzgt, num_frames, (int)t, B, H, s);
      if (rc == YT8M_OK) rc = cell_step_bwd(dzgt, Wq_g, dh_prev, B, H, 2, s);
      if (rc != YT8M_OK) return rc;
      float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
      continue;
    }
    {
      ProfScope prof(F_LSTM, s);
      hipLaunchKernelGGL(gru_bwd1_kernel, grid, dim3(256), 0, s, zgt, zc + t * BH, hs + t * BH, dh_cur,
                         dout ? dout + t * BH : nullptr, dzgt, dzct, dh_prev, num_frames, (int)t, B, H);
    }
    yt8m_gemm_problem pc = {B, H, H, dzct, H, Wc_h, ldc, drh, H, nullptr, 0.0f};                       // drh = dzc . Wc_h^T
    int rc = yt8m_gemm_f32_grouped(0, 1, 1, &pc, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    {
      ProfScope prof(F_LSTM, s);
      hipLaunchKernelGGL(gru_bwd2_kernel, grid, dim3(256), 0, s, zgt, hs + t * BH, drh, dzgt, dh_prev, num_frames, (int)t, B, H);
    }
    yt8m_gemm_problem pg = {B, H, 2 * H, dzgt, 2 * H, Wg_h, ldg, dh_prev, H, nullptr, 1.0f};           // dh_prev += dzg . Wg_h^T
    rc = yt8m_gemm_f32_grouped(0, 1, 1, &pg, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
  }
  return launch_status("gru backward step kernels");
}

extern "C" int yt8m_lnlstm_layer_fwd(float* z, const float* Wh, int64_t ldw, const float* gamma, const float* beta, float* stats,
                                     float* cs, float* hs, float* out, const int32_t* num_frames, int64_t F, int64_t B, int64_t H,
                                     float forget_bias, float keep_prob, uint64_t seed, void* gemm_workspace,
                                     int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(H <= 256 * LN_PT, YT8M_E_SHAPE, "layer-norm LSTM supports H <= 2048");
  YT8M_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, YT8M_E_BADARG, "keep_prob must be in (0, 1]");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wh && gamma && beta && stats && cs && hs, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H;
  const bool packed = packed_ok(B, H, 4, gemm_workspace, gemm_workspace_bytes);
  float* Wp = static_cast<float*>(gemm_workspace);
  if (packed) {
    ProfScope prof(F_LSTM, s);
    int rc = cell_pack(Wh, ldw, Wp, nullptr, H, 4, s);
    if (rc != YT8M_OK) return rc;
  }
  for (int64_t t = 0; t < F; ++t) {
    float* zt = z + t * B * 4 * H;
    int rc;
    if (packed) {
      ProfScope prof(F_LSTM, s);
      rc = cell_step_add4(zt, Wp, hs + t * BH, B, H, s);
    } else {
      yt8m_gemm_problem pr = {B, 4 * H, H, hs + t * BH, H, Wh, ldw, zt, 4 * H, nullptr, 1.0f};
      rc = yt8m_gemm_f32_grouped(0, 0, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
    }
    if (rc != YT8M_OK) return rc;
    ProfScope prof(F_LSTM, s);
    auto kfn = H <= 512 ? lnlstm_fwd_kernel<2> : H <= 1024 ? lnlstm_fwd_kernel<4> : lnlstm_fwd_kernel<LN_PT>;
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(256), 0, s, zt, gamma, beta, stats + t * B * 10, cs + t * BH,
                       hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH, out ? out + t * BH : nullptr, num_frames, (int)t, B, H,
                       forget_bias, keep_prob, seed);
  }
  return launch_status("lnlstm_fwd_kernel");
}

extern "C" int yt8m_lnlstm_layer_bwd(const float* z, const float* Wh, int64_t ldw, const float* gamma, const float* beta,
                                     const float* stats, const float* cs, const float* dout, const float* dc_final,
                                     const float* dh_final, float* dz, float* dyb, float* dyg, float* work,
                                     const int32_t* num_frames, int64_t F, int64_t B, int64_t H, float forget_bias,
                                     float keep_prob, uint64_t seed, void* gemm_workspace, int64_t gemm_workspace_bytes,
                                     yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(H <= 256 * LN_PT, YT8M_E_SHAPE, "layer-norm LSTM supports H <= 2048");
  YT8M_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, YT8M_E_BADARG, "keep_prob must be in (0, 1]");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wh && gamma && beta && stats && cs && dz && dyb && dyg && work, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H;
  float* dh_cur = work;
  float* dc_cur = work + BH;
  float* dh_prev = work + 2 * BH;
  float* dc_prev = work + 3 * BH;
  if (dh_final) YT8M_HIP_CHECK(hipMemcpyAsync(dh_cur, dh_final, BH * sizeof(float), hipMemcpyDeviceToDevice, s));
  else YT8M_HIP_CHECK(hipMemsetAsync(dh_cur, 0, BH * sizeof(float), s));
  if (dc_final) YT8M_HIP_CHECK(hipMemcpyAsync(dc_cur, dc_final, BH * sizeof(float), hipMemcpyDeviceToDevice, s));
  else YT8M_HIP_CHECK(hipMemsetAsync(dc_cur, 0, BH * sizeof(float), s));
  const bool packed = packed_ok(B, H, 4, gemm_workspace, gemm_workspace_bytes);
  float* Wq = static_cast<float*>(gemm_workspace);
  if (packed) {
    ProfScope prof(F_LSTM, s);
    int rc = cell_pack(Wh, ldw, nullptr, Wq, H, 4, s);
    if (rc != YT8M_OK) return rc;
  }
  for (int64_t t = F - 1; t >= 0; --t) {
    float* dzt = dz + t * B * 4 * H;
    {
      ProfScope prof(F_LSTM, s);
      auto kfn = H <= 512 ? lnlstm_bwd_kernel<2> : H <= 1024 ? lnlstm_bwd_kernel<4> : lnlstm_bwd_kernel<LN_PT>;
      hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(256), 0, s, z + t * B * 4 * H, gamma, beta,
                         stats + t * B * 10, cs + t * BH, cs + (t + 1) * BH, dh_cur, dc_cur, dout ? dout + t * BH : nullptr, dzt,
                         dyb + t * B * 5 * H, dyg + t * B * 5 * H, dc_prev, dh_prev, num_frames, (int)t, B, H, forget_bias,
                         keep_prob, seed);
    }
    int rc;
    if (packed) {
      ProfScope prof(F_LSTM, s);
      rc = cell_step_bwd(dzt, Wq, dh_prev, B, H, 4, s);
    } else {
      yt8m_gemm_problem pr = {B, H, 4 * H, dzt, 4 * H, Wh, ldw, dh_prev, H, nullptr, 1.0f};
      rc = yt8m_gemm_f32_grouped(0, 1, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
    }
    if (rc != YT8M_OK) return rc;
    float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
    tmp = dc_cur; dc_cur = dc_prev; dc_prev = tmp;
  }
  return launch_status("lnlstm_bwd_kernel");
}
