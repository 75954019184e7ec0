// sequence.hip -- frame-axis kernels: BasicLSTM gate block, attention / assignment softmaxes, per-row top-k.
// (gfx950, wave64; all HBM/latency-bound pointwise or short-reduction work, wave-shuffle reductions.)
#include "common.h"

namespace {

// ---- BasicLSTMCell pointwise block with dynamic_rnn copy-through (A.5) ----------------
__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(float* __restrict__ z, const float* __restrict__ c_prev,
                                                             const float* __restrict__ h_prev, float* __restrict__ c_new,
                                                             float* __restrict__ h_new, float* __restrict__ out,
                                                             const int32_t* __restrict__ nf, int t, int64_t B, int64_t H,
                                                             float fb) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  const bool live = nf ? (t < nf[b]) : true;
  float* zr = z + b * 4 * H;
  if (!live) {
    c_new[idx] = c_prev[idx];
    h_new[idx] = h_prev[idx];
    if (out) out[idx] = 0.f;
    return;
  }
  const float i = sigmoidf_(zr[h]);
  const float j = tanhf(zr[H + h]);
  const float f = sigmoidf_(zr[2 * H + h] + fb);
  const float o = sigmoidf_(zr[3 * H + h]);
  const float c = c_prev[idx] * f + i * j;
  const float hn = tanhf(c) * o;
  zr[h] = i; zr[H + h] = j; zr[2 * H + h] = f; zr[3 * H + h] = o;
  c_new[idx] = c;
  h_new[idx] = hn;
  if (out) out[idx] = hn;
}

__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                             const float* __restrict__ c_new, const float* __restrict__ dh,
                                                             const float* __restrict__ dc, const float* __restrict__ dout,
                                                             float* __restrict__ dz, float* __restrict__ dc_prev,
                                                             float* __restrict__ dh_prev, const int32_t* __restrict__ nf,
                                                             int t, int64_t B, int64_t H) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int64_t b = idx / H, h = idx - b * H;
  const bool live = nf ? (t < nf[b]) : true;
  float* dzr = dz + b * 4 * H;
  if (!live) {
    dzr[h] = 0.f; dzr[H + h] = 0.f; dzr[2 * H + h] = 0.f; dzr[3 * H + h] = 0.f;
    dc_prev[idx] = dc[idx];
    dh_prev[idx] = dh[idx];
    return;
  }
  const float* gr = gates + b * 4 * H;
  const float i = gr[h], j = gr[H + h], f = gr[2 * H + h], o = gr[3 * H + h];
  const float tc = tanhf(c_new[idx]);
  const float dht = dh[idx] + (dout ? dout[idx] : 0.f);
  const float dct = dc[idx] + dht * o * (1.0f - tc * tc);
  dzr[h] = dct * j * i * (1.0f - i);
  dzr[H + h] = dct * i * (1.0f - j * j);
  dzr[2 * H + h] = dct * c_prev[idx] * f * (1.0f - f);
  dzr[3 * H + h] = dht * tc * o * (1.0f - o);
  dc_prev[idx] = dct * f;
  dh_prev[idx] = 0.f;
}

// ---- attention weights: softmax over FRAMES, mask, renormalise (lstm_attention_max_pooling_model.py:59-60) --
// One wave per (b, a).  w_f = mask_f e_f / sum_valid e,  e = exp(act - max over ALL F frames).
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(const float* __restrict__ act, const int32_t* __restrict__ nf,
                                                               float* __restrict__ w, int64_t B, int64_t F, int64_t A) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= B * A) return;
  const int64_t b = pair / A, a = pair - b * A;
  const int64_t n = nf ? (int64_t)nf[b] : F;
  const float* ar = act + b * F * A + a;
  float mx = -INFINITY;
  for (int64_t f = lane; f < F; f += 64) mx = fmaxf(mx, ar[f * A]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int64_t f = lane; f < F && f < n; f += 64) s += expf(ar[f * A] - mx);
  s = wave_sum(s);
  float* wr = w + b * F * A + a;
  for (int64_t f = lane; f < F; f += 64) wr[f * A] = (f < n) ? expf(ar[f * A] - mx) / s : (n > 0 ? 0.f : NAN);
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw,
                                                               const int32_t* __restrict__ nf, float* __restrict__ dact,
                                                               int64_t B, int64_t F, int64_t A) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= B * A) return;
  const int64_t b = pair / A, a = pair - b * A;
  const int64_t n = nf ? (int64_t)nf[b] : F;
  const float* wr = w + b * F * A + a;
  const float* gr = dw + b * F * A + a;
  float dot = 0.f;
  for (int64_t f = lane; f < F && f < n; f += 64) dot += wr[f * A] * gr[f * A];
  dot = wave_sum(dot);
  float* dr = dact + b * F * A + a;
  for (int64_t f = lane; f < F; f += 64) dr[f * A] = (f < n) ? wr[f * A] * (gr[f * A] - dot) : 0.f;
}

// ---- NetVLAD soft-assignment: softmax over the K clusters of each frame row, times the frame mask --------
// One wave per row (K <= 4096); lanes stride over K.
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ s, const int32_t* __restrict__ nf,
                                                               float* __restrict__ a, int64_t rows, int64_t F, int64_t K) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / F, f = row - b * F;
  float* ar = a + row * K;
  if (nf && f >= (int64_t)nf[b]) {
    for (int64_t k = lane; k < K; k += 64) ar[k] = 0.f;
    return;
  }
  const float* sr = s + row * K;
  float mx = -INFINITY;
  for (int64_t k = lane; k < K; k += 64) mx = fmaxf(mx, sr[k]);
  mx = wave_max(mx);
  float den = 0.f;
  for (int64_t k = lane; k < K; k += 64) den += expf(sr[k] - mx);
  den = wave_sum(den);
  const float inv = 1.0f / den;
  for (int64_t k = lane; k < K; k += 64) ar[k] = expf(sr[k] - mx) * inv;
}

__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ a, const float* __restrict__ da,
                                                               const int32_t* __restrict__ nf, float* __restrict__ ds,
                                                               int64_t rows, int64_t F, int64_t K) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / F, f = row - b * F;
  float* dr = ds + row * K;
  if (nf && f >= (int64_t)nf[b]) {
    for (int64_t k = lane; k < K; k += 64) dr[k] = 0.f;
    return;
  }
  const float* ar = a + row * K;
  const float* gr = da + row * K;
  float dot = 0.f;
  for (int64_t k = lane; k < K; k += 64) dot += ar[k] * gr[k];
  dot = wave_sum(dot);
  for (int64_t k = lane; k < K; k += 64) dr[k] = ar[k] * (gr[k] - dot);
}

// ---- per-row top-k (eval_util.py:158-165 top_k_triplets): row staged in LDS, k rounds of arg-max -----------
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ p, int64_t V, int k, float* __restrict__ vals,
                                                        int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  __shared__ float rv[4];
  __shared__ int ri[4];
  const int64_t b = blockIdx.x;
  const float* pr = p + b * V;
  for (int64_t c = threadIdx.x; c < V; c += 256) row[c] = pr[c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < (int)V; c += 256) {
      const float v = row[c];
      if (v > bv) { bv = v; bi = c; }  // ascending c: first hit of a value keeps the lower index
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { rv[wv] = bv; ri[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float fv = rv[0];
      int fi = ri[0];
      for (int q = 1; q < 4; ++q)
        if (rv[q] > fv || (rv[q] == fv && ri[q] < fi)) { fv = rv[q]; fi = ri[q]; }
      if (fi < 0 || fi >= (int)V) {
        // nothing above -inf is left (NaN / -inf scores, or k > the number of finite ones): hand out the lowest class index not
        // returned yet, with its original score, so that callers can always gather with the indices
        for (int c = 0; c < (int)V; ++c) {
          bool used = false;
          for (int q = 0; q < r; ++q) used = used || idx[b * k + q] == c;
          if (!used) { fi = c; break; }
        }
        fv = pr[fi];
      }
      vals[b * k + r] = fv;
      idx[b * k + r] = fi;
      row[fi] = -INFINITY;
    }
    __syncthreads();
  }
}

// ---- per-row precision at equal recall rate (W/eval_util.py:74-99) ------------------------------------------------
// One workgroup per video: the row and the list of its positive classes sit in LDS; a positive class j counts when its
// rank (#scores above it, ties towards the lower class index = a stable descending sort) is below the number of
// positives and its score is > 0.  Integer counting only: bit-exact against the host restatement when the boundary
// scores are distinct.  A video without labels scores 0 (the reference then averages label hits over ALL classes: 0).
__global__ __launch_bounds__(256) void perr_rows_kernel(const float* __restrict__ p, const uint8_t* __restrict__ y, int64_t V,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float row[];
  int* pos = reinterpret_cast<int*>(row + V);
  __shared__ int npos;
  __shared__ int cnt[4];
  const int64_t b = blockIdx.x;
  if (threadIdx.x == 0) npos = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < (int)V; c += 256) {
    row[c] = p[b * V + c];
    if (y[b * V + c]) pos[atomicAdd(&npos, 1)] = c;
  }
  __syncthreads();
  const int nl = npos;
  if (nl == 0) {
    if (threadIdx.x == 0) out[b] = 0.f;
    return;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int hits = 0;
  for (int q = 0; q < nl; ++q) {
    const int j = pos[q];
    const float pj = row[j];
    int g = 0;
    for (int c = threadIdx.x; c < (int)V; c += 256) {
      const float v = row[c];
      g += (v > pj || (v == pj && c < j)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
    if (lane == 0) cnt[wv] = g;
    __syncthreads();
    if (threadIdx.x == 0 && (cnt[0] + cnt[1] + cnt[2] + cnt[3]) < nl && pj > 0.f) ++hits;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[b] = (float)hits / (float)nl;
}

}  // namespace

using namespace yt8m;

namespace yt8m {  // lstm_fused.hip
bool lstm_fused_supported(int64_t B, int64_t H, int64_t workspace_bytes);
int lstm_pack(const float* Wh, int64_t ldw, float* Wp, float* Wq, int64_t H, hipStream_t s);
int lstm_step_fwd(float* z, const float* Wp, const float* c_prev, const float* h_prev, float* c_new, float* h_new, float* out,
                  const int32_t* nf, int t, int64_t B, int64_t H, float fb, hipStream_t s);
int lstm_step_bwd(const float* dz, const float* Wq, float* dh_prev, int64_t B, int64_t H, hipStream_t s);
int lstm_step_bwd_fused(const float* dz_t, const float* Wq, const float* dh_prev, const float* gates1, const float* c_prev1,
                        const float* c_new1, const float* dc_in, const float* dout1, float* dz1, float* dc_out, float* dh_out,
                        const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s);
}  // namespace yt8m

// Packed-weight backward over steps t_hi .. t_lo: ONE launch per step -- the pointwise gate backward runs once for t_hi,
// afterwards the product kernel of step t carries the gate backward of step t-1 as its epilogue; a plain product closes
// the range.  (dh, dc) ping-pong between the two halves of `work` exactly as in the two-launch form.
static int lstm_bwd_range_packed(const float* gates, const float* Wq, const float* cs, const float* dout, float* dz,
                                 float*& dh_cur, float*& dc_cur, float*& dh_prev, float*& dc_prev, const int32_t* num_frames,
                                 int64_t t_lo, int64_t t_hi, int64_t B, int64_t H, yt8m_stream_t stream) {
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H, Z = B * 4 * H;
  int rc = yt8m_lstm_gates_bwd(gates + t_hi * Z, cs + t_hi * BH, cs + (t_hi + 1) * BH, dh_cur, dc_cur,
                               dout ? dout + t_hi * BH : nullptr, dz + t_hi * Z, dc_prev, dh_prev, num_frames, (int32_t)t_hi, B, H,
                               stream);
  ProfScope prof(F_LSTM, s);
  for (int64_t t = t_hi; t >= t_lo && rc == YT8M_OK; --t) {
    if (t > t_lo) {
      const int64_t t1 = t - 1;
      rc = lstm_step_bwd_fused(dz + t * Z, Wq, dh_prev, gates + t1 * Z, cs + t1 * BH, cs + (t1 + 1) * BH, dc_prev,
                               dout ? dout + t1 * BH : nullptr, dz + t1 * Z, dc_cur, dh_cur, num_frames, (int)t, B, H, s);
    } else {
      rc = lstm_step_bwd(dz + t * Z, Wq, dh_prev, B, H, s);
    }
    float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
    tmp = dc_cur; dc_cur = dc_prev; dc_prev = tmp;
  }
  return rc;
}

extern "C" int yt8m_lstm_gates_fwd(float* z, const float* c_prev, const float* h_prev, float* c_new, float* h_new,
                                   float* out, const int32_t* num_frames, int32_t t, int64_t B, int64_t H,
                                   float forget_bias, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && c_prev && h_prev && c_new && h_new, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_LSTM, s);
  hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, z, c_prev, h_prev,
                     c_new, h_new, out, num_frames, (int)t, B, H, forget_bias);
  return launch_status("lstm_gates_fwd_kernel");
}

extern "C" int yt8m_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh,
                                   const float* dc, const float* dout, float* dz, float* dc_prev, float* dh_prev,
                                   const int32_t* num_frames, int32_t t, int64_t B, int64_t H, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(gates && c_prev && c_new && dh && dc && dz && dc_prev && dh_prev, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_LSTM, s);
  hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, s, gates, c_prev, c_new,
                     dh, dc, dout, dz, dc_prev, dh_prev, num_frames, (int)t, B, H);
  return launch_status("lstm_gates_bwd_kernel");
}

// ---- whole-layer drivers: the time loop lives here (host side of the library), not in Python -------------
extern "C" int yt8m_lstm_layer_fwd(float* z, const float* Wh, int64_t ldw, float* cs, float* hs, float* out,
                                   const int32_t* num_frames, int64_t F, int64_t B, int64_t H, float forget_bias,
                                   void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wh && cs && hs, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  const int64_t BH = B * H;
  if (gemm_workspace && lstm_fused_supported(B, H, gemm_workspace_bytes)) {
    // fused path: one launch per step (recurrent product + gates + copy-through), W_h re-packed once per call
    ProfScope prof(F_LSTM, as_stream(stream));
    float* Wp = static_cast<float*>(gemm_workspace);
    int rc = lstm_pack(Wh, ldw, Wp, nullptr, H, as_stream(stream));
    for (int64_t t = 0; t < F && rc == YT8M_OK; ++t)
      rc = lstm_step_fwd(z + t * B * 4 * H, Wp, cs + t * BH, hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH,
                         out ? out + t * BH : nullptr, num_frames, (int)t, B, H, forget_bias, as_stream(stream));
    return rc;
  }
  for (int64_t t = 0; t < F; ++t) {
    float* zt = z + t * B * 4 * H;
    // z_t += h_{t-1} . Wh        [B,H] x [H,4H]
    yt8m_gemm_problem pr = {B, 4 * H, H, hs + t * BH, H, Wh, ldw, zt, 4 * H, nullptr, 1.0f};
    int rc = yt8m_gemm_f32_grouped(0, 0, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    rc = yt8m_lstm_gates_fwd(zt, cs + t * BH, hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH,
                             out ? out + t * BH : nullptr, num_frames, (int32_t)t, B, H, forget_bias, stream);
    if (rc != YT8M_OK) return rc;
  }
  return YT8M_OK;
}

extern "C" int yt8m_lstm_layer_bwd(const float* gates, const float* Wh, int64_t ldw, const float* cs, const float* dout,
                                   const float* dc_final, const float* dh_final, float* dz, float* work,
                                   const int32_t* num_frames, int64_t F, int64_t B, int64_t H, void* gemm_workspace,
                                   int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(F >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (F * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(gates && Wh && cs && dz && work, YT8M_E_BADARG, "null operand");
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
  const bool fused = gemm_workspace && lstm_fused_supported(B, H, gemm_workspace_bytes);
  float* Wq = static_cast<float*>(gemm_workspace);
  if (fused) {
    int rc = lstm_pack(Wh, ldw, nullptr, Wq, H, s);
    if (rc != YT8M_OK) return rc;
    return lstm_bwd_range_packed(gates, Wq, cs, dout, dz, dh_cur, dc_cur, dh_prev, dc_prev, num_frames, 0, F - 1, B, H, stream);
  }
  for (int64_t t = F - 1; t >= 0; --t) {
    float* dzt = dz + t * B * 4 * H;
    int rc = yt8m_lstm_gates_bwd(gates + t * B * 4 * H, cs + t * BH, cs + (t + 1) * BH, dh_cur, dc_cur,
                                 dout ? dout + t * BH : nullptr, dzt, dc_prev, dh_prev, num_frames, (int32_t)t, B, H, stream);
    if (rc != YT8M_OK) return rc;
    // dh_{t-1} += dz_t . Wh^T    [B,4H] x [4H,H]   (Wh stored [H,4H] => transB)
    if (fused) {
      rc = lstm_step_bwd(dzt, Wq, dh_prev, B, H, s);
    } else {
      yt8m_gemm_problem pr = {B, H, 4 * H, dzt, 4 * H, Wh, ldw, dh_prev, H, nullptr, 1.0f};
      rc = yt8m_gemm_f32_grouped(0, 1, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
    }
    if (rc != YT8M_OK) return rc;
    float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
    tmp = dc_cur; dc_cur = dc_prev; dc_prev = tmp;
  }
  return YT8M_OK;
}

// ---- time-range forms: the same step kernels over steps [t0, t0 + T) of a layer, with the packed recurrent weights owned
// by the caller.  They let the host pipeline the layers of a stack over time chunks on separate streams (layer l+1's
// input projection and recurrence for chunk c run while layer l is in chunk c+1), seq_ops._LstmStack.
extern "C" int64_t yt8m_lstm_packed_floats(int64_t B, int64_t H) {
  return lstm_fused_supported(B, H, (int64_t)sizeof(float) * H * 4 * H) ? H * 4 * H : 0;
}

extern "C" int yt8m_lstm_pack(const float* Wh, int64_t ldw, int64_t H, float* Wp, float* Wq, yt8m_stream_t stream) {
  YT8M_REQUIRE(Wh && H > 0 && ldw >= 4 * H && (Wp || Wq), YT8M_E_BADARG, "bad operand");
  YT8M_REQUIRE(lstm_fused_supported(1, H, (int64_t)sizeof(float) * H * 4 * H), YT8M_E_SHAPE, "H must be a multiple of 128");
  ProfScope prof(F_LSTM, as_stream(stream));
  return lstm_pack(Wh, ldw, Wp, Wq, H, as_stream(stream));
}

extern "C" int yt8m_lstm_steps_fwd(float* z, const float* Wh, int64_t ldw, const float* Wp, float* cs, float* hs, float* out,
                                   const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                                   void* gemm_workspace, int64_t gemm_workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wh && cs && hs, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  const int64_t BH = B * H;
  if (Wp) {
    YT8M_REQUIRE(lstm_fused_supported(B, H, (int64_t)sizeof(float) * H * 4 * H), YT8M_E_SHAPE, "packed path needs H % 128 == 0");
    ProfScope prof(F_LSTM, as_stream(stream));
    hipStream_t s = as_stream(stream);
    // T tiny kernels back to back: launch-bound -> captured once per argument tuple, replayed as one hipGraph afterwards
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 1;
    key.p[0] = z; key.p[1] = Wp; key.p[2] = cs; key.p[3] = hs; key.p[4] = out; key.p[5] = num_frames;
    key.v[0] = t0; key.v[1] = T; key.v[2] = B; key.v[3] = H;
    memcpy(&key.v[4], &forget_bias, sizeof(float));
    return run_chain(key, s, [&]() {
      int rc = YT8M_OK;
      for (int64_t t = t0; t < t0 + T && rc == YT8M_OK; ++t)
        rc = lstm_step_fwd(z + t * B * 4 * H, Wp, cs + t * BH, hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH,
                           out ? out + t * BH : nullptr, num_frames, (int)t, B, H, forget_bias, s);
      return rc;
    });
  }
  for (int64_t t = t0; t < t0 + T; ++t) {
    float* zt = z + t * B * 4 * H;
    yt8m_gemm_problem pr = {B, 4 * H, H, hs + t * BH, H, Wh, ldw, zt, 4 * H, nullptr, 1.0f};
    int rc = yt8m_gemm_f32_grouped(0, 0, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
    if (rc != YT8M_OK) return rc;
    rc = yt8m_lstm_gates_fwd(zt, cs + t * BH, hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH,
                             out ? out + t * BH : nullptr, num_frames, (int32_t)t, B, H, forget_bias, stream);
    if (rc != YT8M_OK) return rc;
  }
  return YT8M_OK;
}

// steps t0 + T - 1 down to t0.  work [4,B,H]: the running (dh, dc) live in work[0], work[1] when `phase` is 0 and in
// work[2], work[3] when it is 1; every step flips it -- the caller carries phase' = (phase + T) % 2 to the next chunk.
extern "C" int yt8m_lstm_steps_bwd(const float* gates, const float* Wh, int64_t ldw, const float* Wq, const float* cs,
                                   const float* dout, float* dz, float* work, int phase, const int32_t* num_frames, int64_t t0,
                                   int64_t T, int64_t B, int64_t H, void* gemm_workspace, int64_t gemm_workspace_bytes,
                                   yt8m_stream_t stream) {
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(phase == 0 || phase == 1, YT8M_E_BADARG, "phase must be 0 or 1");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(gates && Wh && cs && dz && work, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldw >= 4 * H, YT8M_E_SHAPE, "ldw < 4H");
  if (Wq) YT8M_REQUIRE(lstm_fused_supported(B, H, (int64_t)sizeof(float) * H * 4 * H), YT8M_E_SHAPE, "packed path needs H % 128 == 0");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H;
  auto chain = [&]() {
    float* dh_cur = work + (phase ? 2 : 0) * BH;
    float* dc_cur = dh_cur + BH;
    float* dh_prev = work + (phase ? 0 : 2) * BH;
    float* dc_prev = dh_prev + BH;
    if (Wq) return lstm_bwd_range_packed(gates, Wq, cs, dout, dz, dh_cur, dc_cur, dh_prev, dc_prev, num_frames, t0, t0 + T - 1, B, H, stream);
    for (int64_t t = t0 + T - 1; t >= t0; --t) {
      float* dzt = dz + t * B * 4 * H;
      int rc = yt8m_lstm_gates_bwd(gates + t * B * 4 * H, cs + t * BH, cs + (t + 1) * BH, dh_cur, dc_cur,
                                   dout ? dout + t * BH : nullptr, dzt, dc_prev, dh_prev, num_frames, (int32_t)t, B, H, stream);
      if (rc != YT8M_OK) return rc;
      if (Wq) {
        rc = lstm_step_bwd(dzt, Wq, dh_prev, B, H, s);
      } else {
        yt8m_gemm_problem pr = {B, H, 4 * H, dzt, 4 * H, Wh, ldw, dh_prev, H, nullptr, 1.0f};
        rc = yt8m_gemm_f32_grouped(0, 1, 1, &pr, gemm_workspace, gemm_workspace_bytes, stream);
      }
      if (rc != YT8M_OK) return rc;
      float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
      tmp = dc_cur; dc_cur = dc_prev; dc_prev = tmp;
    }
    return (int)YT8M_OK;
  };
  if (!Wq) return chain();                        // generic path: host-side split-K decisions, not captured
  GraphKey key;
  memset(&key, 0, sizeof(key));
  key.kind = 2;
  key.p[0] = gates; key.p[1] = Wq; key.p[2] = cs; key.p[3] = dout; key.p[4] = dz; key.p[5] = work; key.p[6] = num_frames;
  key.v[0] = t0; key.v[1] = T; key.v[2] = B; key.v[3] = H; key.v[4] = phase;
  return run_chain(key, s, chain);
}

extern "C" int yt8m_attn_softmax_fwd(const float* act, const int32_t* num_frames, float* w, int64_t B, int64_t F,
                                     int64_t A, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * A == 0) return YT8M_OK;
  YT8M_REQUIRE(act && w, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3((unsigned)((B * A + 3) / 4)), dim3(256), 0, s, act, num_frames, w, B, F, A);
  return launch_status("attn_softmax_fwd_kernel");
}

extern "C" int yt8m_attn_softmax_bwd(const float* w, const float* dw, const int32_t* num_frames, float* dact, int64_t B,
                                     int64_t F, int64_t A, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * A == 0) return YT8M_OK;
  YT8M_REQUIRE(w && dw && dact, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3((unsigned)((B * A + 3) / 4)), dim3(256), 0, s, w, dw, num_frames, dact, B, F, A);
  return launch_status("attn_softmax_bwd_kernel");
}

extern "C" int yt8m_softmax_rows_fwd(const float* sin, const int32_t* num_frames, float* a, int64_t B, int64_t F,
                                     int64_t K, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && K >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * K == 0) return YT8M_OK;
  YT8M_REQUIRE(sin && a, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, sin, num_frames, a, B * F, F, K);
  return launch_status("softmax_rows_fwd_kernel");
}

extern "C" int yt8m_softmax_rows_bwd(const float* a, const float* da, const int32_t* num_frames, float* ds, int64_t B,
                                     int64_t F, int64_t K, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && K >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * K == 0) return YT8M_OK;
  YT8M_REQUIRE(a && da && ds, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, a, da, num_frames, ds, B * F, F, K);
  return launch_status("softmax_rows_bwd_kernel");
}

extern "C" int yt8m_topk_rows(const float* p, int64_t B, int64_t V, int k, float* vals, int32_t* idx,
                              yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && V >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(k >= 1 && k <= 64 && k <= V, YT8M_E_BADARG, "k must be in [1, min(64, V)]");
  YT8M_REQUIRE(V * (int64_t)sizeof(float) <= 150 * 1024, YT8M_E_SHAPE, "row does not fit LDS (V > 38400)");
  if (B == 0) return YT8M_OK;
  YT8M_REQUIRE(p && vals && idx, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const size_t shm = (size_t)V * sizeof(float);
  if (shm > 64 * 1024) {
    YT8M_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  }
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)B), dim3(256), shm, s, p, V, k, vals, idx);
  return launch_status("topk_rows_kernel");
}

extern "C" int yt8m_perr_rows(const float* p, const uint8_t* labels, int64_t B, int64_t V, float* perr, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && V >= 1, YT8M_E_SHAPE, "bad dimension");
  YT8M_REQUIRE(V * 8 <= 150 * 1024, YT8M_E_SHAPE, "row + label list do not fit LDS (V > 19200)");
  if (B == 0) return YT8M_OK;
  YT8M_REQUIRE(p && labels && perr, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const size_t shm = (size_t)V * 8;
  if (shm > 64 * 1024) {
    YT8M_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(perr_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  }
  hipLaunchKernelGGL(perr_rows_kernel, dim3((unsigned)B), dim3(256), shm, s, p, labels, V, perr);
  return launch_status("perr_rows_kernel");
}
