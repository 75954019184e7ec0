// lstm_bf16.hip -- BasicLSTM recurrence with bf16 OPERANDS for the recurrent product (--compute_dtype=bfloat16), gfx950.
// Same decomposition as lstm_fused.hip (a workgroup = 32 batch rows x 8 hidden units forward, 16 rows x 16 units backward, four
// waves split K, gate block / gate backward as the epilogue), but h_{t-1} / dz_t and the packed W_h enter the matrix cores as
// bf16 (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16, fp32 accumulate): the fp32 step kernels are bound by the 128 MB of
// h / W_h they pull through the L2s per step, bf16 operands halve that traffic and cut the matrix time 8x.  The cell state,
// the gates, the hidden state handed to the next layer and every gradient stay fp32; each step additionally emits the bf16 copy
// of h_t (forward) / dz_{t-1} (backward) that the next step's product reads.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// Wp16[ug][k/8][c][k%8], c = gate*8 + u  <-  Wh[k][gate*H + ug*8 + u]       (H/8 groups, k < H)
// Wq16[ug][k/8][u][k%8]                  <-  Wh[ug*16 + u][k]               (H/16 groups, k < 4H)
__global__ __launch_bounds__(256) void lstm_pack16_kernel(const float* __restrict__ Wh, int64_t ldw, unsigned short* __restrict__ Wp,
                                                          unsigned short* __restrict__ Wq, int H) {
  const int64_t n = (int64_t)H * 4 * H;
  const int K = 4 * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int k8 = (int)(e & 7);
    if (Wp) {
      const int c = (int)((e >> 3) & 31);
      const int64_t r = e >> 8;           // ug*(H/8) + kk
      const int k = (int)(r % (H >> 3)) * 8 + k8, ug = (int)(r / (H >> 3));
      Wp[e] = f2bf(Wh[(int64_t)k * ldw + (c >> 3) * H + ug * 8 + (c & 7)]);
    }
    if (Wq) {
      const int u = (int)((e >> 3) & 15);
      const int64_t r = e >> 7;           // ug*(K/8) + kk
      const int k = (int)(r % (K >> 3)) * 8 + k8, ug = (int)(r / (K >> 3));
      Wq[e] = f2bf(Wh[(int64_t)(ug * 16 + u) * ldw + k]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_rows16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n) {
  const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n) return;                     // n % 4 == 0
  const float4 v = *reinterpret_cast<const float4*>(src + e);
  *reinterpret_cast<uint2*>(dst + e) = uint2{(unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16), (unsigned)f2bf(v.z) | ((unsigned)f2bf(v.w) << 16)};
}

__global__ __launch_bounds__(256) void lstm_step_fwd16_kernel(float* __restrict__ z, const unsigned short* __restrict__ Wp,
                                                              const unsigned short* __restrict__ h16_prev,
                                                              const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                                              float* __restrict__ c_new, float* __restrict__ h_new,
                                                              unsigned short* __restrict__ h16_new, float* __restrict__ out,
                                                              const int32_t* __restrict__ nf, int t, int B, int H, float fb) {
  __shared__ float red[4][32][33];
  const int groups = H >> 3;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int kq = H >> 2;                                   // K-range of one wave, a multiple of 64
  int row = m0 + i;
  if (row >= B) row = B - 1;
  const uint4* ap = reinterpret_cast<const uint4*>(h16_prev + (int64_t)row * H + w * kq + 8 * kh);        // + 2 per MFMA (16 k)
  const uint4* bp = reinterpret_cast<const uint4*>(Wp) + ((int64_t)ug * (H >> 3) + ((w * kq) >> 3) + kh) * 32 + i;   // + 64 per MFMA
  // epilogue operands first: their latency hides under the K loop
  float zpre[4] = {0.f, 0.f, 0.f, 0.f}, cpre = 0.f;
  int nfpre = 0x7fffffff;
  {
    const int eb = m0 + (tid >> 3);
    if (eb < B) {
      if (nf) nfpre = nf[eb];
      const int eu = ug * 8 + (tid & 7);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) zpre[g4] = z[(int64_t)eb * 4 * H + g4 * H + eu];
      cpre = c_prev[(int64_t)eb * H + eu];
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int g0 = 0; g0 < (kq >> 4); g0 += 4) {              // 4 MFMAs (64 k) per block
    uint4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = ap[(g0 + q) * 2];
      b[q] = bp[(g0 + q) * 64];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q]), __builtin_bit_cast(bf16x8, b[q]), acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * kh][i] = acc[r];
  __syncthreads();
  const int r = tid >> 3, u = tid & 7;
  const int b = m0 + r;
  if (b >= B) return;
  const int unit = ug * 8 + u;
  const int64_t idx = (int64_t)b * H + unit;
  if (!(t < nfpre)) {
    const float hp = h_prev[idx];
    c_new[idx] = cpre;
    h_new[idx] = hp;
    h16_new[idx] = f2bf(hp);
    if (out) out[idx] = 0.f;
    return;
  }
  float* zr = z + (int64_t)b * 4 * H + unit;
  float pre[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    pre[g4] = zpre[g4] + ((red[0][r][g4 * 8 + u] + red[1][r][g4 * 8 + u]) + (red[2][r][g4 * 8 + u] + red[3][r][g4 * 8 + u]));
  const float gi = sigmoidf_(pre[0]);
  const float gj = tanhf(pre[1]);
  const float gf = sigmoidf_(pre[2] + fb);
  const float go = sigmoidf_(pre[3]);
  const float c = cpre * gf + gi * gj;
  const float hn = tanhf(c) * go;
  zr[0] = gi; zr[H] = gj; zr[2 * H] = gf; zr[3 * H] = go;
  c_new[idx] = c;
  h_new[idx] = hn;
  h16_new[idx] = f2bf(hn);
  if (out) out[idx] = hn;
}

struct GateBwd16 {
  const float* gates1; const float* c_prev1; const float* c_new1; const float* dc_in; const float* dout1;
  float* dz1; unsigned short* dz16_1; float* dc_out; float* dh_out;
};

// dh_prev[B,H] += dz16[B,4H] . Wh^T;  FUSE: gate backward of step t-1 from the finished dL/dh_{t-1} (as lstm_fused.hip BEP 2),
// writing dz_{t-1} in fp32 (for the hoisted products) and bf16 (for the next step's product)
template <bool FUSE>
__global__ __launch_bounds__(256) void lstm_step_bwd16_kernel(const unsigned short* __restrict__ dz16, const unsigned short* __restrict__ Wq,
                                                              float* __restrict__ dh_prev, int B, int H,
                                                              const int32_t* __restrict__ nf, int t, GateBwd16 gb) {
  __shared__ float red[4][16][17];
  const int groups = H >> 4;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int K4 = 4 * H, KW = H;                            // K-range of one wave (a multiple of 128)
  int row = m0 + i;
  if (row >= B) row = B - 1;
  const uint4* ap = reinterpret_cast<const uint4*>(dz16 + (int64_t)row * K4 + w * KW + 8 * kq);           // + 4 per MFMA (32 k)
  const uint4* bp = reinterpret_cast<const uint4*>(Wq) + ((int64_t)ug * (K4 >> 3) + ((w * KW) >> 3) + kq) * 16 + i;  // + 64 per MFMA
  float dpre = 0.f, gpre[4] = {0.f, 0.f, 0.f, 0.f}, cp1 = 0.f, cn1 = 0.f, dc1 = 0.f, do1 = 0.f;
  int nfpre = 0x7fffffff;
  {
    const int eb = m0 + (tid >> 4), eu = ug * 16 + (tid & 15);
    if (eb < B) {
      dpre = dh_prev[(int64_t)eb * H + eu];
      if (FUSE) {
        if (nf) nfpre = nf[eb];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) gpre[g4] = gb.gates1[(int64_t)eb * 4 * H + g4 * H + eu];
        cp1 = gb.c_prev1[(int64_t)eb * H + eu];
        cn1 = gb.c_new1[(int64_t)eb * H + eu];
        dc1 = gb.dc_in[(int64_t)eb * H + eu];
        do1 = gb.dout1 ? gb.dout1[(int64_t)eb * H + eu] : 0.f;
      }
    }
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int g0 = 0; g0 < (KW >> 5); g0 += 4) {              // 4 MFMAs (128 k) per block
    uint4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = ap[(g0 + q) * 4];
      b[q] = bp[(g0 + q) * 64];
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[1]), acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2]), __builtin_bit_cast(bf16x8, b[2]), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[3]), __builtin_bit_cast(bf16x8, b[3]), acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][4 * kq + r][i] = acc0[r] + acc1[r];
  __syncthreads();
  const int r = tid >> 4, u = tid & 15;
  const int b = m0 + r;
  if (b >= B) return;
  const int64_t idx = (int64_t)b * H + ug * 16 + u;
  const float sum = (red[0][r][u] + red[1][r][u]) + (red[2][r][u] + red[3][r][u]);
  if (!FUSE) {
    dh_prev[idx] = dpre + sum;
    return;
  }
  float* dzr = gb.dz1 + (int64_t)b * 4 * H + ug * 16 + u;
  unsigned short* dzh = gb.dz16_1 + (int64_t)b * 4 * H + ug * 16 + u;
  const float dh_in = dpre + sum;
  if (!((t - 1) < nfpre)) {
    dzr[0] = 0.f; dzr[H] = 0.f; dzr[2 * H] = 0.f; dzr[3 * H] = 0.f;
    dzh[0] = 0; dzh[H] = 0; dzh[2 * H] = 0; dzh[3 * H] = 0;
    gb.dc_out[idx] = dc1;
    gb.dh_out[idx] = dh_in;
  } else {
    const float gi = gpre[0], gj = gpre[1], gf = gpre[2], go = gpre[3];
    const float tc = tanhf(cn1);
    const float dht = dh_in + do1;
    const float dct = dc1 + dht * go * (1.0f - tc * tc);
    const float d0 = dct * gj * gi * (1.0f - gi), d1 = dct * gi * (1.0f - gj * gj), d2 = dct * cp1 * gf * (1.0f - gf),
                d3 = dht * tc * go * (1.0f - go);
    dzr[0] = d0; dzr[H] = d1; dzr[2 * H] = d2; dzr[3 * H] = d3;
    dzh[0] = f2bf(d0); dzh[H] = f2bf(d1); dzh[2 * H] = f2bf(d2); dzh[3 * H] = f2bf(d3);
    gb.dc_out[idx] = dct * gf;
    gb.dh_out[idx] = 0.f;
  }
}

}  // namespace

using namespace yt8m;

extern "C" int64_t yt8m_lstm_packed16_elems(int64_t B, int64_t H) { return (B >= 1 && H >= 256 && H % 256 == 0) ? H * 4 * H : 0; }

extern "C" int yt8m_lstm_pack_bf16(const float* Wh, int64_t ldw, int64_t H, void* Wp16, void* Wq16, yt8m_stream_t stream) {
  YT8M_REQUIRE(Wh && H > 0 && ldw >= 4 * H && (Wp16 || Wq16), YT8M_E_BADARG, "bad operand");
  YT8M_REQUIRE(yt8m_lstm_packed16_elems(1, H) > 0, YT8M_E_SHAPE, "bf16 recurrence needs H % 256 == 0");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_LSTM, s);
  hipLaunchKernelGGL(lstm_pack16_kernel, dim3(2048), dim3(256), 0, s, Wh, ldw, static_cast<unsigned short*>(Wp16),
                     static_cast<unsigned short*>(Wq16), (int)H);
  return launch_status("lstm_pack16_kernel");
}

extern "C" int yt8m_lstm_steps_fwd_bf16(float* z, const void* Wp16, float* cs, float* hs, void* hs16, float* out,
                                        const int32_t* num_frames, int64_t t0, int64_t T, int64_t B, int64_t H, float forget_bias,
                                        yt8m_stream_t stream) {
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(z && Wp16 && cs && hs && hs16, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(yt8m_lstm_packed16_elems(B, H) > 0, YT8M_E_SHAPE, "bf16 recurrence needs H % 256 == 0");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_LSTM, s);
  const int64_t BH = B * H;
  unsigned short* h16 = static_cast<unsigned short*>(hs16);
  GraphKey key;
  memset(&key, 0, sizeof(key));
  key.kind = 3;
  key.p[0] = z; key.p[1] = Wp16; key.p[2] = cs; key.p[3] = hs; key.p[4] = out; key.p[5] = num_frames; key.p[6] = hs16;
  key.v[0] = t0; key.v[1] = T; key.v[2] = B; key.v[3] = H;
  memcpy(&key.v[4], &forget_bias, sizeof(float));
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 8));
  return run_chain(key, s, [&]() {
    for (int64_t t = t0; t < t0 + T; ++t)
      hipLaunchKernelGGL(lstm_step_fwd16_kernel, dim3(grid), dim3(256), 0, s, z + t * B * 4 * H, static_cast<const unsigned short*>(Wp16),
                         h16 + t * BH, cs + t * BH, hs + t * BH, cs + (t + 1) * BH, hs + (t + 1) * BH, h16 + (t + 1) * BH,
                         out ? out + t * BH : nullptr, num_frames, (int)t, (int)B, (int)H, forget_bias);
    return launch_status("lstm_step_fwd16_kernel");
  });
}

// steps t0 + T - 1 down to t0; work / phase as yt8m_lstm_steps_bwd.  dz16 [F,B,4H] bf16 scratch (written here).
extern "C" int yt8m_lstm_steps_bwd_bf16(const float* gates, const void* Wq16, const float* cs, const float* dout, float* dz,
                                        void* dz16, float* work, int phase, const int32_t* num_frames, int64_t t0, int64_t T,
                                        int64_t B, int64_t H, yt8m_stream_t stream) {
  YT8M_REQUIRE(t0 >= 0 && T >= 0 && B >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(phase == 0 || phase == 1, YT8M_E_BADARG, "phase must be 0 or 1");
  if (T * B * H == 0) return YT8M_OK;
  YT8M_REQUIRE(gates && Wq16 && cs && dz && dz16 && work, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(yt8m_lstm_packed16_elems(B, H) > 0, YT8M_E_SHAPE, "bf16 recurrence needs H % 256 == 0");
  hipStream_t s = as_stream(stream);
  const int64_t BH = B * H, Z = B * 4 * H;
  unsigned short* d16 = static_cast<unsigned short*>(dz16);
  const unsigned short* Wq = static_cast<const unsigned short*>(Wq16);
  GraphKey key;
  memset(&key, 0, sizeof(key));
  key.kind = 4;
  key.p[0] = gates; key.p[1] = Wq16; key.p[2] = cs; key.p[3] = dout; key.p[4] = dz; key.p[5] = work; key.p[6] = num_frames; key.p[7] = dz16;
  key.v[0] = t0; key.v[1] = T; key.v[2] = B; key.v[3] = H; key.v[4] = phase;
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  return run_chain(key, s, [&]() {
    float* dh_cur = work + (phase ? 2 : 0) * BH;
    float* dc_cur = dh_cur + BH;
    float* dh_prev = work + (phase ? 0 : 2) * BH;
    float* dc_prev = dh_prev + BH;
    const int64_t t_hi = t0 + T - 1;
    int rc = yt8m_lstm_gates_bwd(gates + t_hi * Z, cs + t_hi * BH, cs + (t_hi + 1) * BH, dh_cur, dc_cur, dout ? dout + t_hi * BH : nullptr,
                                 dz + t_hi * Z, dc_prev, dh_prev, num_frames, (int32_t)t_hi, B, H, stream);
    if (rc != YT8M_OK) return rc;
    ProfScope prof(F_LSTM, s);
    hipLaunchKernelGGL(cast_rows16_kernel, dim3((unsigned)((Z / 4 + 255) / 256)), dim3(256), 0, s, dz + t_hi * Z, d16 + t_hi * Z, Z);
    for (int64_t t = t_hi; t >= t0; --t) {
      if (t > t0) {
        const int64_t t1 = t - 1;
        GateBwd16 gb = {gates + t1 * Z, cs + t1 * BH, cs + (t1 + 1) * BH, dc_prev, dout ? dout + t1 * BH : nullptr,
                        dz + t1 * Z, d16 + t1 * Z, dc_cur, dh_cur};
        hipLaunchKernelGGL((lstm_step_bwd16_kernel<true>), dim3(grid), dim3(256), 0, s, d16 + t * Z, Wq, dh_prev, (int)B, (int)H,
                           num_frames, (int)t, gb);
      } else {
        hipLaunchKernelGGL((lstm_step_bwd16_kernel<false>), dim3(grid), dim3(256), 0, s, d16 + t * Z, Wq, dh_prev, (int)B, (int)H,
                           num_frames, (int)t, GateBwd16{});
      }
      float* tmp = dh_cur; dh_cur = dh_prev; dh_prev = tmp;
      tmp = dc_cur; dc_cur = dc_prev; dc_prev = tmp;
    }
    return launch_status("lstm_step_bwd16_kernel");
  });
}
