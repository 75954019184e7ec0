// lstm_fused.hip -- one-launch-per-step BasicLSTM recurrence for gfx950 (tf.contrib.rnn.BasicLSTMCell under
// tf.nn.dynamic_rnn; W/all_frame_models/lstm_model.py:34-47).
//
// The recurrent product of one step, z_t += h_{t-1} . W_h  ([B,H] x [H,4H], 1.07 GFLOP at B=128, H=1024), is far too small
// for 128x128 tiles (32 tiles on 256 CUs).  Here a workgroup owns 32 batch rows x 8 hidden units = the 32 gate columns
// (i|j|f|o) x 8 of those units, so the gate non-linearity, the cell update and the dynamic_rnn copy-through are the
// epilogue of the same launch: 512 workgroups at B = 128, every CU busy, ONE kernel per step.
//   * W_h is re-packed once per layer call into Wp[unit-group][k][32] (the 32 gate columns of a group contiguous: every
//     fragment fetch is two fully coalesced 128-byte rows).  Workgroup id % 8 == unit-group % 8, and the dispatcher places
//     id % 8 on XCD id % 8, so a unit group is always served by the same XCD: its 2 MiB share of W_h stays resident in
//     that XCD's 4 MiB L2 across all 300 steps (no LDS/HBM re-streaming of the 16 MiB weight matrix per step).
//   * the four waves of a workgroup split K = H; operands go straight to VGPRs in MFMA fragment order (k-permuted float4
//     loads for h, as in gemm_f32.hip), partial 32x32 tiles meet in LDS, and thread t finishes (row t/8, unit t%8).
// Backward: dz_t comes from the pointwise kernel (sequence.hip); dh_{t-1} += dz_t . W_h^T runs on 16x16x4 MFMAs with
// 16 rows x 16 units per workgroup (512 workgroups) over Wq[unit-group][k][16], K = 4H split over the four waves.
#include "common.h"

#ifndef STEP_KDIV
#define STEP_KDIV 1     // timing experiments only (tools/build_variant.sh -DSTEP_KDIV=n): run 1/n of every K loop
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// G gates of H units each (LSTM 4: i|j|f|o, GRU gate block 2: r|u, GRU candidate 1); U = 32 / G units per group.
// Both packed forms keep FOUR consecutive k of one column adjacent, so a lane fetches its B fragments for four successive
// MFMAs with one 16-byte load and a wave-wide load instruction reads 1 KB contiguous (8 full cache lines):
// Wp[ug][k/4][c][k%4], c = gate*U + u  <-  Wh[k][gate*H + ug*U + u]      (H/U groups, k < H)
// Wq[ug][k/4][u][k%4]                  <-  Wh[ug*16 + u][k]               (H/16 groups, k < G*H)
__global__ __launch_bounds__(256) void lstm_pack_kernel(const float* __restrict__ Wh, int64_t ldw, float* __restrict__ Wp,
                                                        float* __restrict__ Wq, int H, int G) {
  const int64_t n = (int64_t)H * G * H;
  const int U = 32 / G, K = G * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int k4 = (int)(e & 3);
    if (Wp) {
      const int c = (int)((e >> 2) & 31);
      const int64_t r = e >> 7;           // ug*(H/4) + kb
      const int k = (int)(r % (H >> 2)) * 4 + k4, ug = (int)(r / (H >> 2));
      Wp[e] = Wh[(int64_t)k * ldw + (c / U) * H + ug * U + (c % U)];
    }
    if (Wq) {
      const int u = (int)((e >> 2) & 15);
      const int64_t r = e >> 6;           // ug*(K/4) + kb
      const int k = (int)(r % (K >> 2)) * 4 + k4, ug = (int)(r / (K >> 2));
      Wq[e] = Wh[(int64_t)(ug * 16 + u) * ldw + k];
    }
  }
}

// epilogues of the forward step kernel
enum { EP_LSTM = 0,        // G = 4: BasicLSTMCell gates + cell update + copy-through
       EP_ADD = 1,         // any G: z += a . W only (LayerNormBasicLSTMCell: the normalisations need whole rows -> cells.hip)
       EP_GRU_GATES = 2,   // G = 2: r|u = sigmoid(zg + h . Wg_h), stored over zg; rh = r * h
       EP_GRU_CAND = 3 };  // G = 1: c = tanh(zc + rh . Wc_h) stored over zc; h' = u*h + (1-u)*c with copy-through

// a_in [B,H]: the A operand of the recurrent product (h_{t-1}; r*h for the GRU candidate).  gates / rh: GRU only.
// MODE bit 0: software prefetch of the next 32-k block; bit 1: the A tile (32 rows x 32 k per wave and block) is fetched
// with line-coalesced loads (a wave instruction = 8 rows x 128 B) and re-read in MFMA fragment order from a wave-private
// padded LDS tile -- fetching fragments directly costs four partial touches of every 128-byte line.
template <int G, int EP, int MODE>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float* __restrict__ z, const float* __restrict__ Wp,
                                                            const float* __restrict__ a_in, const float* __restrict__ c_prev,
                                                            const float* __restrict__ h_prev, float* __restrict__ c_new,
                                                            float* __restrict__ h_new, float* __restrict__ out,
                                                            const int32_t* __restrict__ nf, int t, int B, int H, float fb,
                                                            const float* __restrict__ gates, float* __restrict__ rh) {
  __shared__ float red[4][32][33];
  constexpr int U = 32 / G;
  const int groups = H / U;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int kq = H >> 2;                                   // K-range of one wave
  constexpr bool PREF = (MODE & 1) != 0, STAGE = (MODE & 2) != 0;
  __shared__ float stage[STAGE ? 4 : 1][STAGE ? 32 * 36 : 1];   // per wave: 32 rows x (32 k + 4 pad) floats
  // A addressing.  direct: lane (i, kh) reads its fragment row i, k = 8q + 4kh.. ; staged: lane l reads row l/8 + 8j, the
  // 16-byte chunk l%8 of the block's 128-byte row segment.
  const int srow = lane >> 3, sch = lane & 7;
  const float* apd;
  const float* aps[4];
  {
    int row = m0 + i;
    if (row >= B) row = B - 1;                             // clamped rows feed output rows >= B only (never stored)
    apd = a_in + (int64_t)row * H + w * kq + 4 * kh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r = m0 + srow + 8 * j;
      if (r >= B) r = B - 1;
      aps[j] = a_in + (int64_t)r * H + w * kq + 4 * sch;
    }
  }
  const float4* bp4 = reinterpret_cast<const float4*>(Wp) + ((int64_t)ug * (H >> 2) + ((w * kq) >> 2) + kh) * 32 + i;
  float* lw = &stage[STAGE ? w : 0][0];
  // the epilogue's operands do not depend on the product: fetch them now, their latency hides under the K loop
  float zpre[4] = {0.f, 0.f, 0.f, 0.f}, cpre = 0.f;
  int nfpre = 0x7fffffff;
  {
    const int eb = m0 + (tid >> 3);
    if (eb < B) {
      if (nf) nfpre = nf[eb];
      if (EP == EP_LSTM) {
        const int eu = ug * 8 + (tid & 7);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) zpre[g4] = z[(int64_t)eb * 4 * H + g4 * H + eu];
        cpre = c_prev[(int64_t)eb * H + eu];
      } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int c = (tid & 7) + 8 * jj;
          zpre[jj] = z[(int64_t)eb * G * H + (c / U) * H + ug * U + (c % U)];
        }
      }
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 v[4], bq[4];
  if (PREF) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] : apd + 8 * q);
      bq[q] = bp4[(2 * q) * 32];
    }
  }
  for (int g0 = 0; g0 < kq / STEP_KDIV; g0 += 32) {       // kq % 32 == 0 (H % 128 == 0); one block = 32 k = 16 MFMAs
    if (!PREF) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] + g0 : apd + g0 + 8 * q);
        bq[q] = bp4[((g0 >> 2) + 2 * q) * 32];
      }
    }
    float4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b[q] = bq[q];
      if (STAGE) *reinterpret_cast<float4*>(lw + (srow + 8 * q) * 36 + 4 * sch) = v[q];
      else a[q] = v[q];
    }
    if (PREF) {                                // no branch around the loads: the last block re-fetches itself
      const int gn = (g0 + 32 < kq) ? g0 + 32 : g0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] + gn : apd + gn + 8 * q);
        bq[q] = bp4[((gn >> 2) + 2 * q) * 32];
      }
    }
    if (STAGE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const float4*>(lw + i * 36 + 8 * q + 4 * kh);
    }
    if (PREF) __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE the MFMA block (the scheduler sinks loads)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, b[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, b[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, b[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, b[q].w, acc, 0, 0, 0);
    }
    if (PREF) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * kh][i] = acc[r];
  __syncthreads();
  // thread -> (batch row, hidden unit): gate pre-activations = sum of the 4 K-parts + hoisted input projection
  const int r = tid >> 3, u = tid & 7;
  const int b = m0 + r;
  if (b >= B) return;
  if (EP != EP_LSTM) {
    const bool live = t < nfpre;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int c = u + 8 * jj;                              // column of the 32-wide group
      const float sum = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);
      if (EP == EP_ADD) {
        z[(int64_t)b * G * H + (c / U) * H + ug * U + (c % U)] = zpre[jj] + sum;
      } else if (EP == EP_GRU_GATES) {                       // columns 0..15 = r of 16 units, 16..31 = u of the same units
        const int unit = ug * 16 + (c & 15);
        float* zp = z + (int64_t)b * 2 * H + (c >> 4) * H + unit;
        const float gv = sigmoidf_(zpre[jj] + sum);
        *zp = gv;
        if (c < 16) rh[(int64_t)b * H + unit] = gv * h_prev[(int64_t)b * H + unit];
      } else {                                               // EP_GRU_CAND
        const int64_t idx = (int64_t)b * H + ug * 32 + c;
        const float cand = tanhf(zpre[jj] + sum);
        z[idx] = cand;
        const float uu = gates[(int64_t)b * 2 * H + H + ug * 32 + c], hp = h_prev[idx];
        const float hn = live ? uu * hp + (1.0f - uu) * cand : hp;
        h_new[idx] = hn;
        if (out) out[idx] = live ? hn : 0.f;
      }
    }
    return;
  }
  const int unit = ug * 8 + u;
  const int64_t idx = (int64_t)b * H + unit;
  const bool live = t < nfpre;
  if (!live) {
    c_new[idx] = cpre;
    h_new[idx] = h_prev[idx];
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
  if (out) out[idx] = hn;
}

// operands of the BasicLSTM gate backward of step t-1, run as the epilogue of step t's product (BEP 2)
struct GateBwd {
  const float* gates1;   // [B,4H] saved gates i|j|f|o of step t-1
  const float* c_prev1;  // [B,H] c_{t-2}
  const float* c_new1;   // [B,H] c_{t-1}
  const float* dc_in;    // [B,H] dL/dc_{t-1} carried from step t
  const float* dout1;    // [B,H] dL/d(output_{t-1}) or NULL
  float* dz1;            // [B,4H] out: dL/dz_{t-1}
  float* dc_out;         // [B,H]  out: dL/dc_{t-2}
  float* dh_out;         // [B,H]  out: the part of dL/dh_{t-2} that does not come from the product (0, or the copy-through)
};

// dh_prev[B,H] += dz[B,K] . Wh^T, K = G*H (16 rows x 16 units per workgroup, v_mfma_f32_16x16x4_f32, K over 4 waves)
// BEP 0: accumulate (LSTM, LN-LSTM, GRU gate block).  BEP 1 (GRU candidate, K = H): the product is d(r*h):
//        dzg_r = d * h * r * (1 - r) for live rows (0 otherwise), dh_prev += d * r.
// BEP 2 (LSTM): dL/dh_{t-1} = dh_prev + product stays in a register and feeds the gate backward of step t-1 at once (one
//        (row, unit) per thread = exactly the pointwise kernel's work item): one launch per backward step instead of two.
template <int BEP, int MODE>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ Wq,
                                                            float* __restrict__ dh_prev, int B, int H, int K4,
                                                            const float* __restrict__ gates, const float* __restrict__ h_prev,
                                                            float* __restrict__ dzg, const int32_t* __restrict__ nf, int t,
                                                            GateBwd gb) {
  __shared__ float red[4][16][17];
  const int groups = H >> 4;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int KW = K4 >> 2;                                   // K-range of one wave
  constexpr bool PREF = (MODE & 1) != 0, STAGE = (MODE & 2) != 0;
  __shared__ float stage[STAGE ? 4 : 1][STAGE ? 16 * 68 : 1];   // per wave: 16 rows x (64 k + 4 pad) floats
  const int srow = lane >> 4, sch = lane & 15;                // staged: lane l reads row l/16 + 4j, chunk l%16 of 256 bytes
  const float* apd;
  const float* aps[4];
  {
    int row = m0 + i;
    if (row >= B) row = B - 1;
    apd = dz + (int64_t)row * K4 + w * KW + 4 * kq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r = m0 + srow + 4 * j;
      if (r >= B) r = B - 1;
      aps[j] = dz + (int64_t)r * K4 + w * KW + 4 * sch;
    }
  }
  const float4* bp4 = reinterpret_cast<const float4*>(Wq) + ((int64_t)ug * (K4 >> 2) + ((w * KW) >> 2) + kq) * 16 + i;
  float* lw = &stage[STAGE ? w : 0][0];
  float dpre = 0.f, rpre = 0.f, hpre = 0.f;                 // epilogue operands, fetched ahead of the K loop
  float gpre[4] = {0.f, 0.f, 0.f, 0.f}, cp1 = 0.f, cn1 = 0.f, dc1 = 0.f, do1 = 0.f;
  int nfpre = 0x7fffffff;
  {
    const int eb = m0 + (tid >> 4), eu = ug * 16 + (tid & 15);
    if (eb < B) {
      dpre = dh_prev[(int64_t)eb * H + eu];
      if (BEP == 1) {
        rpre = gates[(int64_t)eb * 2 * H + eu];
        hpre = h_prev[(int64_t)eb * H + eu];
        if (nf) nfpre = nf[eb];
      }
      if (BEP == 2) {
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
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: 16x16x4 has 40-cycle dependent latency
  float4 v[4], bq[4];
  if (PREF) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] : apd + 16 * q);
      bq[q] = bp4[(4 * q) * 16];
    }
  }
  for (int g0 = 0; g0 < KW / STEP_KDIV; g0 += 64) {       // KW % 64 == 0; one block = 64 k = 16 MFMAs
    if (!PREF) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] + g0 : apd + g0 + 16 * q);
        bq[q] = bp4[((g0 >> 2) + 4 * q) * 16];
      }
    }
    float4 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      b[q] = bq[q];
      if (STAGE) *reinterpret_cast<float4*>(lw + (srow + 4 * q) * 68 + 4 * sch) = v[q];
      else a[q] = v[q];
    }
    if (PREF) {
      const int gn = (g0 + 64 < KW) ? g0 + 64 : g0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = *reinterpret_cast<const float4*>(STAGE ? aps[q] + gn : apd + gn + 16 * q);
        bq[q] = bp4[((gn >> 2) + 4 * q) * 16];
      }
    }
    if (STAGE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const float4*>(lw + i * 68 + 16 * q + 4 * kq);
    }
    if (PREF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc1, 0, 0, 0);
    }
    if (PREF) __builtin_amdgcn_sched_barrier(0);
  }
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][4 * kq + r][i] = acc0[r] + acc1[r];
  __syncthreads();
  const int r = tid >> 4, u = tid & 15;
  const int b = m0 + r;
  if (b >= B) return;
  float* d = dh_prev + (int64_t)b * H + ug * 16 + u;
  const float sum = (red[0][r][u] + red[1][r][u]) + (red[2][r][u] + red[3][r][u]);
  if (BEP == 0) {
    *d = dpre + sum;
  } else if (BEP == 2) {                                     // sequence.hip lstm_gates_bwd_kernel for step t-1, fed from registers
    const int64_t idx = (int64_t)b * H + ug * 16 + u;
    float* dzr = gb.dz1 + (int64_t)b * 4 * H + ug * 16 + u;
    const float dh_in = dpre + sum;
    if (!((t - 1) < nfpre)) {
      dzr[0] = 0.f; dzr[H] = 0.f; dzr[2 * H] = 0.f; dzr[3 * H] = 0.f;
      gb.dc_out[idx] = dc1;
      gb.dh_out[idx] = dh_in;
    } else {
      const float gi = gpre[0], gj = gpre[1], gf = gpre[2], go = gpre[3];
      const float tc = tanhf(cn1);
      const float dht = dh_in + do1;
      const float dct = dc1 + dht * go * (1.0f - tc * tc);
      dzr[0] = dct * gj * gi * (1.0f - gi);
      dzr[H] = dct * gi * (1.0f - gj * gj);
      dzr[2 * H] = dct * cp1 * gf * (1.0f - gf);
      dzr[3 * H] = dht * tc * go * (1.0f - go);
      gb.dc_out[idx] = dct * gf;
      gb.dh_out[idx] = 0.f;
    }
  } else {
    const bool live = t < nfpre;
    const int64_t gi = (int64_t)b * 2 * H + ug * 16 + u;
    dzg[gi] = live ? sum * hpre * rpre * (1.0f - rpre) : 0.f;
    if (live) *d = dpre + sum * rpre;
  }
}

}  // namespace

namespace yt8m {

// loop form per kernel class (MODE of the templates): YT8M_STEP_MODE = four decimal digits "abcd" -> fwd G=4, fwd G<4,
// bwd K=4H, bwd K<4H; each digit 0..3 (bit 0 prefetch, bit 1 LDS-staged A).  Defaults chosen on the B=128, 2x1024 stack.
static int step_mode(int cls) {
  static int m[4] = {-1, -1, -1, -1};
  if (m[0] < 0) {
    const char* e = getenv("YT8M_STEP_MODE");
    const char* d = (e && strlen(e) == 4) ? e : "2122";
    for (int c = 0; c < 4; ++c) m[c] = (d[c] - '0') & 3;
  }
  return m[cls];
}
#define FWD_LAUNCH(G, EP, CLS, ...)                                                                                    \
  do {                                                                                                                 \
    switch (step_mode(CLS)) {                                                                                          \
      case 0: hipLaunchKernelGGL((lstm_step_fwd_kernel<G, EP, 0>), __VA_ARGS__); break;                                \
      case 1: hipLaunchKernelGGL((lstm_step_fwd_kernel<G, EP, 1>), __VA_ARGS__); break;                                \
      case 2: hipLaunchKernelGGL((lstm_step_fwd_kernel<G, EP, 2>), __VA_ARGS__); break;                                \
      default: hipLaunchKernelGGL((lstm_step_fwd_kernel<G, EP, 3>), __VA_ARGS__); break;                               \
    }                                                                                                                  \
  } while (0)
#define BWD_LAUNCH(BEP, CLS, ...)                                                                                      \
  do {                                                                                                                 \
    switch (step_mode(CLS)) {                                                                                          \
      case 0: hipLaunchKernelGGL((lstm_step_bwd_kernel<BEP, 0>), __VA_ARGS__); break;                                  \
      case 1: hipLaunchKernelGGL((lstm_step_bwd_kernel<BEP, 1>), __VA_ARGS__); break;                                  \
      case 2: hipLaunchKernelGGL((lstm_step_bwd_kernel<BEP, 2>), __VA_ARGS__); break;                                  \
      default: hipLaunchKernelGGL((lstm_step_bwd_kernel<BEP, 3>), __VA_ARGS__); break;                                 \
    }                                                                                                                  \
  } while (0)

bool lstm_fused_supported(int64_t B, int64_t H, int64_t workspace_bytes) {
  return H >= 128 && (H % 128) == 0 && B >= 1 && workspace_bytes >= (int64_t)sizeof(float) * H * 4 * H;
}

int lstm_pack(const float* Wh, int64_t ldw, float* Wp, float* Wq, int64_t H, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(2048), dim3(256), 0, s, Wh, ldw, Wp, Wq, (int)H, 4);
  return launch_status("lstm_pack_kernel");
}

int lstm_step_fwd(float* z, const float* Wp, const float* c_prev, const float* h_prev, float* c_new, float* h_new, float* out,
                  const int32_t* nf, int t, int64_t B, int64_t H, float fb, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 8));
  FWD_LAUNCH(4, EP_LSTM, 0, dim3(grid), dim3(256), 0, s, z, Wp, h_prev, c_prev, h_prev, c_new,
                     h_new, out, nf, t, (int)B, (int)H, fb, (const float*)nullptr, (float*)nullptr);
  return launch_status("lstm_step_fwd_kernel");
}

int lstm_step_bwd(const float* dz, const float* Wq, float* dh_prev, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  BWD_LAUNCH(0, 2, dim3(grid), dim3(256), 0, s, dz, Wq, dh_prev, (int)B, (int)H, (int)(4 * H),
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, 0, GateBwd{});
  return launch_status("lstm_step_bwd_kernel");
}

// product of step t (dz_t . Wh^T added to dh_prev) + gate backward of step t1 = t - 1 in one launch
int lstm_step_bwd_fused(const float* dz_t, const float* Wq, const float* dh_prev, const float* gates1, const float* c_prev1,
                        const float* c_new1, const float* dc_in, const float* dout1, float* dz1, float* dc_out, float* dh_out,
                        const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  GateBwd gb = {gates1, c_prev1, c_new1, dc_in, dout1, dz1, dc_out, dh_out};
  BWD_LAUNCH(2, 2, dim3(grid), dim3(256), 0, s, dz_t, Wq, const_cast<float*>(dh_prev), (int)B, (int)H, (int)(4 * H),
             (const float*)nullptr, (const float*)nullptr, (float*)nullptr, nf, t, gb);
  return launch_status("lstm_step_bwd_kernel<fused gates>");
}

// ---- the same packed-weight step products for the other cells (cells.hip) ----------------------------------------------
// G gates: packed sizes are H * G*H floats for both Wp and Wq.  Needs H % 256 == 0 (the K = H backward split).
bool cell_packed_supported(int64_t B, int64_t H) { return B >= 1 && H >= 256 && (H % 256) == 0; }

int cell_pack(const float* Wh, int64_t ldw, float* Wp, float* Wq, int64_t H, int G, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(2048), dim3(256), 0, s, Wh, ldw, Wp, Wq, (int)H, G);
  return launch_status("lstm_pack_kernel");
}

// z[B,4H] += h . Wh (packed, G = 4), nothing else
int cell_step_add4(float* z, const float* Wp, const float* h_prev, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 8));
  FWD_LAUNCH(4, EP_ADD, 0, dim3(grid), dim3(256), 0, s, z, Wp, h_prev, (const float*)nullptr, h_prev,
                     (float*)nullptr, (float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, 0, (int)B, (int)H, 0.f,
                     (const float*)nullptr, (float*)nullptr);
  return launch_status("lstm_step_fwd_kernel<add>");
}

int gru_step_gates(float* zg, const float* Wp_g, const float* h_prev, float* rh, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 16));
  FWD_LAUNCH(2, EP_GRU_GATES, 1, dim3(grid), dim3(256), 0, s, zg, Wp_g, h_prev, (const float*)nullptr,
                     h_prev, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, 0, (int)B, (int)H, 0.f,
                     (const float*)nullptr, rh);
  return launch_status("lstm_step_fwd_kernel<gru gates>");
}

int gru_step_cand(float* zc, const float* Wp_c, const float* rh, const float* zg, const float* h_prev, float* h_new, float* out,
                  const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 32));
  FWD_LAUNCH(1, EP_GRU_CAND, 1, dim3(grid), dim3(256), 0, s, zc, Wp_c, rh, (const float*)nullptr, h_prev,
                     (float*)nullptr, h_new, out, nf, t, (int)B, (int)H, 0.f, zg, (float*)nullptr);
  return launch_status("lstm_step_fwd_kernel<gru candidate>");
}

// dh_prev += dz[B,G*H] . Wh^T
int cell_step_bwd(const float* dz, const float* Wq, float* dh_prev, int64_t B, int64_t H, int G, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  BWD_LAUNCH(0, (G == 4 ? 2 : 3), dim3(grid), dim3(256), 0, s, dz, Wq, dh_prev, (int)B, (int)H, (int)(G * H),
                     (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, 0, GateBwd{});
  return launch_status("lstm_step_bwd_kernel");
}

// d(r*h) = dzc . Wc_h^T folded into dzg_r and dh_prev
int gru_step_bwd_cand(const float* dzc, const float* Wq_c, float* dh_prev, const float* zg, const float* h_prev, float* dzg,
                      const int32_t* nf, int t, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  BWD_LAUNCH(1, 3, dim3(grid), dim3(256), 0, s, dzc, Wq_c, dh_prev, (int)B, (int)H, (int)H, zg, h_prev,
                     dzg, nf, t, GateBwd{});
  return launch_status("lstm_step_bwd_kernel<gru candidate>");
}

}  // namespace yt8m
