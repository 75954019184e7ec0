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

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wp[ug][k][c], c = gate*8 + u  <-  Wh[k][gate*H + ug*8 + u]        (H/8 groups, k < H)
// Wq[ug][k][u]                  <-  Wh[ug*16 + u][k]                 (H/16 groups, k < 4H)
__global__ __launch_bounds__(256) void lstm_pack_kernel(const float* __restrict__ Wh, int64_t ldw, float* __restrict__ Wp,
                                                        float* __restrict__ Wq, int H) {
  const int64_t n = (int64_t)H * 4 * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    if (Wp) {
      const int c = (int)(e & 31);
      const int64_t r = e >> 5;           // ug*H + k
      const int k = (int)(r % H), ug = (int)(r / H);
      Wp[e] = Wh[(int64_t)k * ldw + (c >> 3) * H + ug * 8 + (c & 7)];
    }
    if (Wq) {
      const int u = (int)(e & 15);
      const int64_t r = e >> 4;           // ug*4H + k
      const int k = (int)(r % (4 * H)), ug = (int)(r / (4 * H));
      Wq[e] = Wh[(int64_t)(ug * 16 + u) * ldw + k];
    }
  }
}

__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float* __restrict__ z, const float* __restrict__ Wp,
                                                            const float* __restrict__ c_prev, const float* __restrict__ h_prev,
                                                            float* __restrict__ c_new, float* __restrict__ h_new,
                                                            float* __restrict__ out, const int32_t* __restrict__ nf, int t, int B,
                                                            int H, float fb) {
  __shared__ float red[4][32][33];
  const int groups = H >> 3;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 31, kh = lane >> 5;
  const int kq = H >> 2;                                   // K-range of one wave
  int row = m0 + i;
  if (row >= B) row = B - 1;                               // clamped rows feed output rows >= B only (never stored)
  const float* ap = h_prev + (int64_t)row * H + w * kq + 4 * kh;
  const float* bp = Wp + ((int64_t)ug * H + w * kq + 4 * kh) * 32 + i;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int g0 = 0; g0 < kq; g0 += 32) {       // kq % 32 == 0 (H % 128 == 0); 4 groups of 8 k in flight
    float4 a[4];
    float bb[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = g0 + 8 * q;
      a[q] = *reinterpret_cast<const float4*>(ap + g);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bb[q][jj] = bp[(g + jj) * 32];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bb[q][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bb[q][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bb[q][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bb[q][3], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * kh][i] = acc[r];
  __syncthreads();
  // thread -> (batch row, hidden unit): gate pre-activations = sum of the 4 K-parts + hoisted input projection
  const int r = tid >> 3, u = tid & 7;
  const int b = m0 + r;
  if (b >= B) return;
  const int unit = ug * 8 + u;
  const int64_t idx = (int64_t)b * H + unit;
  const bool live = nf ? (t < nf[b]) : true;
  if (!live) {
    c_new[idx] = c_prev[idx];
    h_new[idx] = h_prev[idx];
    if (out) out[idx] = 0.f;
    return;
  }
  float* zr = z + (int64_t)b * 4 * H + unit;
  float pre[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4)
    pre[g4] = zr[g4 * H] + ((red[0][r][g4 * 8 + u] + red[1][r][g4 * 8 + u]) + (red[2][r][g4 * 8 + u] + red[3][r][g4 * 8 + u]));
  const float gi = sigmoidf_(pre[0]);
  const float gj = tanhf(pre[1]);
  const float gf = sigmoidf_(pre[2] + fb);
  const float go = sigmoidf_(pre[3]);
  const float c = c_prev[idx] * gf + gi * gj;
  const float hn = tanhf(c) * go;
  zr[0] = gi; zr[H] = gj; zr[2 * H] = gf; zr[3 * H] = go;
  c_new[idx] = c;
  h_new[idx] = hn;
  if (out) out[idx] = hn;
}

// dh_prev[B,H] += dz[B,4H] . Wh^T     (16 rows x 16 units per workgroup, v_mfma_f32_16x16x4_f32, K = 4H over 4 waves)
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ Wq,
                                                            float* __restrict__ dh_prev, int B, int H) {
  __shared__ float red[4][16][17];
  const int groups = H >> 4;
  const int ug = blockIdx.x % groups, rt = blockIdx.x / groups;
  const int m0 = rt * 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int K4 = 4 * H;
  int row = m0 + i;
  if (row >= B) row = B - 1;
  const float* ap = dz + (int64_t)row * K4 + w * H + 4 * kq;
  const float* bp = Wq + ((int64_t)ug * K4 + w * H + 4 * kq) * 16 + i;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: 16x16x4 has 40-cycle dependent latency
  for (int g0 = 0; g0 < H; g0 += 64) {        // H % 64 == 0; 4 groups of 16 k in flight
    float4 a[4];
    float bb[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = g0 + 16 * q;
      a[q] = *reinterpret_cast<const float4*>(ap + g);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bb[q][jj] = bp[(g + jj) * 16];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, bb[q][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, bb[q][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, bb[q][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, bb[q][3], acc1, 0, 0, 0);
    }
  }
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][4 * kq + r][i] = acc0[r] + acc1[r];
  __syncthreads();
  const int r = tid >> 4, u = tid & 15;
  const int b = m0 + r;
  if (b >= B) return;
  float* d = dh_prev + (int64_t)b * H + ug * 16 + u;
  *d += (red[0][r][u] + red[1][r][u]) + (red[2][r][u] + red[3][r][u]);
}

}  // namespace

namespace yt8m {

bool lstm_fused_supported(int64_t B, int64_t H, int64_t workspace_bytes) {
  return H >= 128 && (H % 128) == 0 && B >= 1 && workspace_bytes >= (int64_t)sizeof(float) * H * 4 * H;
}

int lstm_pack(const float* Wh, int64_t ldw, float* Wp, float* Wq, int64_t H, hipStream_t s) {
  hipLaunchKernelGGL(lstm_pack_kernel, dim3(2048), dim3(256), 0, s, Wh, ldw, Wp, Wq, (int)H);
  return launch_status("lstm_pack_kernel");
}

int lstm_step_fwd(float* z, const float* Wp, const float* c_prev, const float* h_prev, float* c_new, float* h_new, float* out,
                  const int32_t* nf, int t, int64_t B, int64_t H, float fb, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 31) / 32) * (H / 8));
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(grid), dim3(256), 0, s, z, Wp, c_prev, h_prev, c_new, h_new, out, nf, t, (int)B,
                     (int)H, fb);
  return launch_status("lstm_step_fwd_kernel");
}

int lstm_step_bwd(const float* dz, const float* Wq, float* dh_prev, int64_t B, int64_t H, hipStream_t s) {
  const unsigned grid = (unsigned)(((B + 15) / 16) * (H / 16));
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(grid), dim3(256), 0, s, dz, Wq, dh_prev, (int)B, (int)H);
  return launch_status("lstm_step_bwd_kernel");
}

}  // namespace yt8m
