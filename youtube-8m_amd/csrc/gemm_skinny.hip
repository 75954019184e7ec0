// gemm_skinny.hip -- fully-connected layers with a handful of outputs over very many rows (gfx950, wave64).
// The attention logits of W/all_frame_models/lstm_attention_max_pooling_model.py:51-56 are slim.fully_connected on
// [B*300, 1152 + 1024] -> 8: M = 38 400 ... 307 200 rows, N = 8.  On 128 x 128 MFMA tiles 94 % of the matrix work is padding
// (measured 5.5-7.2 TFLOP/s "useful"); the layer is a pure stream over x, so it is done on the vector ALU at the HBM rate:
//   forward   y[M,N]  (+)= x[M,K] . W[K,N] (+ bias)       8 B of FLOP per byte of x -> HBM-bound
//   dW        dW[K,N] (+)= x[M,K]^T . dy[M,N]             x read once; row chunks -> partials -> fixed-order reduction
//   dx        dx[M,K] (+)= dy[M,N] . W[K,N]^T             output-write-bound
// N <= 16 (padded to NT = 8 or 16 in registers), K % 4 == 0, fp32.  Algorithmic bytes: 4*M*K (+ 4*M*N) per pass.
//
// uint8 rows (round 3): the same three kernels read the reader's RAW frames q [M, K] uint8 (W/readers.py:178-187) for
//   x = diag(rs) (a0 q + c0 1 1^T),  a0 = 4/255, c0 = 4/512 - 2 (W/utils.py:23-38 Dequantize),  rs = 1 / ||a0 q + c0|| or 0 for padding
// (the l2-normalised, masked frames of W/all_feature_transform/default_transformer.py:4-8) so that no fp32 [B, F, D] tensor exists:
//   fwd : y = rs (.) (a0 q.W + c0 colsum(W)) + bias        dW : a0 q^T (rs (.) dy) + c0 1 (x) colsum(rs (.) dy)
// 1 B instead of 4 B per input element; q in [0, 255] is exact in fp32.
#include <type_traits>
#include "common.h"

namespace {

constexpr float U8_A0 = 4.0f / 255.0f, U8_C0 = 4.0f / 512.0f - 2.0f;

__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 unpack4(uint32_t u) {
  return float4{(float)(u & 255u), (float)((u >> 8) & 255u), (float)((u >> 16) & 255u), (float)(u >> 24)};
}
__device__ __forceinline__ float4 load4(const uint8_t* p) { return unpack4(*reinterpret_cast<const uint32_t*>(p)); }

// ---- forward: a wave owns 4 rows at a time; lane l covers k = 256 j + 4 l + e; W lives in LDS as [j][q = 2e+h][lane][4]
//      (n = 4h..4h+3 for NT = 8; q = 4e + h for NT = 16) so that the per-lane b128 reads are conflict-free -------------------
// Batched form (blockIdx.y = batch, element strides bx / bw / by): the attention pooling backward dw[b] = x[b] . dC[b]^T of
// lstm_attention_max_pooling_model.py:63, with W[k][n] = dC[b][n][k] addressed through (wsk, wsn).
// XT = uint8_t: rs [M] (batch stride brs, may be NULL) and cs [N] = colsum(W) (batch stride bcs) complete the affine form above.
// JMAX: 4-byte words of a uint8 row a lane holds (K <= 256 JMAX): 5 covers the 1152-wide frames with 24 fewer registers than 8
// (three waves per SIMD instead of two).
template <int NT, typename XT = float, int JMAX = 8>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const XT* __restrict__ x, int64_t ldx, const float* __restrict__ W,
                                                         int64_t wsk, int64_t wsn, const float* __restrict__ bias,
                                                         float* __restrict__ y, int64_t ldy, int64_t M, int K, int N, float beta,
                                                         int rows_per_wg, int64_t bx, int64_t bw, int64_t by,
                                                         const float* __restrict__ rs = nullptr, int64_t brs = 0,
                                                         const float* __restrict__ cs = nullptr, int64_t bcs = 0) {
  extern __shared__ __attribute__((aligned(16))) float wl[];           // J * (NT) * 64 * 4 floats
  constexpr bool U8 = std::is_same<XT, uint8_t>::value;
  if constexpr (U8) { if (rs) rs += (int64_t)blockIdx.y * brs; if (cs) cs += (int64_t)blockIdx.y * bcs; }
  x += (int64_t)blockIdx.y * bx;
  W += (int64_t)blockIdx.y * bw;
  y += (int64_t)blockIdx.y * by;
  constexpr int QN = NT / 4;                                           // float4 groups per k
  constexpr int R = 4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int J = (K + 255) >> 8;
  for (int e = tid; e < J * 4 * QN * 64; e += 256) {                   // e -> (j, q, l); element group of 4 n
    const int l = e & 63, q = (e >> 6) % (4 * QN), j = e / (64 * 4 * QN);
    const int k = 256 * j + 4 * l + q / QN, n0 = 4 * (q % QN);
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (k < K) {
      if (n0 + 0 < N) v.x = W[(int64_t)k * wsk + (n0 + 0) * wsn];
      if (n0 + 1 < N) v.y = W[(int64_t)k * wsk + (n0 + 1) * wsn];
      if (n0 + 2 < N) v.z = W[(int64_t)k * wsk + (n0 + 2) * wsn];
      if (n0 + 3 < N) v.w = W[(int64_t)k * wsk + (n0 + 3) * wsn];
    }
    *reinterpret_cast<float4*>(wl + (int64_t)e * 4) = v;
  }
  __syncthreads();
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r_end = r_begin + rows_per_wg < M ? r_begin + rows_per_wg : M;
  // uint8 rows: a whole row group (R rows x up to 2048 bytes = JM words per lane) is fetched one group AHEAD of the arithmetic --
  // a 4-byte load per lane carries a quarter of the float4's bytes, so the loads in flight, not the bytes, bound the stream
  constexpr int JM = U8 ? JMAX : 1;
  uint32_t nxt[R][JM];
  auto fetch = [&](int64_t r0) {
    if constexpr (U8) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(x + (r0 + r < M ? r0 + r : M - 1) * ldx) + lane;
#pragma unroll
        for (int j = 0; j < JM; ++j) nxt[r][j] = (j < J && 256 * j + 4 * lane < K) ? p[64 * j] : 0u;
      }
    }
  };
  if (r_begin + w * R < r_end) fetch(r_begin + w * R);
  for (int64_t r0 = r_begin + w * R; r0 < r_end; r0 += 4 * R) {
    float acc[R * NT];
#pragma unroll
    for (int i = 0; i < R * NT; ++i) acc[i] = 0.f;
    auto fma_block = [&](int j, const float4 (&xv)[R]) {
      const float* wj = wl + ((int64_t)j * 4 * QN * 64 + lane) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int h = 0; h < QN; ++h) {
          const float4 wv = *reinterpret_cast<const float4*>(wj + (e * QN + h) * 256);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float xk = e == 0 ? xv[r].x : e == 1 ? xv[r].y : e == 2 ? xv[r].z : xv[r].w;
            acc[r * NT + 4 * h + 0] = fmaf(xk, wv.x, acc[r * NT + 4 * h + 0]);
            acc[r * NT + 4 * h + 1] = fmaf(xk, wv.y, acc[r * NT + 4 * h + 1]);
            acc[r * NT + 4 * h + 2] = fmaf(xk, wv.z, acc[r * NT + 4 * h + 2]);
            acc[r * NT + 4 * h + 3] = fmaf(xk, wv.w, acc[r * NT + 4 * h + 3]);
          }
        }
      }
    };
    if constexpr (U8) {
      uint32_t cur[R][JM];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < JM; ++j) cur[r][j] = nxt[r][j];
      if (r0 + 4 * R < r_end) fetch(r0 + 4 * R);
#pragma unroll
      for (int j = 0; j < JM; ++j) {
        if (j < J) {
          float4 xv[R];
#pragma unroll
          for (int r = 0; r < R; ++r) xv[r] = unpack4(cur[r][j]);
          fma_block(j, xv);
        }
      }
    } else {
      const XT* xr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) xr[r] = x + (r0 + r < M ? r0 + r : M - 1) * ldx + 4 * lane;
      for (int j = 0; j < J; ++j) {
        float4 xv[R];
        const bool in = 256 * j + 4 * lane < K;                        // K % 4 == 0: a float4 is inside or outside
#pragma unroll
        for (int r = 0; r < R; ++r) xv[r] = in ? load4(xr[r] + 256 * j) : float4{0.f, 0.f, 0.f, 0.f};
        fma_block(j, xv);
      }
    }
    // transposing butterfly: R*NT values x 64 lanes -> every value summed over the wave with R*NT (+ tail) shuffles instead
    // of 6 per value.  After the masks 32..(64/(R*NT)) lane L holds value index L / (64 / (R*NT)).
    constexpr int NV = R * NT;                                         // 32 or 64
    int n = NV;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      if (n > 1) {
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
          if (i < n / 2) {
            const float a = acc[i], b = acc[i + n / 2];
            const float recv = __shfl_xor(up ? a : b, m, 64);
            acc[i] = (up ? b : a) + recv;
          }
        }
        n >>= 1;
      } else {
        acc[0] += __shfl_xor(acc[0], m, 64);
      }
    }
    constexpr int LPV = 64 / NV;                                       // lanes per value (2 for NV = 32, 1 for 64)
    const int vi = lane / LPV;
    if ((lane % LPV) == 0) {
      const int r = vi / NT, nn = vi % NT;
      if (r0 + r < r_end && nn < N) {
        float* yp = y + (r0 + r) * ldy + nn;
        float v = acc[0];
        if constexpr (U8) {
          v = fmaf(U8_A0, v, cs ? U8_C0 * cs[nn] : 0.f);
          if (rs) v *= rs[r0 + r];
        }
        v += bias ? bias[nn] : 0.f;
        if (beta != 0.f) v += *yp;
        *yp = v;
      }
    }
  }
}

// ---- dW / dx: a thread owns 4 consecutive k of a 1024-wide k slice; the dy rows of the chunk sit in LDS (broadcast reads) ----
// Batched form (blockIdx.z = batch): `direct` (dW only, one row chunk) writes out[b][n * ldo + k] itself -- the attention
// pooling C[b] = w[b]^T . x[b] of lstm_attention_max_pooling_model.py:63 ([A, H] per video, k contiguous).
// XT = uint8_t (dW / pooling only): the staged dy rows are scaled by rs [M] (batch stride brs, may be NULL) and the result is
// a0 acc + c0 colsum(rs (.) dy) -- the affine form of the file header.
template <int NT, bool DX, bool BATCH, typename XT = float>
__global__ __launch_bounds__(256) void skinny_bwd_kernel(const XT* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                         int64_t ldy, const float* __restrict__ W, int64_t wsk, int64_t wsn,
                                                         float* __restrict__ out, int64_t ldo, int64_t M, int K, int N,
                                                         float beta, int rows_per_chunk, int kslice, int direct, int64_t bx,
                                                         int64_t bdy, int64_t bw, int64_t bo, const float* __restrict__ rs = nullptr,
                                                         int64_t brs = 0) {
  constexpr int RC = 64;                                               // dy rows staged per pass
  constexpr bool U8 = std::is_same<XT, uint8_t>::value;
  static_assert(!(U8 && DX), "the frames are an input: no dx from uint8 rows");
  __shared__ __attribute__((aligned(16))) float dyl[RC * NT];
  const int tid = threadIdx.x;
  if constexpr (U8 && BATCH) if (rs) rs += (int64_t)blockIdx.z * brs;
  float sdy[U8 ? NT : 1];
#pragma unroll
  for (int i = 0; i < (U8 ? NT : 1); ++i) sdy[i] = 0.f;
  if (BATCH) {                      // a separate instantiation: the un-batched kernels keep their (faster) code
    if (!DX) x += (int64_t)blockIdx.z * bx;
    dy += (int64_t)blockIdx.z * bdy;
    if (DX) W += (int64_t)blockIdx.z * bw;
    out += (int64_t)blockIdx.z * bo;
  }
  const int k0 = blockIdx.x * kslice + 4 * tid;                        // kslice <= 1024, % 4 == 0: balanced k slices
  const bool kin = 4 * tid < kslice && k0 < K;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r_end = r_begin + rows_per_chunk < M ? r_begin + rows_per_chunk : M;
  float acc[4 * NT];                                                   // dW: partial sums; dx: the W[k0..k0+3][0..NT) block
#pragma unroll
  for (int i = 0; i < 4 * NT; ++i) acc[i] = 0.f;
  if (DX && kin) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[e * NT + n] = n < N ? W[(int64_t)(k0 + e) * wsk + n * wsn] : 0.f;
  }
  for (int64_t rb = r_begin; rb < r_end; rb += RC) {
    __syncthreads();
    for (int e = tid; e < RC * NT; e += 256) {
      const int64_t r = rb + e / NT;
      const int n = e % NT;
      float v = (r < r_end && n < N) ? dy[r * ldy + n] : 0.f;
      if constexpr (U8) { if (rs && r < r_end) v *= rs[r]; }
      dyl[e] = v;
    }
    __syncthreads();
    if (!kin) continue;
    const int nr = (int)(r_end - rb < RC ? r_end - rb : RC);
    if constexpr (U8) {                                                // 8 rows of 4-byte loads in flight per thread (see forward)
      for (int rr = 0; rr < nr; rr += 8) {
        uint32_t xw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xw[i] = rr + i < nr ? *reinterpret_cast<const uint32_t*>(x + (rb + rr + i) * ldx + k0) : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                  // staged rows past r_end are zero
          float dv[NT];
#pragma unroll
          for (int q = 0; q < NT / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(dyl + (rr + i) * NT + 4 * q);
            dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
          }
          const float4 xv = unpack4(xw[i]);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            sdy[n] += dv[n];
            acc[0 * NT + n] = fmaf(xv.x, dv[n], acc[0 * NT + n]);
            acc[1 * NT + n] = fmaf(xv.y, dv[n], acc[1 * NT + n]);
            acc[2 * NT + n] = fmaf(xv.z, dv[n], acc[2 * NT + n]);
            acc[3 * NT + n] = fmaf(xv.w, dv[n], acc[3 * NT + n]);
          }
        }
      }
      continue;
    }
    for (int r = 0; r < nr; ++r) {
      float dv[NT];
#pragma unroll
      for (int q = 0; q < NT / 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(dyl + r * NT + 4 * q);
        dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
      }
      if (DX) {
        float4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          o.x = fmaf(dv[n], acc[0 * NT + n], o.x);
          o.y = fmaf(dv[n], acc[1 * NT + n], o.y);
          o.z = fmaf(dv[n], acc[2 * NT + n], o.z);
          o.w = fmaf(dv[n], acc[3 * NT + n], o.w);
        }
        float4* op = reinterpret_cast<float4*>(out + (rb + r) * ldo + k0);
        if (beta != 0.f) { const float4 p = *op; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        *op = o;
      } else {
        const float4 xv = load4(x + (rb + r) * ldx + k0);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[0 * NT + n] = fmaf(xv.x, dv[n], acc[0 * NT + n]);
          acc[1 * NT + n] = fmaf(xv.y, dv[n], acc[1 * NT + n]);
          acc[2 * NT + n] = fmaf(xv.z, dv[n], acc[2 * NT + n]);
          acc[3 * NT + n] = fmaf(xv.w, dv[n], acc[3 * NT + n]);
        }
      }
    }
  }
  if constexpr (U8) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e * NT + n] = fmaf(U8_A0, acc[e * NT + n], U8_C0 * sdy[n]);
    }
  }
  if (BATCH && !DX && kin && direct) {                                          // single row chunk: out[n][k0..k0+3] (+)= acc
    for (int n = 0; n < N; ++n) {
      float4* op = reinterpret_cast<float4*>(out + (int64_t)n * ldo + k0);
      float4 o = {acc[0 * NT + n], acc[1 * NT + n], acc[2 * NT + n], acc[3 * NT + n]};
      if (beta != 0.f) { const float4 pv = *op; o.x += pv.x; o.y += pv.y; o.z += pv.z; o.w += pv.w; }
      *op = o;
    }
  } else if (!DX && kin) {                                             // partial of this row chunk: out = ws[chunk][K][NT]
    float* p = out + ((int64_t)blockIdx.y * K + k0) * NT;
#pragma unroll
    for (int i = 0; i < 4 * NT; i += 4) *reinterpret_cast<float4*>(p + i) = float4{acc[i], acc[i + 1], acc[i + 2], acc[i + 3]};
  }
}

// 32 elements x 8 chunk lanes per workgroup: chunk lane c sums chunks c, c + 8, ... and the 8 partial sums are added in lane
// order -- a fixed order, so the result is deterministic; K * NT / 32 workgroups instead of K * NT / 256
template <int NT>
__global__ __launch_bounds__(256) void skinny_dw_reduce_kernel(const float* __restrict__ ws, int chunks, float* __restrict__ dW,
                                                               int64_t lddw, int K, int N, float beta) {
  __shared__ float part[8][33];
  const int el = threadIdx.x & 31, cl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;                                  // (k, n) of the padded [K][NT] image
  float s = 0.f;
  if (e < K * NT)
    for (int c = cl; c < chunks; c += 8) s += ws[(int64_t)c * K * NT + e];
  part[cl][el] = s;
  __syncthreads();
  if (cl != 0 || e >= K * NT) return;
  const int k = e / NT, n = e % NT;
  if (n >= N) return;
#pragma unroll
  for (int c = 1; c < 8; ++c) s += part[c][el];
  float* d = dW + (int64_t)k * lddw + n;
  *d = beta != 0.f ? *d + s : s;
}

int chunk_rows(int64_t M, int chunks) {
  int64_t r = (M + chunks - 1) / chunks;
  r = (r + 63) / 64 * 64;
  return (int)(r < 64 ? 64 : r);
}

// k slices of equal width (<= 1024 = 256 threads x float4) and enough row chunks for >= ~1536 workgroups
struct BwdPlan { int nslices, kslice, chunks, rows; };
BwdPlan bwd_plan(int64_t M, int64_t K) {
  BwdPlan p;
  p.nslices = (int)((K + 1023) / 1024);
  p.kslice = (int)(((K + p.nslices - 1) / p.nslices + 3) / 4 * 4);
  int want = (1536 + p.nslices - 1) / p.nslices;
  if (want > 512) want = 512;
  if (want < 64) want = 64;
  p.rows = chunk_rows(M, want);
  p.chunks = (int)((M + p.rows - 1) / p.rows);
  return p;
}

}  // namespace

using namespace yt8m;

static int skinny_check(int64_t M, int64_t K, int64_t N, float beta) {
  YT8M_REQUIRE(M >= 0 && K >= 0 && N >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(N <= 16, YT8M_E_SHAPE, "skinny GEMM needs N <= 16");
  YT8M_REQUIRE(K % 4 == 0 && K < (1 << 24), YT8M_E_SHAPE, "skinny GEMM needs K % 4 == 0");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  return YT8M_OK;
}

extern "C" int yt8m_skinny_supported(int64_t M, int64_t K, int64_t N) {
  const int NT = N <= 8 ? 8 : 16;
  return N >= 1 && N <= 16 && K >= 4 && K % 4 == 0 && M >= 1 && ((K + 255) / 256) * 256 * NT * 4 <= 144 * 1024;
}

extern "C" int64_t yt8m_skinny_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  const int NT = N <= 8 ? 8 : 16;
  const BwdPlan p = bwd_plan(M, K);
  return (int64_t)(p.chunks > 0 ? p.chunks : 1) * K * NT * (int64_t)sizeof(float);
}

extern "C" int yt8m_skinny_fwd_f32(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* y,
                                   int64_t ldy, int64_t M, int64_t K, int64_t N, float beta, yt8m_stream_t stream) {
  int rc = skinny_check(M, K, N, beta);
  if (rc != YT8M_OK) return rc;
  if (M == 0 || N == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_skinny_supported(M, K, N), YT8M_E_SHAPE, "K too large for the LDS-resident weight image");
  YT8M_REQUIRE(x && W && y, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldx >= K && ldx % 4 == 0 && ldw >= N && ldy >= N && ((uintptr_t)x & 15) == 0, YT8M_E_SHAPE, "bad leading dimension / alignment");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const int J = (int)((K + 255) / 256);
  int rows_per_wg = (int)((M + 1023) / 1024);                          // ~1024 workgroups
  rows_per_wg = (rows_per_wg + 15) / 16 * 16;
  const unsigned grid = (unsigned)((M + rows_per_wg - 1) / rows_per_wg);
  if (N <= 8) {
    const size_t lds = (size_t)J * 8 * 64 * 4 * sizeof(float);
    static DeviceOnce once8;
    YT8M_HIP_CHECK(once8.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8>), 144 * 1024));
    hipLaunchKernelGGL(skinny_fwd_kernel<8>, dim3(grid), dim3(256), lds, s, x, ldx, W, ldw, (int64_t)1, bias, y, ldy, M, (int)K, (int)N,
                       beta, rows_per_wg, (int64_t)0, (int64_t)0, (int64_t)0);
  } else {
    const size_t lds = (size_t)J * 16 * 64 * 4 * sizeof(float);
    static DeviceOnce once16;
    YT8M_HIP_CHECK(once16.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<16>), 144 * 1024));
    hipLaunchKernelGGL(skinny_fwd_kernel<16>, dim3(grid), dim3(256), lds, s, x, ldx, W, ldw, (int64_t)1, bias, y, ldy, M, (int)K, (int)N,
                       beta, rows_per_wg, (int64_t)0, (int64_t)0, (int64_t)0);
  }
  return launch_status("skinny_fwd_kernel");
}

extern "C" int yt8m_skinny_dw_f32(const float* x, int64_t ldx, const float* dy, int64_t ldy, float* dW, int64_t lddw, int64_t M,
                                  int64_t K, int64_t N, float beta, void* workspace, int64_t workspace_bytes,
                                  yt8m_stream_t stream) {
  int rc = skinny_check(M, K, N, beta);
  if (rc != YT8M_OK) return rc;
  if (K == 0 || N == 0) return YT8M_OK;
  YT8M_REQUIRE(dW && (M == 0 || (x && dy)), YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(ldx >= K && ldx % 4 == 0 && ldy >= N && lddw >= N && ((uintptr_t)x & 15) == 0, YT8M_E_SHAPE, "bad leading dimension / alignment");
  YT8M_REQUIRE(workspace && workspace_bytes >= yt8m_skinny_workspace_bytes(M, K, N), YT8M_E_BADARG, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const BwdPlan pl = bwd_plan(M, K);
  const int rows = pl.rows, chunks = pl.chunks, kslice = pl.kslice;
  float* ws = static_cast<float*>(workspace);
  const dim3 grid((unsigned)pl.nslices, (unsigned)(chunks > 0 ? chunks : 1));
  if (N <= 8) {
    if (chunks > 0)
      hipLaunchKernelGGL((skinny_bwd_kernel<8, false, false>), grid, dim3(256), 0, s, x, ldx, dy, ldy, (const float*)nullptr, (int64_t)0,
                         (int64_t)0, ws, (int64_t)0, M, (int)K, (int)N, 0.f, rows, kslice, 0, (int64_t)0, (int64_t)0, (int64_t)0,
                         (int64_t)0);
    hipLaunchKernelGGL(skinny_dw_reduce_kernel<8>, dim3((unsigned)((K * 8 + 31) / 32)), dim3(256), 0, s, ws, chunks, dW, lddw, (int)K,
                       (int)N, beta);
  } else {
    if (chunks > 0)
      hipLaunchKernelGGL((skinny_bwd_kernel<16, false, false>), grid, dim3(256), 0, s, x, ldx, dy, ldy, (const float*)nullptr, (int64_t)0,
                         (int64_t)0, ws, (int64_t)0, M, (int)K, (int)N, 0.f, rows, kslice, 0, (int64_t)0, (int64_t)0, (int64_t)0,
                         (int64_t)0);
    hipLaunchKernelGGL(skinny_dw_reduce_kernel<16>, dim3((unsigned)((K * 16 + 31) / 32)), dim3(256), 0, s, ws, chunks, dW, lddw,
                       (int)K, (int)N, beta);
  }
  return launch_status("skinny_bwd_kernel<dW>");
}

extern "C" int yt8m_skinny_dx_f32(const float* dy, int64_t ldy, const float* W, int64_t ldw, float* dx, int64_t lddx, int64_t M,
                                  int64_t K, int64_t N, float beta, yt8m_stream_t stream) {
  int rc = skinny_check(M, K, N, beta);
  if (rc != YT8M_OK) return rc;
  if (M == 0 || K == 0) return YT8M_OK;
  YT8M_REQUIRE(dy && W && dx, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(lddx >= K && lddx % 4 == 0 && ldy >= N && ldw >= N && ((uintptr_t)dx & 15) == 0, YT8M_E_SHAPE, "bad leading dimension / alignment");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const BwdPlan pl = bwd_plan(M, K);
  const int rows = pl.rows, kslice = pl.kslice;
  const dim3 grid((unsigned)pl.nslices, (unsigned)pl.chunks);
  if (N <= 8)
    hipLaunchKernelGGL((skinny_bwd_kernel<8, true, false>), grid, dim3(256), 0, s, (const float*)nullptr, (int64_t)0, dy, ldy, W, ldw,
                       (int64_t)1, dx, lddx, M, (int)K, (int)N, beta, rows, kslice, 0, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);
  else
    hipLaunchKernelGGL((skinny_bwd_kernel<16, true, false>), grid, dim3(256), 0, s, (const float*)nullptr, (int64_t)0, dy, ldy, W, ldw,
                       (int64_t)1, dx, lddx, M, (int)K, (int)N, beta, rows, kslice, 0, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);
  return launch_status("skinny_bwd_kernel<dx>");
}

// ---- attention pooling over frames (W/all_frame_models/lstm_attention_max_pooling_model.py:63: einsum("ijk,ijl->ikl")) ---------
// w [B,F,A] (A <= 16), x [B,F,H] (H % 4 == 0), C [B,A,H], all dense row-major fp32.
//   fwd : C[b]  = w[b]^T . x[b]                 x streamed once
//   bwd : dw[b] = x[b] . dC[b]^T   (optional),  dx[b] = w[b] . dC[b]   (optional)
extern "C" int yt8m_attn_pool_supported(int64_t B, int64_t F, int64_t A, int64_t H) {
  const int NT = A <= 8 ? 8 : 16;
  return B >= 1 && B <= 65535 && F >= 1 && A >= 1 && A <= 16 && H >= 4 && H % 4 == 0 && ((H + 255) / 256) * 256 * NT * 4 <= 144 * 1024;
}

extern "C" int yt8m_attn_pool_fwd(const float* w, const float* x, float* C, int64_t B, int64_t F, int64_t A, int64_t H,
                                  yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * A * H == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_attn_pool_supported(B, F, A, H), YT8M_E_SHAPE, "attention pooling needs A <= 16, H % 4 == 0");
  YT8M_REQUIRE(w && x && C && (((uintptr_t)x | (uintptr_t)C) & 15) == 0, YT8M_E_BADARG, "null / unaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const int nsl = (int)((H + 1023) / 1024);
  const int kslice = (int)(((H + nsl - 1) / nsl + 3) / 4 * 4);
  const dim3 grid((unsigned)nsl, 1, (unsigned)B);
  const int rows = (int)((F + 63) / 64 * 64);
  if (A <= 8)
    hipLaunchKernelGGL((skinny_bwd_kernel<8, false, true>), grid, dim3(256), 0, s, x, H, w, A, (const float*)nullptr, (int64_t)0, (int64_t)0, C,
                       H, F, (int)H, (int)A, 0.f, rows, kslice, 1, F * H, F * A, (int64_t)0, A * H);
  else
    hipLaunchKernelGGL((skinny_bwd_kernel<16, false, true>), grid, dim3(256), 0, s, x, H, w, A, (const float*)nullptr, (int64_t)0, (int64_t)0, C,
                       H, F, (int)H, (int)A, 0.f, rows, kslice, 1, F * H, F * A, (int64_t)0, A * H);
  return launch_status("skinny_bwd_kernel<pool>");
}

extern "C" int yt8m_attn_pool_bwd(const float* w, const float* x, const float* dC, float* dw, float* dx, int64_t B, int64_t F,
                                  int64_t A, int64_t H, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * A * H == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_attn_pool_supported(B, F, A, H), YT8M_E_SHAPE, "attention pooling needs A <= 16, H % 4 == 0");
  YT8M_REQUIRE(dC && (!dw || x) && (!dx || w), YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((((uintptr_t)x | (uintptr_t)dx) & 15) == 0, YT8M_E_BADARG, "unaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  if (dw) {                                            // dw[b][f][a] = sum_h x[b][f][h] * dC[b][a][h]: W[k][n] = dC[b][n][k]
    const int J = (int)((H + 255) / 256);
    int rows_per_wg = (int)((F + 3) / 4);              // ~4 workgroups per video
    rows_per_wg = (rows_per_wg + 15) / 16 * 16;
    const dim3 grid((unsigned)((F + rows_per_wg - 1) / rows_per_wg), (unsigned)B);
    if (A <= 8) {
      static DeviceOnce once;
      YT8M_HIP_CHECK(once.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8>), 144 * 1024));
      hipLaunchKernelGGL(skinny_fwd_kernel<8>, grid, dim3(256), (size_t)J * 8 * 64 * 4 * sizeof(float), s, x, H, dC, (int64_t)1, H,
                         (const float*)nullptr, dw, A, F, (int)H, (int)A, 0.f, rows_per_wg, F * H, A * H, F * A);
    } else {
      static DeviceOnce once;
      YT8M_HIP_CHECK(once.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<16>), 144 * 1024));
      hipLaunchKernelGGL(skinny_fwd_kernel<16>, grid, dim3(256), (size_t)J * 16 * 64 * 4 * sizeof(float), s, x, H, dC, (int64_t)1, H,
                         (const float*)nullptr, dw, A, F, (int)H, (int)A, 0.f, rows_per_wg, F * H, A * H, F * A);
    }
  }
  if (dx) {                                            // dx[b][f][h] = sum_a w[b][f][a] * dC[b][a][h]
    const int nsl = (int)((H + 1023) / 1024);
    const int kslice = (int)(((H + nsl - 1) / nsl + 3) / 4 * 4);
    const dim3 grid((unsigned)nsl, 1, (unsigned)B);
    const int rows = (int)((F + 63) / 64 * 64);
    if (A <= 8)
      hipLaunchKernelGGL((skinny_bwd_kernel<8, true, true>), grid, dim3(256), 0, s, (const float*)nullptr, (int64_t)0, w, A, dC, (int64_t)1, H,
                         dx, H, F, (int)H, (int)A, 0.f, rows, kslice, 0, (int64_t)0, F * A, A * H, F * H);
    else
      hipLaunchKernelGGL((skinny_bwd_kernel<16, true, true>), grid, dim3(256), 0, s, (const float*)nullptr, (int64_t)0, w, A, dC, (int64_t)1, H,
                         dx, H, F, (int)H, (int)A, 0.f, rows, kslice, 0, (int64_t)0, F * A, A * H, F * H);
  }
  return launch_status("attention pooling backward");
}

// ---- the same layers on the reader's raw uint8 frames (file header: x = diag(rs) (a0 q + c0)) -------------------------------------
namespace {

// rs[row] = 1 / max(||a0 q_row + c0||, sqrt(eps)), 0 for padding rows (f >= num_frames[b]); one wave per frame row, the arithmetic
// and summation order of dequant_l2norm_kernel (csrc/elementwise.hip) and u8_frames_tm_kernel (csrc/u8proj.hip).
__global__ __launch_bounds__(256) void u8_frame_scales_kernel(const uint8_t* __restrict__ q, const int32_t* __restrict__ nf,
                                                              float* __restrict__ rs, int64_t rows, int F, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = (int)(row / F), f = (int)(row - (int64_t)b * F);
  const bool live = nf ? (f < nf[b]) : true;
  float ss = 0.f;
  if (live) {
    const uint32_t* qr = reinterpret_cast<const uint32_t*>(q + row * D);
    for (int c0 = 0; c0 < (D >> 2); c0 += 512) {                       // 8 loads in flight per lane
      uint32_t w[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) w[it] = c0 + lane + 64 * it < (D >> 2) ? qr[c0 + lane + 64 * it] : 0u;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (c0 + lane + 64 * it < (D >> 2)) {
#pragma unroll
          for (int k = 0; k < 4; ++k) { const float v = fmaf((float)((w[it] >> (8 * k)) & 255u), U8_A0, U8_C0); ss += v * v; }
        }
      }
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) rs[row] = live ? rsqrtf(fmaxf(ss, eps)) : 0.f;
}

int u8_check(const void* q, int64_t ldq, int64_t K) {
  YT8M_REQUIRE(K <= 2048, YT8M_E_SHAPE, "uint8 rows: K <= 2048 (a row group is held in registers)");
  YT8M_REQUIRE(q && ldq % 4 == 0 && ((uintptr_t)q & 3) == 0, YT8M_E_SHAPE, "uint8 rows must be 4-byte aligned with a row stride % 4 == 0");
  return YT8M_OK;
}

}  // namespace

extern "C" int yt8m_u8_frame_scales(const uint8_t* q, const int32_t* num_frames, int64_t B, int64_t F, int64_t D, float eps, float* rs,
                                    yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && D >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F == 0) return YT8M_OK;
  YT8M_REQUIRE(D >= 4 && D % 4 == 0 && B * F < (int64_t)4 * 0x7fffffff, YT8M_E_SHAPE, "D must be a multiple of 4");
  YT8M_REQUIRE(q && rs && ((uintptr_t)q & 3) == 0, YT8M_E_BADARG, "null / unaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  hipLaunchKernelGGL(u8_frame_scales_kernel, dim3((unsigned)((B * F + 3) / 4)), dim3(256), 0, s, q, num_frames, rs, B * F, (int)F, (int)D, eps);
  return launch_status("u8_frame_scales_kernel");
}

// y[M,N] (+)= rs (.) (a0 q . W + c0 colsum_w) (+ bias); colsum_w [N] = column sums of W[0:K] (yt8m_colsum_f32), rs may be NULL (= 1)
extern "C" int yt8m_skinny_fwd_u8(const uint8_t* q, int64_t ldq, const float* W, int64_t ldw, const float* bias, const float* rs,
                                  const float* colsum_w, float* y, int64_t ldy, int64_t M, int64_t K, int64_t N, float beta,
                                  yt8m_stream_t stream) {
  int rc = skinny_check(M, K, N, beta);
  if (rc != YT8M_OK) return rc;
  if (M == 0 || N == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_skinny_supported(M, K, N), YT8M_E_SHAPE, "K too large for the LDS-resident weight image");
  YT8M_REQUIRE(W && y && colsum_w, YT8M_E_BADARG, "null operand");
  if ((rc = u8_check(q, ldq, K)) != YT8M_OK) return rc;
  YT8M_REQUIRE(ldq >= K && ldw >= N && ldy >= N, YT8M_E_SHAPE, "bad leading dimension");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const int J = (int)((K + 255) / 256);
  int rows_per_wg = (int)((M + 1023) / 1024);
  rows_per_wg = (rows_per_wg + 15) / 16 * 16;
  const unsigned grid = (unsigned)((M + rows_per_wg - 1) / rows_per_wg);
  if (N <= 8) {
    static DeviceOnce once8, once85;
    if (K <= 1280) {
      YT8M_HIP_CHECK(once85.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8, uint8_t, 5>), 144 * 1024));
      hipLaunchKernelGGL((skinny_fwd_kernel<8, uint8_t, 5>), dim3(grid), dim3(256), (size_t)J * 8 * 64 * 4 * sizeof(float), s, q, ldq, W, ldw,
                         (int64_t)1, bias, y, ldy, M, (int)K, (int)N, beta, rows_per_wg, (int64_t)0, (int64_t)0, (int64_t)0, rs, (int64_t)0,
                         colsum_w, (int64_t)0);
    } else {
      YT8M_HIP_CHECK(once8.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8, uint8_t>), 144 * 1024));
      hipLaunchKernelGGL((skinny_fwd_kernel<8, uint8_t>), dim3(grid), dim3(256), (size_t)J * 8 * 64 * 4 * sizeof(float), s, q, ldq, W, ldw,
                         (int64_t)1, bias, y, ldy, M, (int)K, (int)N, beta, rows_per_wg, (int64_t)0, (int64_t)0, (int64_t)0, rs, (int64_t)0,
                         colsum_w, (int64_t)0);
    }
  } else {
    static DeviceOnce once16;
    YT8M_HIP_CHECK(once16.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<16, uint8_t>), 144 * 1024));
    hipLaunchKernelGGL((skinny_fwd_kernel<16, uint8_t>), dim3(grid), dim3(256), (size_t)J * 16 * 64 * 4 * sizeof(float), s, q, ldq, W, ldw,
                       (int64_t)1, bias, y, ldy, M, (int)K, (int)N, beta, rows_per_wg, (int64_t)0, (int64_t)0, (int64_t)0, rs, (int64_t)0,
                       colsum_w, (int64_t)0);
  }
  return launch_status("skinny_fwd_kernel<u8>");
}

// dW[K,N] (+)= a0 q^T . (rs (.) dy) + c0 1 (x) colsum(rs (.) dy)   (workspace as yt8m_skinny_dw_f32)
extern "C" int yt8m_skinny_dw_u8(const uint8_t* q, int64_t ldq, const float* dy, int64_t ldy, const float* rs, float* dW, int64_t lddw,
                                 int64_t M, int64_t K, int64_t N, float beta, void* workspace, int64_t workspace_bytes,
                                 yt8m_stream_t stream) {
  int rc = skinny_check(M, K, N, beta);
  if (rc != YT8M_OK) return rc;
  if (K == 0 || N == 0) return YT8M_OK;
  YT8M_REQUIRE(dW && (M == 0 || dy), YT8M_E_BADARG, "null operand");
  if (M > 0 && (rc = u8_check(q, ldq, K)) != YT8M_OK) return rc;
  YT8M_REQUIRE(ldq >= K && ldy >= N && lddw >= N, YT8M_E_SHAPE, "bad leading dimension");
  YT8M_REQUIRE(workspace && workspace_bytes >= yt8m_skinny_workspace_bytes(M, K, N), YT8M_E_BADARG, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const BwdPlan pl = bwd_plan(M, K);
  float* ws = static_cast<float*>(workspace);
  const dim3 grid((unsigned)pl.nslices, (unsigned)(pl.chunks > 0 ? pl.chunks : 1));
  if (N <= 8) {
    if (pl.chunks > 0)
      hipLaunchKernelGGL((skinny_bwd_kernel<8, false, false, uint8_t>), grid, dim3(256), 0, s, q, ldq, dy, ldy, (const float*)nullptr,
                         (int64_t)0, (int64_t)0, ws, (int64_t)0, M, (int)K, (int)N, 0.f, pl.rows, pl.kslice, 0, (int64_t)0, (int64_t)0,
                         (int64_t)0, (int64_t)0, rs, (int64_t)0);
    hipLaunchKernelGGL(skinny_dw_reduce_kernel<8>, dim3((unsigned)((K * 8 + 31) / 32)), dim3(256), 0, s, ws, pl.chunks, dW, lddw, (int)K,
                       (int)N, beta);
  } else {
    if (pl.chunks > 0)
      hipLaunchKernelGGL((skinny_bwd_kernel<16, false, false, uint8_t>), grid, dim3(256), 0, s, q, ldq, dy, ldy, (const float*)nullptr,
                         (int64_t)0, (int64_t)0, ws, (int64_t)0, M, (int)K, (int)N, 0.f, pl.rows, pl.kslice, 0, (int64_t)0, (int64_t)0,
                         (int64_t)0, (int64_t)0, rs, (int64_t)0);
    hipLaunchKernelGGL(skinny_dw_reduce_kernel<16>, dim3((unsigned)((K * 16 + 31) / 32)), dim3(256), 0, s, ws, pl.chunks, dW, lddw,
                       (int)K, (int)N, beta);
  }
  return launch_status("skinny_bwd_kernel<dW, u8>");
}

// attention pooling of the raw frames (W/all_frame_models/lstm_attention_max_pooling_model.py:63 on x above):
//   fwd: C[b] = (w[b] (.) rs[b])^T (a0 q[b] + c0)      q [B,F,H] uint8, w [B,F,A], rs [B,F], C [B,A,H]
//   dw : dw[b,f,a] = rs[b,f] (a0 q[b,f,:] . dC[b,a,:] + c0 dCsum[b,a]),  dCsum[b,a] = sum_h dC[b,a,h]
extern "C" int yt8m_attn_pool_fwd_u8(const float* w, const uint8_t* q, const float* rs, float* C, int64_t B, int64_t F, int64_t A,
                                     int64_t H, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * A * H == 0) return YT8M_OK;
  YT8M_REQUIRE(yt8m_attn_pool_supported(B, F, A, H), YT8M_E_SHAPE, "attention pooling needs A <= 16, H % 4 == 0");
  YT8M_REQUIRE(w && q && C && ((uintptr_t)C & 15) == 0 && ((uintptr_t)q & 3) == 0, YT8M_E_BADARG, "null / unaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const int nsl = (int)((H + 1023) / 1024);
  const int kslice = (int)(((H + nsl - 1) / nsl + 3) / 4 * 4);
  const dim3 grid((unsigned)nsl, 1, (unsigned)B);
  const int rows = (int)((F + 63) / 64 * 64);
  if (A <= 8)
    hipLaunchKernelGGL((skinny_bwd_kernel<8, false, true, uint8_t>), grid, dim3(256), 0, s, q, H, w, A, (const float*)nullptr, (int64_t)0,
                       (int64_t)0, C, H, F, (int)H, (int)A, 0.f, rows, kslice, 1, F * H, F * A, (int64_t)0, A * H, rs, F);
  else
    hipLaunchKernelGGL((skinny_bwd_kernel<16, false, true, uint8_t>), grid, dim3(256), 0, s, q, H, w, A, (const float*)nullptr, (int64_t)0,
                       (int64_t)0, C, H, F, (int)H, (int)A, 0.f, rows, kslice, 1, F * H, F * A, (int64_t)0, A * H, rs, F);
  return launch_status("skinny_bwd_kernel<pool, u8>");
}

extern "C" int yt8m_attn_pool_dw_u8(const uint8_t* q, const float* rs, const float* dC, const float* dCsum, float* dw, int64_t B,
                                    int64_t F, int64_t A, int64_t H, yt8m_stream_t stream) {
  YT8M_REQUIRE(B >= 0 && F >= 0 && A >= 0 && H >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * F * A == 0) return YT8M_OK;
  YT8M_REQUIRE(H > 0 && H <= 2048 && yt8m_attn_pool_supported(B, F, A, H), YT8M_E_SHAPE, "attention pooling needs A <= 16, H % 4 == 0, H <= 2048");
  YT8M_REQUIRE(q && dC && dCsum && dw && ((uintptr_t)q & 3) == 0, YT8M_E_BADARG, "null / unaligned operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  const int J = (int)((H + 255) / 256);
  int rows_per_wg = (int)((F + 3) / 4);
  rows_per_wg = (rows_per_wg + 15) / 16 * 16;
  const dim3 grid((unsigned)((F + rows_per_wg - 1) / rows_per_wg), (unsigned)B);
  if (A <= 8) {
    static DeviceOnce once, once5;
    if (H <= 1280) {
      YT8M_HIP_CHECK(once5.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8, uint8_t, 5>), 144 * 1024));
      hipLaunchKernelGGL((skinny_fwd_kernel<8, uint8_t, 5>), grid, dim3(256), (size_t)J * 8 * 64 * 4 * sizeof(float), s, q, H, dC, (int64_t)1, H,
                         (const float*)nullptr, dw, A, F, (int)H, (int)A, 0.f, rows_per_wg, F * H, A * H, F * A, rs, F, dCsum, A);
    } else {
      YT8M_HIP_CHECK(once.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<8, uint8_t>), 144 * 1024));
      hipLaunchKernelGGL((skinny_fwd_kernel<8, uint8_t>), grid, dim3(256), (size_t)J * 8 * 64 * 4 * sizeof(float), s, q, H, dC, (int64_t)1, H,
                         (const float*)nullptr, dw, A, F, (int)H, (int)A, 0.f, rows_per_wg, F * H, A * H, F * A, rs, F, dCsum, A);
    }
  } else {
    static DeviceOnce once;
    YT8M_HIP_CHECK(once.lds(reinterpret_cast<const void*>(skinny_fwd_kernel<16, uint8_t>), 144 * 1024));
    hipLaunchKernelGGL((skinny_fwd_kernel<16, uint8_t>), grid, dim3(256), (size_t)J * 16 * 64 * 4 * sizeof(float), s, q, H, dC, (int64_t)1, H,
                       (const float*)nullptr, dw, A, F, (int)H, (int)A, 0.f, rows_per_wg, F * H, A * H, F * A, rs, F, dCsum, A);
  }
  return launch_status("skinny_fwd_kernel<pool dw, u8>");
}
