// gemm_f32.hip -- exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, gfx950).
//
// C[M,N] = op(A)[M,K] . op(B)[K,N] (+ bias[N]) (+ C)        row-major, fp32 in / fp32 accumulate.
//
// Why f32-input MFMA: BASELINE config[1] is quoted in fp32 and the parity bar is 1e-3 on
// probabilities; the f32 MFMA is bit-for-bit an fmaf chain (MI355X_MICROARCH.md "Matrix cores"), runs
// at the 157 TFLOP/s fp32 peak and leaves the VALU free for the epilogue.
//
// Tiling (wave64, 4 waves / workgroup):
//   workgroup tile 128 x 128, K-step 16, waves arranged 2 (M) x 2 (N), each wave owns 64 x 64 =
//   2 x 2 MFMA tiles of 32 x 32 (4 accumulators x 16 VGPR).  Per K-step a wave issues 32 MFMAs
//   (2048 matrix-pipe cycles) against 32 ds_read_b32 -- the kernel is matrix-pipe bound by design.
//   LDS holds both operands K-major ([k][m] / [k][n]) so that an MFMA operand fetch (lane l reads
//   element (k = l>>5, i = l&31)) is 32 consecutive floats per half-wave: conflict-free.
//   Operands that are K-contiguous in HBM (A of x.W, B of dZ.W^T) are read as float4 along K
//   (16 rows x 64 B per wave instruction) and transposed on the LDS write; row stride 130 floats makes
//   the 4 x ds_write_b32 conflict-free (4*130 mod 32 = 8).  Operands that are M/N-contiguous are read
//   as float4 along N and written with ds_write_b128 (row stride 132 floats, 16-B aligned).
//   Two LDS buffers + register prefetch of the next K-step: one barrier per K-step.
//   XCD-aware rasterisation: consecutive workgroup ids land on different XCDs (id % 8), so the grid is
//   remapped such that each XCD walks a contiguous strip of N-tiles and re-uses its B panel from its
//   own 4 MiB L2 across the M-tiles.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_KC = 130;  // row stride (floats) of a tile whose source is K-contiguous
constexpr int LDS_XC = 132;  // row stride (floats) of a tile whose source is M/N-contiguous
constexpr int TILE_FLOATS = BK * LDS_XC;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n;
  int vecA, vecB;
  int accumulate;
  int64_t strideA, strideB, strideC;  // batched form: blockIdx.y selects the problem
};

// global -> registers for one [BK x 128] operand tile.  KC: element (x, k) at P[x*ld + k]; else P[k*ld + x].
template <bool KC>
__device__ __forceinline__ void gload(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, int K,
                                      bool vec, int tid, float4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    if (KC) {
      const int gx = x0 + (idx >> 2), gk = k0 + (idx & 3) * 4;
      const float* p = P + (int64_t)gx * ld + gk;
      if (vec && gx < X && gk + 3 < K) {
        r[i] = *reinterpret_cast<const float4*>(p);
      } else {
        const bool okx = gx < X;
        r[i].x = (okx && gk + 0 < K) ? p[0] : 0.f;
        r[i].y = (okx && gk + 1 < K) ? p[1] : 0.f;
        r[i].z = (okx && gk + 2 < K) ? p[2] : 0.f;
        r[i].w = (okx && gk + 3 < K) ? p[3] : 0.f;
      }
    } else {
      const int gk = k0 + (idx >> 5), gx = x0 + (idx & 31) * 4;
      const float* p = P + (int64_t)gk * ld + gx;
      if (vec && gk < K && gx + 3 < X) {
        r[i] = *reinterpret_cast<const float4*>(p);
      } else {
        const bool okk = gk < K;
        r[i].x = (okk && gx + 0 < X) ? p[0] : 0.f;
        r[i].y = (okk && gx + 1 < X) ? p[1] : 0.f;
        r[i].z = (okk && gx + 2 < X) ? p[2] : 0.f;
        r[i].w = (okk && gx + 3 < X) ? p[3] : 0.f;
      }
    }
  }
}

template <bool KC>
__device__ __forceinline__ void sstore(float* __restrict__ S, int tid, const float4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    if (KC) {
      const int x = idx >> 2, kq = (idx & 3) * 4;
      S[(kq + 0) * LDS_KC + x] = r[i].x;
      S[(kq + 1) * LDS_KC + x] = r[i].y;
      S[(kq + 2) * LDS_KC + x] = r[i].z;
      S[(kq + 3) * LDS_KC + x] = r[i].w;
    } else {
      const int k = idx >> 5, xq = (idx & 31) * 4;
      *reinterpret_cast<float4*>(&S[k * LDS_XC + xq]) = r[i];
    }
  }
}

// A_KC: A stored [M,K] (transA = 0).  B_KC: B stored [N,K] (transB = 1).
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[4 * TILE_FLOATS];
  float* const As = smem;                     // As + buf * TILE_FLOATS
  float* const Bs = smem + 2 * TILE_FLOATS;   // Bs + buf * TILE_FLOATS
  constexpr int SA = A_KC ? LDS_KC : LDS_XC;
  constexpr int SB = B_KC ? LDS_KC : LDS_XC;

  // XCD-aware rasterisation (bijective for any grid size): wg -> (xcd, slot) -> linear tile id where each
  // XCD owns a contiguous range; within the range M-tiles are fastest so neighbours share the B panel.
  const int nwg = g.tiles_m * g.tiles_n;
  const int wg = blockIdx.x;
  const int xcd = wg & 7, slot = wg >> 3;
  const int q = nwg >> 3, rem = nwg & 7;
  const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
  const int tm = tile % g.tiles_m, tn = tile / g.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const float* __restrict__ Ap = g.A + (int64_t)blockIdx.y * g.strideA;
  const float* __restrict__ Bp = g.B + (int64_t)blockIdx.y * g.strideB;
  float* __restrict__ Cp = g.C + (int64_t)blockIdx.y * g.strideC;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2], rb[2];
  const int nk = (g.K + BK - 1) / BK;
  gload<A_KC>(Ap, g.lda, m0, 0, g.M, g.K, g.vecA, tid, ra);
  gload<B_KC>(Bp, g.ldb, n0, 0, g.N, g.K, g.vecB, tid, rb);
  sstore<A_KC>(As, tid, ra);
  sstore<B_KC>(Bs, tid, rb);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      gload<A_KC>(Ap, g.lda, m0, (kt + 1) * BK, g.M, g.K, g.vecA, tid, ra);
      gload<B_KC>(Bp, g.ldb, n0, (kt + 1) * BK, g.N, g.K, g.vecB, tid, rb);
    }
    const float* as = As + cur * TILE_FLOATS + lk * SA + wm + li;
    const float* bs = Bs + cur * TILE_FLOATS + lk * SB + wn + li;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = as[kk * SA], a1 = as[kk * SA + 32];
      const float b0 = bs[kk * SB], b1 = bs[kk * SB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      sstore<A_KC>(As + (cur ^ 1) * TILE_FLOATS, tid, ra);
      sstore<B_KC>(Bs + (cur ^ 1) * TILE_FLOATS, tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + j * 32 + li;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < g.M) {
          float* c = Cp + (int64_t)row * g.ldc + col;
          float v = acc[i][j][r] + bv;
          if (g.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

static int gemm_launch(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                       int64_t strideC, const float* bias, float beta, int64_t batch, yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  if (M == 0 || N == 0 || batch == 0) return YT8M_OK;
  YT8M_REQUIRE(A && B && C, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), YT8M_E_SHAPE, "dimension >= 2^31");
  YT8M_REQUIRE(batch <= 65535, YT8M_E_SHAPE, "batch > 65535");
  YT8M_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, YT8M_E_SHAPE, "leading dimension too small");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.tiles_m = (int)((M + BM - 1) / BM);
  g.tiles_n = (int)((N + BN - 1) / BN);
  g.vecA = (lda % 4 == 0) && aligned16(A) && (strideA % 4 == 0);
  g.vecB = (ldb % 4 == 0) && aligned16(B) && (strideB % 4 == 0);
  g.accumulate = beta != 0.f;
  const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
  YT8M_REQUIRE(nwg < (1LL << 31), YT8M_E_SHAPE, "grid too large");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s);
  dim3 grid((unsigned)nwg, (unsigned)batch), block(256);
  if (!transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g);
  else if (transA && !transB) hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g);
  else if (!transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, g);
  return launch_status("gemm_f32_kernel");
}

extern "C" int yt8m_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float beta,
                             yt8m_stream_t stream) {
  return gemm_launch(transA, transB, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, bias, beta, 1, stream);
}

extern "C" int yt8m_gemm_f32_batched(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                                     int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                                     int64_t strideC, float beta, int64_t batch, yt8m_stream_t stream) {
  return gemm_launch(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, nullptr, beta, batch, stream);
}
