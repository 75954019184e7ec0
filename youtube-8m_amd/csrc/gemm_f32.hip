// gemm_f32.hip -- exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, gfx950).
//
// C[M,N] = op(A)[M,K] . op(B)[K,N] (+ bias[N]) (+ C)        row-major, fp32 in / fp32 accumulate.
//
// Why f32-input MFMA: BASELINE config[1] is quoted in fp32 and the parity bar is 1e-3 on
// probabilities; the f32 MFMA is bit-for-bit an fmaf chain (MI355X_MICROARCH.md "Matrix cores"), runs
// at the 157 TFLOP/s fp32 peak and leaves the VALU free for the epilogue.
//
// Tiling (wave64, 4 waves / workgroup):
//   workgroup tile 128 x 128, K-step 16, waves arranged 2 (M) x 2 (N), each wave owns 64 x 64 = 2 x 2 MFMA tiles of
//   32 x 32 (4 accumulators x 16 AGPR).  Per K-step a wave issues 32 MFMAs = 2048 matrix-pipe cycles.
//   Operand tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass, issued a
//   whole K-step ahead, two LDS buffers, ONE barrier per K-step).  The LDS image is lane-linear as the DMA requires;
//   K-contiguous operands are XOR-swizzled through the SOURCE address so their ds_read_b128 fragment fetch is
//   conflict-free, N-contiguous operands are read with conflict-free ds_read_b32 (32 consecutive floats / half-wave).
//   Edge tiles, unaligned leading dimensions and K % 16 != 0 take a guarded register-staged path with the same image.
//   XCD-aware rasterisation: consecutive workgroup ids land on different XCDs (id % 8), so the grid is remapped such
//   that each XCD walks a contiguous strip of tiles and re-uses its B panel from its own 4 MiB L2 across M-tiles.
#include "common.h"

#ifdef YT8M_GEMM_TIMING
__device__ unsigned long long yt8m_gemm_phase[8];
extern "C" int yt8m_debug_gemm_phase(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(yt8m_gemm_phase), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -3;
}
#endif

// Experiment switch (tools/build_variant.sh -DYT8M_GEMM_TEPI=1): accumulate C^T sub-tiles (operands swapped in the MFMA) so that
// a lane owns 4 CONSECUTIVE columns of one C row per register quad and stores 16 bytes straight from the accumulators (16
// instead of 64 store instructions per wave, no LDS pass).  Bit-identical results; measured SLOWER: cfg[1] step 1.240 -> 1.266
// ms, M = 8192 heads 123 -> 119 TFLOP/s -- a wave store then touches 32 rows x 32 bytes instead of 2 rows x 128 bytes, and the
// write path prefers whole 128-byte segments over fewer instructions.  Off.
#ifndef YT8M_GEMM_TEPI
#define YT8M_GEMM_TEPI 0
#endif

namespace {

constexpr bool TEPI = YT8M_GEMM_TEPI != 0;
constexpr int BM = 128, BN = 128, BK = 16;
constexpr int TILE_FLOATS = BK * 128;  // one operand tile in LDS (8 KiB), no padding
constexpr int STAGES = 3;               // LDS-DMA pipeline depth (3 x 16 KiB per workgroup, 3 workgroups per CU = 144 KiB)

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

// ---- LDS image of a [BK x 128] operand tile (same image whichever way it is filled) --------------------------
// Thread slot idx in [0,512) owns the 16 bytes at float offset idx*4 (lane-linear, which is what the LDS-DMA
// form of global_load requires: destination = wave-uniform base + lane*16).
//   KC source (operand rows are K-contiguous in HBM, e.g. x[B,D] of x.W): slot -> row x = idx>>2, chunk slot
//     cs = idx&3; the slot holds the row's k-chunk c = cs ^ ((x>>2)&3)  (XOR swizzle applied on the SOURCE
//     address, so the 16-lane groups of the ds_read_b128 fragment fetch hit 16 distinct 16-B bank slots).
//   XC source (operand is M/N-contiguous, e.g. W[D,N]): slot -> k = idx>>5, x = (idx&31)*4; image = [k][128].
template <bool KC>
__device__ __forceinline__ const float* slot_src(const float* __restrict__ P, int64_t ld, int x0, int k0, int idx,
                                                  int& gx, int& gk) {
  if (KC) {
    const int x = idx >> 2;
    gx = x0 + x;
    gk = k0 + 4 * ((idx & 3) ^ ((x >> 2) & 3));
    return P + (int64_t)gx * ld + gk;
  } else {
    gk = k0 + (idx >> 5);
    gx = x0 + (idx & 31) * 4;
    return P + (int64_t)gk * ld + gx;
  }
}

// asynchronous fill of one tile by LDS-DMA (16-byte aligned operands, K % 16 == 0).  Rows / columns of an EDGE tile that
// lie beyond the matrix (gx >= X) are redirected to the last valid row / 16-byte column chunk: they only ever feed
// output rows / columns >= M / N, which the epilogue never stores, so edge tiles run at full DMA speed too.
template <bool KC>
__device__ __forceinline__ void fill_dma(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, float* S, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    int gx, gk;
    const float* src = slot_src<KC>(P, ld, x0, k0, idx, gx, gk);
    if (gx >= X) src -= KC ? (int64_t)(gx - (X - 1)) * ld : (int64_t)(gx - ((X - 1) & ~3));
    float* dst = S + (idx & ~63) * 4;  // wave-uniform base; the hardware adds lane*16 bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

// guarded register fill (edge tiles, unaligned leading dimensions, K tails): same image, zero padded
template <bool KC>
__device__ __forceinline__ float4 load_guarded(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, int K,
                                               bool vec, int idx) {
  int gx, gk;
  const float* p = slot_src<KC>(P, ld, x0, k0, idx, gx, gk);
  float4 r;
  if (KC) {
    if (vec && gx < X && gk + 3 < K) {
      r = *reinterpret_cast<const float4*>(p);
    } else {
      const bool ok = gx < X;
      r.x = (ok && gk + 0 < K) ? p[0] : 0.f;
      r.y = (ok && gk + 1 < K) ? p[1] : 0.f;
      r.z = (ok && gk + 2 < K) ? p[2] : 0.f;
      r.w = (ok && gk + 3 < K) ? p[3] : 0.f;
    }
  } else {
    if (vec && gk < K && gx + 3 < X) {
      r = *reinterpret_cast<const float4*>(p);
    } else {
      const bool ok = gk < K;
      r.x = (ok && gx + 0 < X) ? p[0] : 0.f;
      r.y = (ok && gx + 1 < X) ? p[1] : 0.f;
      r.z = (ok && gx + 2 < X) ? p[2] : 0.f;
      r.w = (ok && gx + 3 < X) ? p[3] : 0.f;
    }
  }
  return r;
}

// fill of K-step kt into a stage: LDS-DMA when the step lies wholly inside K, guarded zero-padded store for the K tail
// (K % 16 != 0: e.g. the 300-frame reduction of the NetVLAD / attention aggregation, the 4716-wide chain FC)
template <bool KC>
__device__ __forceinline__ void fill_step(const float* __restrict__ P, int64_t ld, int x0, int kt, int X, int K, float* S,
                                          int tid) {
  if ((kt + 1) * BK <= K) {
    fill_dma<KC>(P, ld, x0, kt * BK, X, S, tid);
  } else {
    const float4 r0 = load_guarded<KC>(P, ld, x0, kt * BK, X, K, true, tid);
    const float4 r1 = load_guarded<KC>(P, ld, x0, kt * BK, X, K, true, tid + 256);
    *reinterpret_cast<float4*>(&S[tid * 4]) = r0;
    *reinterpret_cast<float4*>(&S[(tid + 256) * 4]) = r1;
  }
}

// ---- MFMA operand fragments -------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 takes, per lane, ONE A value A[i = lane&31][k = lane>>5] and one B value.  Which two k of
// the K-step an MFMA reduces is free as long as A and B agree, so step (h, j), h in {0,1}, j in {0..3}, uses
//   k = 8h + j (lanes 0-31)  and  k = 8h + 4 + j (lanes 32-63):
// a KC operand then needs ONE ds_read_b128 per (tile t, h): lane (li, lk) fetches chunk c = 2h + lk of its row
// and register component j feeds step (h, j); an XC operand reads row k = 8h + 4*lk + j with ds_read_b32.
struct Frag {
  float v[2][2][4];  // [t: 32-row sub-tile][h][j]
};

template <bool KC>
__device__ __forceinline__ void load_frag(const float* __restrict__ S, int xo, int li, int lk, Frag& f) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (KC) {
        const int row = xo + t * 32 + li;
        const float4 q = *reinterpret_cast<const float4*>(&S[row * 16 + 4 * ((2 * h + lk) ^ ((row >> 2) & 3))]);
        f.v[t][h][0] = q.x; f.v[t][h][1] = q.y; f.v[t][h][2] = q.z; f.v[t][h][3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.v[t][h][j] = S[(8 * h + 4 * lk + j) * 128 + xo + t * 32 + li];
      }
    }
  }
}

__device__ __forceinline__ void mma_step(const Frag& fa, const Frag& fb, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (TEPI) {   // D = B^T-operand x A-operand: the same products in the same order, the tile lands transposed
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.v[0][h][j], fa.v[0][h][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.v[1][h][j], fa.v[0][h][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.v[0][h][j], fa.v[1][h][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb.v[1][h][j], fa.v[1][h][j], acc[1][1], 0, 0, 0);
      } else {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v[0][h][j], fb.v[0][h][j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v[0][h][j], fb.v[1][h][j], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v[1][h][j], fb.v[0][h][j], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.v[1][h][j], fb.v[1][h][j], acc[1][1], 0, 0, 0);
      }
    }
  }
}

// bf16 operands (K-contiguous on both sides, "NT"): a bf16 matrix [M,K] IS, byte for byte, a float matrix [M,K/2], so the
// fill / swizzle / pipeline code above is reused unchanged with K and the leading dimensions counted in floats; only the
// MMA differs: the 16-byte chunk c = 2h + lk that load_frag<KC> fetches holds the bf16 elements k = 16h + 8*lk .. +7 of
// the 32-wide bf16 K-step, which is exactly the per-lane operand of v_mfma_f32_32x32x16_bf16 number h.  2 MFMAs per
// sub-tile pair and K-step (8 per wave) instead of 32: this path is bound by operand delivery (LDS-DMA / LDS reads).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mma_step_bf16(const Frag& fa, const Frag& fb, f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    bf16x8 a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 qa = {fa.v[t][h][0], fa.v[t][h][1], fa.v[t][h][2], fa.v[t][h][3]};
      const float4 qb = {fb.v[t][h][0], fb.v[t][h][1], fb.v[t][h][2], fb.v[t][h][3]};
      a[t] = __builtin_bit_cast(bf16x8, qa);
      b[t] = __builtin_bit_cast(bf16x8, qb);
    }
    if (TEPI) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[0], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], a[1], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1], a[1], acc[1][1], 0, 0, 0);
    } else {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[1][1], 0, 0, 0);
    }
  }
}

// One pass over K for a workgroup tile.
//  DMA path    : 3 LDS stages.  K-step kt+2 is issued as LDS-DMA at the TOP of iteration kt (side-effecting => it stays
//                there) and has two MFMA blocks (2 x 2048 matrix-pipe cycles) to land; the wait before the barrier is a
//                COUNTED vmcnt(4) -- each lane has exactly 4 DMA instructions per K-step in flight order -- so only
//                K-step kt+1 is waited for, kt+2 stays on the wire across the barrier (raw s_barrier: __syncthreads()
//                would drain vmcnt(0)).  A stage is re-filled two barriers after its last ds_read.
//  guarded path: register staging, 2 stages; used for unaligned operands / K % 16 != 0.
// In both, every fragment of the K-step is fetched from LDS before the first MFMA (sched_barrier pins the order) so
// the LDS latency is paid once per K-step and the MFMAs stream.
template <bool A_KC, bool B_KC, bool DMA, bool BF16>
__device__ __forceinline__ void mainloop(const GemmArgs& g, const float* __restrict__ Ap, const float* __restrict__ Bp,
                                         float* __restrict__ As, float* __restrict__ Bs, int m0, int n0, int kb, int ke,
                                         int tid, int wm, int wn, int li, int lk, f32x16 (&acc)[2][2]) {
  // K-steps [kb, ke) of this tile (a split-K part of a remainder tile processes a sub-range)
  if (DMA) {
    fill_step<A_KC>(Ap, g.lda, m0, kb, g.M, g.K, As, tid);
    fill_step<B_KC>(Bp, g.ldb, n0, kb, g.N, g.K, Bs, tid);
    if (kb + 1 < ke) {
      fill_step<A_KC>(Ap, g.lda, m0, kb + 1, g.M, g.K, As + TILE_FLOATS, tid);
      fill_step<B_KC>(Bp, g.ldb, n0, kb + 1, g.N, g.K, Bs + TILE_FLOATS, tid);
    }
    // the K-tail step (if any) is the LAST one: whenever it has been stored above there is nothing younger in flight, so
    // a full drain is exact; otherwise leave K-step kb+1's 4 DMAs on the wire
    if (kb + 2 < ke || (kb + 1 < ke && (ke * BK <= g.K))) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
#ifdef YT8M_GEMM_TIMING   // (timing experiments only) per-phase cycle sums of one wave: DMA issue, LDS fragment reads, MFMA
    uint64_t ph[5] = {0, 0, 0, 0, 0};   // block, vmcnt wait, barrier
#define PH_T(x) const uint64_t x = __builtin_amdgcn_s_memtime()
#else
#define PH_T(x)
#endif
    for (int kt = kb; kt < ke; ++kt) {
      const int nxt2 = cur >= 1 ? cur - 1 : 2;           // (cur + 2) % 3
      PH_T(p0);
      if (kt + 2 < ke) {
        fill_step<A_KC>(Ap, g.lda, m0, kt + 2, g.M, g.K, As + nxt2 * TILE_FLOATS, tid);
        fill_step<B_KC>(Bp, g.ldb, n0, kt + 2, g.N, g.K, Bs + nxt2 * TILE_FLOATS, tid);
      }
      PH_T(p1);
      Frag fa, fb;
      load_frag<A_KC>(As + cur * TILE_FLOATS, wm, li, lk, fa);
      load_frag<B_KC>(Bs + cur * TILE_FLOATS, wn, li, lk, fb);
      __builtin_amdgcn_sched_barrier(0);
#ifdef YT8M_GEMM_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      PH_T(p2);
      if (BF16) mma_step_bf16(fa, fb, acc);
      else mma_step(fa, fb, acc);
      __builtin_amdgcn_sched_barrier(0);
      PH_T(p3);
      // K-step kt+1 must have landed; kt+2 may stay in flight -- unless kt+2 was the guarded K-tail store (then it is
      // not a DMA: nothing younger than kt+1 is in flight, drain everything incl. the ds_writes)
      if (kt + 2 < ke && (kt + 3) * BK <= g.K) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      PH_T(p4);
      __builtin_amdgcn_s_barrier();
#ifdef YT8M_GEMM_TIMING
      PH_T(p5);
      ph[0] += p1 - p0; ph[1] += p2 - p1; ph[2] += p3 - p2; ph[3] += p4 - p3; ph[4] += p5 - p4;
#endif
      cur = cur == 2 ? 0 : cur + 1;
    }
#ifdef YT8M_GEMM_TIMING
    if (tid == 0 && blockIdx.x == 300)
      for (int i = 0; i < 5; ++i) yt8m_gemm_phase[i] = ph[i];
    if (tid == 0 && blockIdx.x == 300) yt8m_gemm_phase[5] = (uint64_t)(ke - kb);
#endif
    return;
  }
  float4 ra0, ra1, rb0, rb1;
  ra0 = load_guarded<A_KC>(Ap, g.lda, m0, kb * BK, g.M, g.K, g.vecA, tid);
  ra1 = load_guarded<A_KC>(Ap, g.lda, m0, kb * BK, g.M, g.K, g.vecA, tid + 256);
  rb0 = load_guarded<B_KC>(Bp, g.ldb, n0, kb * BK, g.N, g.K, g.vecB, tid);
  rb1 = load_guarded<B_KC>(Bp, g.ldb, n0, kb * BK, g.N, g.K, g.vecB, tid + 256);
  *reinterpret_cast<float4*>(&As[tid * 4]) = ra0;
  *reinterpret_cast<float4*>(&As[(tid + 256) * 4]) = ra1;
  *reinterpret_cast<float4*>(&Bs[tid * 4]) = rb0;
  *reinterpret_cast<float4*>(&Bs[(tid + 256) * 4]) = rb1;
  __syncthreads();
  int cur = 0;
  for (int kt = kb; kt < ke; ++kt) {
    const bool more = kt + 1 < ke;
    float* An = As + (cur ^ 1) * TILE_FLOATS;
    float* Bn = Bs + (cur ^ 1) * TILE_FLOATS;
    const int kn = (more ? kt + 1 : kt) * BK;
    ra0 = load_guarded<A_KC>(Ap, g.lda, m0, kn, g.M, g.K, g.vecA, tid);
    ra1 = load_guarded<A_KC>(Ap, g.lda, m0, kn, g.M, g.K, g.vecA, tid + 256);
    rb0 = load_guarded<B_KC>(Bp, g.ldb, n0, kn, g.N, g.K, g.vecB, tid);
    rb1 = load_guarded<B_KC>(Bp, g.ldb, n0, kn, g.N, g.K, g.vecB, tid + 256);
    Frag fa, fb;
    load_frag<A_KC>(As + cur * TILE_FLOATS, wm, li, lk, fa);
    load_frag<B_KC>(Bs + cur * TILE_FLOATS, wn, li, lk, fb);
    __builtin_amdgcn_sched_barrier(0);
    if (BF16) mma_step_bf16(fa, fb, acc);
    else mma_step(fa, fb, acc);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<float4*>(&An[tid * 4]) = ra0;
    *reinterpret_cast<float4*>(&An[(tid + 256) * 4]) = ra1;
    *reinterpret_cast<float4*>(&Bn[tid * 4]) = rb0;
    *reinterpret_cast<float4*>(&Bn[(tid + 256) * 4]) = rb1;
    __syncthreads();
    cur ^= 1;
  }
}

// ---- one workgroup tile: K-steps [kb, ke) -> accumulators -> epilogue --------------------------------------------
// mode 0: full reduction, C = acc (+ bias) (+ C).  mode 1: split-K part, raw accumulators to the workspace slot `ws`
// (tile-local [128][128] image); the fix-up kernel adds the parts in a fixed order.
template <bool A_KC, bool B_KC, bool BF16, bool VEPI = false>
__device__ __forceinline__ void process_tile(const GemmArgs& g, const float* __restrict__ Ap, const float* __restrict__ Bp,
                                             float* __restrict__ Cp, float* __restrict__ smem, int tm, int tn, int kb,
                                             int ke, float* __restrict__ ws) {
  float* const As = smem;                          // As + stage * TILE_FLOATS
  float* const Bs = smem + STAGES * TILE_FLOATS;   // Bs + stage * TILE_FLOATS
  const int m0 = tm * BM, n0 = tn * BN;
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

#ifdef YT8M_GEMM_TIMING
  const uint64_t tm0 = __builtin_amdgcn_s_memtime();
#endif
  // 16-byte aligned operands take the LDS-DMA path (edge tiles and the K tail included: see fill_dma / fill_step)
  const bool dma = g.vecA && g.vecB;
  if (dma) mainloop<A_KC, B_KC, true, BF16>(g, Ap, Bp, As, Bs, m0, n0, kb, ke, tid, wm, wn, li, lk, acc);
  else mainloop<A_KC, B_KC, false, BF16>(g, Ap, Bp, As, Bs, m0, n0, kb, ke, tid, wm, wn, li, lk, acc);

#ifdef YT8M_GEMM_TIMING   // (timing experiments only) cycles of the main loop / epilogue overwrite C[m0, n0..n0+1]
  const uint64_t tm1 = __builtin_amdgcn_s_memtime();
  struct TimingGuard {
    uint64_t a, b; float* c; int t;
    __device__ ~TimingGuard() {
      __syncthreads();
      if (t == 0 && c) { const uint64_t e = __builtin_amdgcn_s_memtime(); c[0] = (float)(b - a); c[1] = (float)(e - b); }
    }
  } tguard{tm0, tm1, ws ? nullptr : Cp + (int64_t)m0 * g.ldc + n0, tid};
#endif
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  // TEPI: acc[i][j][r] = C[m0 + wm + 32i + li][n0 + wn + 32j + (r & 3) + 8 (r >> 2) + 4 lk]: registers 4q..4q+3 of a lane are four
  // consecutive columns of one row
  if (ws) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (TEPI) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(&ws[(wm + i * 32 + li) * BN + wn + j * 32 + 8 * qd + 4 * lk]) =
                float4{acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            ws[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * BN + wn + j * 32 + li] = acc[i][j][r];
        }
      }
    return;
  }
  if (TEPI) {
    const bool vec = (g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(Cp) | reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm + i * 32 + li;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int col = n0 + wn + j * 32 + 8 * qd + 4 * lk;
          if (col >= g.N) continue;
          float4 v = {acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]};
          float* c = Cp + (int64_t)row * g.ldc + col;
          if (vec && col + 3 < g.N) {
            if (g.bias) {
              const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (g.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(c);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(c) = v;
          } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int e = 0; e < 4 && col + e < g.N; ++e) {
              float t = vv[e] + (g.bias ? g.bias[col + e] : 0.f);
              if (g.accumulate) t += c[e];
              c[e] = t;
            }
          }
        }
      }
    }
    return;
  }
  // Output path.  Default: one dword store per accumulator register (64 per wave and tile).  VEPI: the accumulators are
  // transposed through LDS (free after the main loop; each wave stages its own 32 x 64 half-tiles) and leave as 16-byte
  // stores, 16 instead of 64 store instructions per wave -- the store pipe is issue-bound, this cuts the per-tile epilogue
  // from 74k to 29k cycles (tools/gemm_timing.py).  It is a separate instantiation because it only pays where the epilogue
  // is a large share of a tile: measured +7 % on the cfg[1] dW shape (K = 1024, transA), +-0 on the forward shape and
  // -3..5 % at K >= 4096 (the epilogue of one workgroup already hides under the main loops of the two others on the CU).
  if (VEPI && (g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(Cp) | reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0) {
    constexpr int P = 68;                                   // padded row pitch (floats) of the staging image
    float* st = smem + (tid >> 6) * (32 * P);
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * lk) * P + j * 32 + li] = acc[i][j][r];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = lane + 64 * k;
        const int rr = idx >> 4, c4 = (idx & 15) * 4;
        const int row = m0 + wm + i * 32 + rr, col = n0 + wn + c4;
        float4 v = *reinterpret_cast<const float4*>(&st[rr * P + c4]);
        if (row < g.M && col < g.N) {
          float* c = Cp + (int64_t)row * g.ldc + col;
          if (col + 3 < g.N) {
            if (g.bias) {
              const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if (g.accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(c);
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4*>(c) = v;
          } else {                                          // last, partial group of columns of the matrix
            const float vv[4] = {v.x, v.y, v.z, v.w};
            for (int e = 0; e < 4 && col + e < g.N; ++e) {
              float t = vv[e] + (g.bias ? g.bias[col + e] : 0.f);
              if (g.accumulate) t += c[e];
              c[e] = t;
            }
          }
        }
      }
    }
    return;
  }
#ifndef YT8M_GEMM_PAIR_EPI
#define YT8M_GEMM_PAIR_EPI 0      // experiment (-DYT8M_GEMM_PAIR_EPI=1): +1 % on M >= 8192 shapes, -1 % on cfg[1], -3.6 % on the dx shape: off
#endif
  // Pair epilogue: neighbouring lanes (columns c, c+1) swap one register of each accumulator pair (rows R, R+1) with a DPP quad
  // permute, after which the even lane holds (c, c+1) of row R and the odd lane (c-1, c) of row R+1: 8-byte stores, 4 rows x
  // 128 contiguous bytes per wave instruction, half the store instructions of the dword form and no LDS pass.
  if (YT8M_GEMM_PAIR_EPI && (g.ldc & 1) == 0 && ((reinterpret_cast<uintptr_t>(Cp) | reinterpret_cast<uintptr_t>(g.bias)) & 7) == 0) {
    const bool odd = (li & 1) != 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 32 + (li & ~1);
      float2 bv = {0.f, 0.f};
      if (g.bias && col + 1 < g.N) bv = *reinterpret_cast<const float2*>(g.bias + col);
      else if (g.bias && col < g.N) bv.x = g.bias[col];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
          const float a0 = acc[i][j][2 * rp], a1 = acc[i][j][2 * rp + 1];
          const float send = odd ? a0 : a1;
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xf, 0xf, true));
          const int r = 2 * rp + (odd ? 1 : 0);
          const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          float2 v = {(odd ? recv : a0) + bv.x, (odd ? a1 : recv) + bv.y};
          if (row < g.M && col < g.N) {
            float* c = Cp + (int64_t)row * g.ldc + col;
            if (col + 1 < g.N) {
              if (g.accumulate) { const float2 o = *reinterpret_cast<const float2*>(c); v.x += o.x; v.y += o.y; }
              *reinterpret_cast<float2*>(c) = v;
            } else {
              if (g.accumulate) v.x += c[0];
              c[0] = v.x;
            }
          }
        }
      }
    }
    return;
  }
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

// XCD-aware bijective remap of a linear id: consecutive hardware workgroup ids land on different XCDs (id % 8), so
// give each XCD a contiguous range of logical ids (neighbouring tiles then share operand panels in ONE 4 MiB L2).
__device__ __forceinline__ int xcd_remap(int wg, int n) {
  const int xcd = wg & 7, slot = wg >> 3;
  const int q = n >> 3, rem = n & 7;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
}

// Logical tile id -> (tm, tn): bands of GM = 16 M-tiles, inside a band the M index runs fastest.  An XCD's share of a round
// (96 consecutive logical tiles) is then a 16 x 6 block that needs 16 A panels + 6 B panels instead of the 64 + 1.5 of the
// plain M-fastest order when tiles_m >> 16.  For tiles_m <= 16 (BASELINE cfg[1]: 8 and 9 tile rows) the order is the plain
// one it was tuned with (bands of 8 cost cfg[1]'s dW launch 1 %).  Measured effect on
// the tall problems (M = 8192 MoE heads, M = 38400 LSTM projections) is small (+0.6 %): they already ran at 0.76-0.79 of the
// MFMA peak out of the 256 MB Infinity Cache; kept because it cuts the beyond-L2 operand traffic per tile ~10x.
constexpr int RASTER_GM = 16;   // >= 9: BASELINE cfg[1] (8 / 9 tile rows) keeps the plain M-fastest order it was tuned with
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int lt, int& tm, int& tn) {
  const int band_tiles = RASTER_GM * tiles_n;
  const int band = lt / band_tiles;
  const int first = band * RASTER_GM;
  const int rows = min(RASTER_GM, tiles_m - first);
  const int in = lt - band * band_tiles;
  tn = in / rows;
  tm = first + (in - tn * rows);
}

// ---- simple data-parallel kernel: one tile per workgroup; blockIdx.y = batch ---------------------------------------
// A_KC: A stored [M,K] (transA = 0).  B_KC: B stored [N,K] (transB = 1).
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGES * TILE_FLOATS];
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, tile, tm, tn);
  process_tile<A_KC, B_KC, false>(g, g.A + (int64_t)blockIdx.y * g.strideA, g.B + (int64_t)blockIdx.y * g.strideB,
                           g.C + (int64_t)blockIdx.y * g.strideC, smem, tm, tn, 0, (g.K + BK - 1) / BK, nullptr);
}

// ---- persistent grouped kernel ----------------------------------------------------------------------------------------
// Up to 4 problems with the same layout flags share one tile space of T tiles; P = 768 resident slots (3 per CU).
//   whole-tile rounds  : the first floor(T / P) * P tiles, one per workgroup, whole K, K-synchronised across the chip so
//                        operand panels are re-used out of L2;
//   remainder (T % P)  : each tile is split along K into S = P / remainder parts; part s of remainder tile r parks its
//                        raw accumulators in ws[(r*S + s)][128][128]; splitk_fixup_kernel then sums the S parts in a
//                        fixed order (deterministic).
// This removes the wave-quantisation tail (e.g. 888 tiles on 768 slots = 58 % -> 98 % slot utilisation).
constexpr int MAX_GROUP = 4;
struct GroupArgs {
  GemmArgs p[MAX_GROUP];
  int tile_base[MAX_GROUP + 1];
  int nprob;
  int T, P, full_rounds, rem, S;
  float* ws;
};

__device__ __forceinline__ int find_problem(const GroupArgs& G, int tile) {
  int q = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < G.nprob && tile >= G.tile_base[i]) q = i;
  return q;
}

template <bool A_KC, bool B_KC, bool BF16, bool VEPI = false>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const GroupArgs G) {
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGES * TILE_FLOATS];
  // One work item per workgroup; the grid is full_rounds*P whole tiles followed by rem*S split-K parts.  The hardware
  // dispatcher hands out workgroups in id order, 3 resident per CU, so the whole-tile rounds run K-synchronised and the
  // parts backfill the slots freed by the last round.
  const int full = G.full_rounds * G.P;
  const int id = blockIdx.x;
  int tile, part = 0, nparts = 1, slot = 0;
  if (id < full) {
    const int round = id / G.P;
    tile = round * G.P + xcd_remap(id - round * G.P, G.P);
  } else {
    slot = xcd_remap(id - full, G.rem * G.S);
    const int rt = slot / G.S;
    part = slot - rt * G.S;
    nparts = G.S;
    tile = full + rt;
  }
  const int q = find_problem(G, tile);
  const GemmArgs& g = G.p[q];
  const int lt = tile - G.tile_base[q];
  const int nk = (g.K + BK - 1) / BK;
  const int kb = (int)((int64_t)nk * part / nparts), ke = (int)((int64_t)nk * (part + 1) / nparts);
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  process_tile<A_KC, B_KC, BF16, VEPI>(g, g.A, g.B, g.C, smem, tm, tn, kb, ke,
                                 nparts > 1 ? G.ws + (int64_t)slot * (BM * BN) : nullptr);
}

// sums the S split-K parts of each remainder tile (fixed order) and applies the normal epilogue
__global__ __launch_bounds__(256) void splitk_fixup_kernel(const GroupArgs G) {
  const int rt = blockIdx.x >> 4, quarter = blockIdx.x & 15;  // 16 workgroups per remainder tile (8 rows each)
  const int tile = G.full_rounds * G.P + rt;
  const int q = find_problem(G, tile);
  const GemmArgs& g = G.p[q];
  const int lt = tile - G.tile_base[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* base = G.ws + (int64_t)rt * G.S * (BM * BN);
  for (int e = quarter * (BM * BN / 16) + threadIdx.x * 4; e < (quarter + 1) * (BM * BN / 16); e += 256 * 4) {
    float4 v = *reinterpret_cast<const float4*>(base + e);
    for (int s = 1; s < G.S; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(base + (int64_t)s * (BM * BN) + e);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int row = m0 + e / BN, col = n0 + (e % BN);
    if (row >= g.M) continue;
    const float vv[4] = {v.x, v.y, v.z, v.w};
    float* c = g.C + (int64_t)row * g.ldc + col;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (col + k < g.N) {
        float o = vv[k] + (g.bias ? g.bias[col + k] : 0.f);
        if (g.accumulate) o += c[k];
        c[k] = o;
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int RESIDENT_PER_CU = 3;   // 76 VGPR + 64 AGPR, 32 KiB LDS => 3 workgroups per CU
constexpr int NUM_CU = 256;
constexpr int SLOTS = RESIDENT_PER_CU * NUM_CU;

}  // namespace

static int fill_problem(GemmArgs& g, int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                        const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float beta) {
  using namespace yt8m;
  YT8M_REQUIRE(M >= 0 && N >= 0 && K >= 0, YT8M_E_SHAPE, "negative dimension");
  YT8M_REQUIRE(beta == 0.f || beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
  YT8M_REQUIRE(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), YT8M_E_SHAPE, "dimension >= 2^31");
  YT8M_REQUIRE(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, YT8M_E_SHAPE, "leading dimension too small");
  if (M > 0 && N > 0) YT8M_REQUIRE(A && B && C, YT8M_E_BADARG, "null operand");
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.strideA = g.strideB = g.strideC = 0;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.tiles_m = (int)((M + BM - 1) / BM);
  g.tiles_n = (int)((N + BN - 1) / BN);
  g.vecA = (lda % 4 == 0) && aligned16(A);
  g.vecB = (ldb % 4 == 0) && aligned16(B);
  g.accumulate = beta != 0.f;
  return YT8M_OK;
}

template <typename KernelArgs>
static void launch_by_layout(int transA, int transB, void (*k00)(KernelArgs), void (*k10)(KernelArgs), void (*k01)(KernelArgs),
                             void (*k11)(KernelArgs), dim3 grid, hipStream_t s, const KernelArgs& a) {
  void (*k)(KernelArgs) = (!transA && !transB) ? k00 : (transA && !transB) ? k10 : (!transA && transB) ? k01 : k11;
  hipLaunchKernelGGL(k, grid, dim3(256), 0, s, a);
}

static int gemm_launch(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                       int64_t strideA, const float* B, int64_t ldb, int64_t strideB, float* C, int64_t ldc,
                       int64_t strideC, const float* bias, float beta, int64_t batch, yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(batch >= 0 && batch <= 65535, YT8M_E_SHAPE, "batch out of range");
  GemmArgs g;
  int rc = fill_problem(g, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, beta);
  if (rc != YT8M_OK) return rc;
  if (M == 0 || N == 0 || batch == 0) return YT8M_OK;
  g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  g.vecA = g.vecA && (strideA % 4 == 0);
  g.vecB = g.vecB && (strideB % 4 == 0);
  const int64_t nwg = (int64_t)g.tiles_m * g.tiles_n;
  YT8M_REQUIRE(nwg < (1LL << 31), YT8M_E_SHAPE, "grid too large");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_GEMM, s, 2.0 * (double)M * (double)N * (double)K * (double)batch);
  launch_by_layout<GemmArgs>(transA, transB, gemm_f32_kernel<true, false>, gemm_f32_kernel<false, false>,
                             gemm_f32_kernel<true, true>, gemm_f32_kernel<false, true>,
                             dim3((unsigned)nwg, (unsigned)batch), s, g);
  return launch_status("gemm_f32_kernel");
}

// room for 4096 split-K parts of 128 x 128 accumulators (256 MiB): several rounds of parts for long-K, few-tile problems
extern "C" int64_t yt8m_gemm_workspace_bytes(void) { return (int64_t)4096 * BM * BN * (int64_t)sizeof(float); }

static int grouped_launch(int transA, int transB, int bf16, int nprob, const yt8m_gemm_problem* probs, void* workspace,
                          int64_t workspace_bytes, yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(nprob >= 1 && nprob <= MAX_GROUP && probs, YT8M_E_BADARG, "1..4 problems");
  GroupArgs G;
  G.nprob = 0;
  int64_t T = 0;
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    GemmArgs g;
    int rc;
    if (bf16) {
      // a bf16 [rows, K] matrix is byte-identical to a float [rows, K/2] matrix: run the float machinery on it
      YT8M_REQUIRE(q.K % 2 == 0 && q.lda % 2 == 0 && q.ldb % 2 == 0, YT8M_E_SHAPE, "bf16 GEMM needs even K / lda / ldb");
      rc = fill_problem(g, 0, 1, q.M, q.N, q.K / 2, static_cast<const float*>(q.A), q.lda / 2, static_cast<const float*>(q.B),
                        q.ldb / 2, q.C, q.ldc, q.bias, q.beta);
    } else {
      rc = fill_problem(g, transA, transB, q.M, q.N, q.K, static_cast<const float*>(q.A), q.lda, static_cast<const float*>(q.B),
                        q.ldb, q.C, q.ldc, q.bias, q.beta);
    }
    if (rc != YT8M_OK) return rc;
    if (q.M == 0 || q.N == 0) continue;
    G.p[G.nprob] = g;
    G.tile_base[G.nprob] = (int)T;
    T += (int64_t)g.tiles_m * g.tiles_n;
    ++G.nprob;
  }
  if (G.nprob == 0) return YT8M_OK;
  YT8M_REQUIRE(T < (1LL << 30), YT8M_E_SHAPE, "too many tiles");
  for (int i = G.nprob; i <= MAX_GROUP; ++i) G.tile_base[i] = (int)T;
  for (int i = G.nprob; i < MAX_GROUP; ++i) G.p[i] = G.p[0];
  G.T = (int)T;
  G.P = T < SLOTS ? (int)T : SLOTS;
  G.full_rounds = (int)(T / G.P);
  G.rem = (int)(T - (int64_t)G.full_rounds * G.P);
  G.S = 1;
  G.ws = static_cast<float*>(workspace);
  if (T < SLOTS) {  // fewer tiles than slots: everything is "remainder" and may be split along K to fill the chip
    G.P = SLOTS;
    G.full_rounds = 0;
    G.rem = (int)T;
  }
  if (G.rem > 0) {
    // Split factor of the remainder tiles, by a cost model in K-steps: ceil(rem * S / P) rounds of parts, each nk / S steps long
    // plus a fixed per-item cost (prologue, epilogue / workspace round trip ~ 8 K-steps).  Long-K, few-tile problems (the LSTM
    // weight gradients: 288 + 256 tiles, K = 38400) used to run ONE round at 544 / 768 slot fill -- three co-resident tiles
    // on some CUs, two on others, i.e. 71 % of the matrix pipe; S = 7 fills five rounds to 99 %.
    int min_nk = 1 << 30, max_nk = 0;
    for (int i = 0; i < G.nprob; ++i) {
      const int nk = (G.p[i].K + BK - 1) / BK;
      if (nk < min_nk) min_nk = nk;
      if (nk > max_nk) max_nk = nk;
    }
    int Smax = min_nk / 8;                    // keep >= 8 K-steps per part (pipeline fill + epilogue amortisation)
    if (Smax > 96) Smax = 96;
    const int64_t ws_items = workspace ? workspace_bytes / ((int64_t)BM * BN * (int64_t)sizeof(float)) : 0;
    if ((int64_t)G.rem * Smax > ws_items) Smax = (int)(ws_items / G.rem);
    int S = 1;
    double best = 1e30;
    for (int c = 1; c <= Smax; ++c) {
      const int64_t rounds = ((int64_t)G.rem * c + G.P - 1) / G.P;
      const double cost = (double)rounds * ((double)max_nk / c + 8.0) + (c > 1 ? 2.0 : 0.0);
      if (cost < best * 0.98) { best = cost; S = c; }     // a larger split must buy >= 2 %
    }
    G.S = S;
  }
  hipStream_t s = as_stream(stream);
  double fl = 0.0;
  for (int i = 0; i < G.nprob; ++i) fl += 2.0 * (double)G.p[i].M * (double)G.p[i].N * (double)G.p[i].K;
  ProfScope prof(F_GEMM, s, fl);
  const int64_t grid = (int64_t)G.full_rounds * G.P + (int64_t)G.rem * G.S;
  int kmax = 0;
  for (int i = 0; i < G.nprob; ++i) kmax = std::max(kmax, G.p[i].K);
  if (bf16)
    hipLaunchKernelGGL((gemm_grouped_kernel<true, true, true>), dim3((unsigned)grid), dim3(256), 0, s, G);
  else if (transA && !transB && kmax <= 2048)     // weight-gradient shapes with a short reduction: float4 epilogue variant
    hipLaunchKernelGGL((gemm_grouped_kernel<false, false, false, true>), dim3((unsigned)grid), dim3(256), 0, s, G);
  else
    launch_by_layout<GroupArgs>(transA, transB, gemm_grouped_kernel<true, false, false>, gemm_grouped_kernel<false, false, false>,
                                gemm_grouped_kernel<true, true, false>, gemm_grouped_kernel<false, true, false>,
                                dim3((unsigned)grid), s, G);
  if (G.S > 1) hipLaunchKernelGGL(splitk_fixup_kernel, dim3((unsigned)G.rem * 16), dim3(256), 0, s, G);
  return launch_status("gemm_grouped_kernel");
}

extern "C" int yt8m_gemm_f32_grouped(int transA, int transB, int nprob, const yt8m_gemm_problem* probs, void* workspace,
                                     int64_t workspace_bytes, yt8m_stream_t stream) {
  return grouped_launch(transA, transB, 0, nprob, probs, workspace, workspace_bytes, stream);
}

namespace yt8m {  // gemm_bf16.hip: 256 x 256 tiles for problems large enough to fill the chip with them
bool gemm_bf16_big_ok(int nprob, const yt8m_gemm_problem* probs);
int gemm_bf16_big_launch(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes, hipStream_t s);
}  // namespace yt8m

extern "C" int yt8m_gemm_bf16_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes,
                                         yt8m_stream_t stream) {
  using namespace yt8m;
  YT8M_REQUIRE(nprob >= 1 && nprob <= MAX_GROUP && probs, YT8M_E_BADARG, "1..4 problems");
  bool simple = true;                              // the large-tile kernel takes validated, aligned problems only
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    simple = simple && q.M >= 0 && q.N >= 0 && q.K >= 0 && q.A && q.B && q.C && q.lda >= q.K && q.ldb >= q.K && q.ldc >= q.N &&
             (q.beta == 0.f || q.beta == 1.f);
  }
  if (simple && gemm_bf16_big_ok(nprob, probs)) {
    double fl = 0.0;
    for (int i = 0; i < nprob; ++i) fl += 2.0 * (double)probs[i].M * (double)probs[i].N * (double)probs[i].K;
    ProfScope prof(F_GEMM, as_stream(stream), fl);
    return gemm_bf16_big_launch(nprob, probs, workspace, workspace_bytes, as_stream(stream));
  }
  return grouped_launch(0, 1, 1, nprob, probs, workspace, workspace_bytes, stream);
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
