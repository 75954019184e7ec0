// gemm_bf16.hip -- large-tile bf16 "NT" GEMM for the bf16 configuration (BASELINE configs[4]); gfx950, wave64.
//
// C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias), A / B bf16 with K contiguous, fp32 accumulate / output (v_mfma_f32_32x32x16_bf16).
//
// Why a second kernel: on the 128 x 128 tiles of gemm_f32.hip a bf16 K-step carries 8x less matrix time than an fp32 one for
// the same operand bytes, and the kernel runs at the rate the L2 delivers operands: 64 FLOP per byte of L2 -> LDS traffic x
// ~8.5 TB/s = 0.55 PFLOP/s measured on the [8192, 2304] x [23580, 2304] MoE heads (22 % of the 2.5 PFLOP/s peak).  Here a
// workgroup owns a 256 x 256 tile (128 FLOP per operand byte): 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 MFMA tiles
// (128 accumulator registers), operands HBM/L2 -> LDS by LDS-DMA into a 4-stage ring (K-step = 32 bf16 = 64 bytes per row,
// 32 KiB per stage, three steps on the wire), the same XOR-swizzled lane-linear image and conflict-free ds_read_b128 fragment
// fetch as gemm_f32.hip, float4 epilogue through LDS.  One tile per workgroup, banded + XCD-aware tile order, up to four
// problems per launch.  K tails (K % 32 != 0) take a guarded zero-padded store for the last step.
#include "common.h"
#include <algorithm>

namespace {

constexpr int TM = 256, TN = 256, BKF = 16;          // BKF: floats per row and K-step (= 32 bf16)
constexpr int TILE_F = BKF * 256;                     // floats of one operand tile (16 KiB)
#ifndef YT8M_BF16_NST
#define YT8M_BF16_NST 4
#endif
#ifndef YT8M_BF16_GM
#define YT8M_BF16_GM 4
#endif
#ifndef YT8M_BF16_STAGGER
#define YT8M_BF16_STAGGER 0        // two wave groups half a K-step apart (0: every wave in lockstep, one barrier per step)
#endif
constexpr int NST = YT8M_BF16_NST;                    // LDS-DMA ring depth (tools/build_variant.sh -DYT8M_BF16_NST=3 for A/B)
constexpr int STAGE_F = 2 * TILE_F;                   // A tile + B tile

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct BArgs {
  const float* A;       // bf16 [M, K] viewed as float [M, K/2]
  const float* B;       // bf16 [N, K] viewed as float [N, K/2]
  float* C;
  const float* bias;
  int64_t lda, ldb, ldc;   // lda / ldb in floats
  int M, N, K;             // K in floats
  int tiles_m, tiles_n;
  int accumulate;
};
// Tile space of a launch: `full` tiles (a multiple of the 256 resident workgroups) run whole; the `rem` tiles of the last, partial
// round are split along K into S parts each (rem * S <= 256 workgroups), parked as raw accumulators in ws[part slot][256][256] and
// summed in a fixed order by bf16_fixup_kernel -- the wave-quantisation tail of e.g. 288 tiles (dx of the MoE heads) costs a
// quarter of a round instead of a whole one.
struct BGroup {
  BArgs p[4];
  int tile_base[5];
  int nprob;
  int full, rem, S;
  float* ws;
};

__device__ __forceinline__ int xcd_remap(int wg, int n) {
  const int xcd = wg & 7, slot = wg >> 3;
  const int q = n >> 3, rem = n & 7;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int lt, int& tm, int& tn) {
  constexpr int GM = YT8M_BF16_GM;                     // 4 x 8 tile blocks per XCD share of a 256-workgroup round
  const int band_tiles = GM * tiles_n;
  const int band = lt / band_tiles;
  const int first = band * GM;
  const int rows = min(GM, tiles_m - first);
  const int in = lt - band * band_tiles;
  tn = in / rows;
  tm = first + (in - tn * rows);
}

// slot idx in [0, 1024): row x = idx >> 2 of the tile, 16-byte chunk slot idx & 3 holding chunk (idx & 3) ^ ((x >> 2) & 3)
__device__ __forceinline__ void fill_dma(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, float* S, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 512;
    const int x = idx >> 2;
    int gx = x0 + x;
    if (gx >= X) gx = X - 1;                           // rows beyond the matrix only feed outputs that are never stored
    const float* src = P + (int64_t)gx * ld + k0 + 4 * ((idx & 3) ^ ((x >> 2) & 3));
    float* dst = S + (idx & ~63) * 4;                  // wave-uniform base; the hardware adds lane * 16 bytes
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}
__device__ __forceinline__ void fill_guarded(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, int K, float* S, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 512;
    const int x = idx >> 2;
    const int gx = x0 + x, gk = k0 + 4 * ((idx & 3) ^ ((x >> 2) & 3));
    float4 r = {0.f, 0.f, 0.f, 0.f};
    if (gx < X) {
      const float* p = P + (int64_t)gx * ld + gk;
      if (gk + 3 < K) r = *reinterpret_cast<const float4*>(p);
      else {
        if (gk + 0 < K) r.x = p[0];
        if (gk + 1 < K) r.y = p[1];
        if (gk + 2 < K) r.z = p[2];
      }
    }
    *reinterpret_cast<float4*>(&S[idx * 4]) = r;
  }
}
__device__ __forceinline__ void fill_step(const BArgs& g, int m0, int n0, int kt, float* stage, int tid) {
  if ((kt + 1) * BKF <= g.K) {
    fill_dma(g.A, g.lda, m0, kt * BKF, g.M, stage, tid);
    fill_dma(g.B, g.ldb, n0, kt * BKF, g.N, stage + TILE_F, tid);
  } else {
    fill_guarded(g.A, g.lda, m0, kt * BKF, g.M, g.K, stage, tid);
    fill_guarded(g.B, g.ldb, n0, kt * BKF, g.N, g.K, stage + TILE_F, tid);
  }
}

__device__ __forceinline__ bf16x8 frag(const float* __restrict__ S, int row, int h, int lk) {
  const float4 q = *reinterpret_cast<const float4*>(&S[row * 16 + 4 * ((2 * h + lk) ^ ((row >> 2) & 3))]);
  return __builtin_bit_cast(bf16x8, q);
}

__device__ __forceinline__ void wait_dma(int younger_steps) {      // 4 DMA instructions per thread and K-step
  if (younger_steps >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (younger_steps == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

// epilogue shared by both large-tile kernels: accumulators -> wave-private LDS image [32][68] -> 16-byte stores (the store pipe is
// issue-bound); split-K parts park raw accumulators in the workspace
__device__ __forceinline__ void tile_epilogue(const BGroup& G, const BArgs& g, f32x16 (&acc)[4][2], float* smem, int m0, int n0, int wm,
                                              int wn, int lane, int wave, int li, int lk, int nparts, int slot) {
  constexpr int P = 68;
  float* st = smem + wave * (32 * P);
  if (nparts > 1) {                                                // split-K part: raw accumulators to the workspace image
    float* wsl = G.ws + (int64_t)slot * (TM * TN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * lk) * P + j * 32 + li] = acc[i][j][r];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = lane + 64 * k;
        const int rr = idx >> 4, c4 = (idx & 15) * 4;
        *reinterpret_cast<float4*>(&wsl[(wm + i * 32 + rr) * TN + wn + c4]) = *reinterpret_cast<const float4*>(&st[rr * P + c4]);
      }
    }
    return;
  }
  const bool vec = (g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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
        float* c = g.C + (int64_t)row * g.ldc + col;
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
}

__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(const BGroup G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];    // NST * STAGE_F floats = 128 KiB
  int tile, part = 0, nparts = 1, slot = 0;
  if ((int)blockIdx.x < G.full) {
    tile = xcd_remap(blockIdx.x, G.full);
  } else {
    slot = blockIdx.x - G.full;
    const int rt = slot / G.S;
    part = slot - rt * G.S;
    nparts = G.S;
    tile = G.full + rt;
  }
  int q = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < G.nprob && tile >= G.tile_base[i]) q = i;
  const BArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, tile - G.tile_base[q], tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int nk_all = (g.K + BKF - 1) / BKF;
  const int kb = (int)((int64_t)nk_all * part / nparts), ke = (int)((int64_t)nk_all * (part + 1) / nparts);
  const int nk = ke - kb;                                          // K-steps kb .. ke-1 of this tile (local index 0 .. nk-1)
  const bool tail = ke == nk_all && nk_all * BKF != g.K;           // the last step is a guarded (non-DMA) fill

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: steps 0, 1, 2 on the wire; wait for step 0 only
  const int pro = nk < 3 ? nk : 3;
  for (int s = 0; s < pro; ++s) fill_step(g, m0, n0, kb + s, smem + s * STAGE_F, tid);
  // DMA steps younger than step 0 that are still allowed in flight; once the guarded tail has been stored (its register
  // loads completed in order behind every DMA) everything has landed and a full drain is exact
  {
    const bool tail_issued = tail && pro == nk;
    wait_dma(tail_issued ? 0 : pro - 1);
  }
  __builtin_amdgcn_s_barrier();
  int cur = 0;
#if YT8M_BF16_STAGGER
  // Two wave groups (waves 0-3 / 4-7: one wave of each on every SIMD) run the K-step in two phases, HALF A STEP APART:
  //   load phase  : LDS-DMA of step kt+3, the 12 fragment reads of step kt, the counted wait for this wave's share of step kt+1
  //   matrix phase: 16 MFMAs at raised priority
  // so that while one wave of a SIMD is held by its DMA issues (~60-100 cycles each) and LDS reads, its partner owns the matrix
  // pipe.  In lockstep (one barrier per step, every wave loading at the same time) the pipe idled through every load phase: the
  // kernel sat at 0.80-0.90 PFLOP/s whatever was done to the instruction order inside a wave.
  // Hazards (interval = span between two barriers; group 0 loads in the even ones, group 1 in the odd ones):
  //   RAW  a stage is read only after BOTH groups waited for their DMA share of it and passed a barrier: group g waits for its
  //        share of step kt+1 at the end of its load phase of step kt, one full interval before the other group reads it.
  //   WAR  the DMA of step kt+3 overwrites the stage of step kt-1, whose last reader (group 1, one interval earlier) drained
  //        lgkmcnt before the barrier that separates the two.
  const int grp = wave >> 2;
  if (grp == 1) __builtin_amdgcn_s_barrier();                      // the half-step offset (group 0 pays it back after the loop)
  for (int kt = 0; kt < nk; ++kt) {
#ifndef YT8M_BF16_NO_DMA
    if (kt + 3 < nk) fill_step(g, m0, n0, kb + kt + 3, smem + ((cur + 3) & 3) * STAGE_F, tid);
#endif
    const float* As = smem + cur * STAGE_F;
    const float* Bs = As + TILE_F;
    bf16x8 a[2][4], b[2][2];
#ifdef YT8M_BF16_NO_LDS
    if (kt == 0)
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[h][t] = frag(As, wm + t * 32 + li, h, lk);
#pragma unroll
      for (int t = 0; t < 2; ++t) b[h][t] = frag(Bs, wn + t * 32 + li, h, lk);
    }
    {
      const int last_issued = kt + 3 < nk ? kt + 3 : nk - 1;
      const bool tail_issued = tail && last_issued == nk - 1;
      int younger = last_issued - (kt + 1);
      if (younger < 0) younger = 0;
      if (tail_issued || younger == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#ifdef YT8M_BF16_NO_MFMA
    acc[0][0][0] += (float)a[0][0][0] + (float)b[1][1][0] + (float)a[1][3][7];
#else
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
#endif
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    cur = (cur + 1) & 3;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#else
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 3 < nk) fill_step(g, m0, n0, kb + kt + 3, smem + ((cur + 3) & 3) * STAGE_F, tid);
    const float* As = smem + cur * STAGE_F;
    const float* Bs = As + TILE_F;
    bf16x8 a[2][4], b[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[h][t] = frag(As, wm + t * 32 + li, h, lk);
#pragma unroll
      for (int t = 0; t < 2; ++t) b[h][t] = frag(Bs, wn + t * 32 + li, h, lk);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // step kt+1 must have landed; kt+2 and kt+3 may stay on the wire
    {
      const int last_issued = kt + 3 < nk ? kt + 3 : nk - 1;
      const bool tail_issued = tail && last_issued == nk - 1;
      int younger = last_issued - (kt + 1);
      if (younger < 0) younger = 0;
      wait_dma(tail_issued ? 0 : younger);
    }
    __builtin_amdgcn_s_barrier();
    cur = (cur + 1) & 3;
  }

#endif

  tile_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot);
}

// ---- K-step 64: whole 128-byte lines per row and step ---------------------------------------------------------------------------
// Ablations of the kernel above (tools/build_variant.sh -DYT8M_* ablation builds: no DMA / no LDS reads / no MFMA builds) show what bounds it: without the
// MFMAs it runs as long as with them, without the LDS-DMA 1.4-1.75x faster -- the L2 -> LDS operand delivery, at ~6.5 TB/s for
// this access pattern: a K-step of 32 bf16 is 64 bytes per row, HALF a cache line, 16 rows per wave instruction, and every line is
// fetched by two different K-steps.  Here a step carries 64 bf16 = one whole 128-byte line per row (8 rows per wave instruction),
// two 64 KiB stages (double buffer: the refill of a stage is issued at the top of the step after its last read and has a whole
// step of 32 MFMAs per wave to land), one barrier per step, the eight LDS-DMA instructions of the refill spread over the
// products.  LDS image: row x of a tile = 8 chunks of 16 bytes, chunk c stored in slot c ^ ((x >> 1) & 7): the 16 rows of a
// ds_read_b128 lane group cover all 16 sixteen-byte slots of the 256-byte bank row (conflict-free).
constexpr int BKF2 = 32;                              // floats per row and step (= 64 bf16)
constexpr int TILE2_F = BKF2 * 256;                   // 32 KiB per operand tile
constexpr int STAGE2_F = 2 * TILE2_F;

__device__ __forceinline__ const float* dma64_src(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, int idx) {
  const int x = idx >> 3;
  int gx = x0 + x;
  if (gx >= X) gx = X - 1;
  return P + (int64_t)gx * ld + k0 + 4 * ((idx & 7) ^ ((x >> 1) & 7));
}
__device__ __forceinline__ void dma64(const float* src, float* S, int idx) {
  float* dst = S + (idx & ~63) * 4;                    // wave-uniform base; the hardware adds lane * 16 bytes
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void fill64_guarded(const float* __restrict__ P, int64_t ld, int x0, int k0, int X, int K, float* S, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 512;
    const int x = idx >> 3;
    const int gx = x0 + x, gk = k0 + 4 * ((idx & 7) ^ ((x >> 1) & 7));
    float4 r = {0.f, 0.f, 0.f, 0.f};
    if (gx < X) {
      const float* p = P + (int64_t)gx * ld + gk;
      if (gk + 3 < K) r = *reinterpret_cast<const float4*>(p);
      else {
        if (gk + 0 < K) r.x = p[0];
        if (gk + 1 < K) r.y = p[1];
        if (gk + 2 < K) r.z = p[2];
      }
    }
    *reinterpret_cast<float4*>(&S[idx * 4]) = r;
  }
}
__device__ __forceinline__ bf16x8 frag64(const float* __restrict__ S, int row, int h, int lk) {
  const float4 q = *reinterpret_cast<const float4*>(&S[row * 32 + 4 * ((2 * h + lk) ^ ((row >> 1) & 7))]);
  return __builtin_bit_cast(bf16x8, q);
}

__global__ __launch_bounds__(512) void gemm_bf16_k64_kernel(const BGroup G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];    // 2 * STAGE2_F floats = 128 KiB
  int tile, part = 0, nparts = 1, slot = 0;
  if ((int)blockIdx.x < G.full) {
    tile = xcd_remap(blockIdx.x, G.full);
  } else {
    slot = blockIdx.x - G.full;
    const int rt = slot / G.S;
    part = slot - rt * G.S;
    nparts = G.S;
    tile = G.full + rt;
  }
  int q = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < G.nprob && tile >= G.tile_base[i]) q = i;
  const BArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, tile - G.tile_base[q], tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int nk_all = (g.K + BKF2 - 1) / BKF2;
  const int kb = (int)((int64_t)nk_all * part / nparts), ke = (int)((int64_t)nk_all * (part + 1) / nparts);
  const int nk = ke - kb;
  const bool tail = ke == nk_all && nk_all * BKF2 != g.K;         // the last step is a guarded (non-DMA) fill

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // whole-step fill (prologue, and the guarded tail): A then B, four 16-byte chunks per thread and operand
  auto fill_all = [&](int kt, float* stage) {
    if (tail && kt == ke - 1) {
      fill64_guarded(g.A, g.lda, m0, kt * BKF2, g.M, g.K, stage, tid);
      fill64_guarded(g.B, g.ldb, n0, kt * BKF2, g.N, g.K, stage + TILE2_F, tid);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma64(dma64_src(g.A, g.lda, m0, kt * BKF2, g.M, tid + i * 512), stage, tid + i * 512);
#pragma unroll
      for (int i = 0; i < 4; ++i) dma64(dma64_src(g.B, g.ldb, n0, kt * BKF2, g.N, tid + i * 512), stage + TILE2_F, tid + i * 512);
    }
  };
  if (nk > 0) fill_all(kb, smem);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const float* As = smem + (kt & 1) * STAGE2_F;
    const float* Bs = As + TILE2_F;
    float* nxt = smem + ((kt + 1) & 1) * STAGE2_F;
    const bool refill = kt + 1 < nk;
    const bool guarded = refill && tail && kb + kt + 1 == ke - 1;
    const int kn = (kb + kt + 1) * BKF2;
    if (guarded) fill_all(kb + kt + 1, nxt);                       // (once per tile at most: plain loads + LDS stores)
    // the refill's source addresses (8 per thread), computed once; issued between the MFMA groups below
    const float* sa[4];
    const float* sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = dma64_src(g.A, g.lda, m0, kn, g.M, tid + i * 512);
      sb[i] = dma64_src(g.B, g.ldb, n0, kn, g.N, tid + i * 512);
    }
    const bool dma_on = refill && !guarded;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                                  // four 16-wide sub-steps: 6 fragment reads + 8 MFMAs each
      bf16x8 a[4], b[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = frag64(As, wm + t * 32 + li, h, lk);
#pragma unroll
      for (int t = 0; t < 2; ++t) b[t] = frag64(Bs, wn + t * 32 + li, h, lk);
      if (dma_on) {
        dma64(sa[h], nxt, tid + h * 512);
        dma64(sb[h], nxt + TILE2_F, tid + h * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the next step landed; this wave's fragment reads are done
    __builtin_amdgcn_s_barrier();
  }
  tile_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot);
}

// sums the S parts of every remainder tile in a fixed order (deterministic) and applies bias / accumulate
__global__ __launch_bounds__(256) void bf16_fixup_kernel(const BGroup G) {
  const int rt = blockIdx.x >> 4, sixteenth = blockIdx.x & 15;      // 16 workgroups per tile, 16 rows each
  const int tile = G.full + rt;
  int q = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < G.nprob && tile >= G.tile_base[i]) q = i;
  const BArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, tile - G.tile_base[q], tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const float* base = G.ws + (int64_t)rt * G.S * (TM * TN);
  for (int e = sixteenth * (TM * TN / 16) + threadIdx.x * 4; e < (sixteenth + 1) * (TM * TN / 16); e += 256 * 4) {
    float4 v = *reinterpret_cast<const float4*>(base + e);
    for (int s = 1; s < G.S; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(base + (int64_t)s * (TM * TN) + e);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int row = m0 + e / TN, col = n0 + (e % TN);
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

}  // namespace

namespace yt8m {

// true when the large-tile kernel can take the whole group (16-byte aligned K-contiguous rows) and there are enough tiles
bool gemm_bf16_big_ok(int nprob, const yt8m_gemm_problem* probs) {
  const char* e = getenv("YT8M_BF16_BIG_MIN");            // A/B switch (tools/, tests): minimum number of 256 x 256 tiles
  const int64_t min_tiles = e ? atoll(e) : 256;                  // at least one full round of 256 x 256 tiles
  int64_t T = 0;
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    if (q.M == 0 || q.N == 0) continue;
    if (q.K < 32 || q.K % 2 || q.lda % 8 || q.ldb % 8) return false;
    if (((uintptr_t)q.A | (uintptr_t)q.B) & 15) return false;
    T += ((q.M + TM - 1) / TM) * ((q.N + TN - 1) / TN);
  }
  return T >= min_tiles;
}

int gemm_bf16_big_launch(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  BGroup G;
  G.nprob = 0;
  int64_t T = 0;
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    if (q.M == 0 || q.N == 0) continue;
    BArgs g;
    g.A = static_cast<const float*>(q.A); g.B = static_cast<const float*>(q.B); g.C = q.C; g.bias = q.bias;
    g.lda = q.lda / 2; g.ldb = q.ldb / 2; g.ldc = q.ldc;
    g.M = (int)q.M; g.N = (int)q.N; g.K = (int)(q.K / 2);
    g.tiles_m = (int)((q.M + TM - 1) / TM); g.tiles_n = (int)((q.N + TN - 1) / TN);
    g.accumulate = q.beta != 0.f;
    G.p[G.nprob] = g;
    G.tile_base[G.nprob] = (int)T;
    T += (int64_t)g.tiles_m * g.tiles_n;
    ++G.nprob;
  }
  if (G.nprob == 0) return YT8M_OK;
  for (int i = G.nprob; i <= 4; ++i) G.tile_base[i] = (int)T;
  for (int i = G.nprob; i < 4; ++i) G.p[i] = G.p[0];
  constexpr int SLOTS = 256;                                       // one 128 KiB workgroup per CU
  G.full = (int)(T / SLOTS) * SLOTS;
  G.rem = (int)(T - G.full);
  G.S = 1;
  G.ws = static_cast<float*>(workspace);
  if (G.rem > 0) {
    int S = SLOTS / G.rem;
    int min_nk = 1 << 30;
    for (int i = 0; i < G.nprob; ++i) min_nk = std::min(min_nk, (G.p[i].K + BKF - 1) / BKF);
    if (S > min_nk / 16) S = min_nk / 16;                          // >= 16 K-steps per part
    if (S > 8) S = 8;
    const int64_t per_part = (int64_t)TM * TN * sizeof(float);
    if (!workspace) S = 1;
    else if ((int64_t)G.rem * S * per_part > workspace_bytes) S = (int)(workspace_bytes / (G.rem * per_part));
    if (S < 1) S = 1;
    G.S = S;
  }
  const int64_t grid = (int64_t)G.full + (int64_t)G.rem * G.S;
  static const int k64 = getenv("YT8M_BF16_K64") ? atoi(getenv("YT8M_BF16_K64")) : 1;      // A/B: 0 = the K-step-32 ring kernel
  if (k64) {
    static DeviceOnce lds_once64;
    YT8M_HIP_CHECK(lds_once64.lds(reinterpret_cast<const void*>(gemm_bf16_k64_kernel), 2 * STAGE2_F * (int)sizeof(float)));
    hipLaunchKernelGGL(gemm_bf16_k64_kernel, dim3((unsigned)grid), dim3(512), 2 * STAGE2_F * sizeof(float), s, G);
  } else {
  static DeviceOnce lds_once;
  YT8M_HIP_CHECK(lds_once.lds(reinterpret_cast<const void*>(gemm_bf16_big_kernel), NST * STAGE_F * (int)sizeof(float)));
  hipLaunchKernelGGL(gemm_bf16_big_kernel, dim3((unsigned)grid), dim3(512), NST * STAGE_F * sizeof(float), s, G);
  }
  if (G.S > 1) hipLaunchKernelGGL(bf16_fixup_kernel, dim3((unsigned)G.rem * 16), dim3(256), 0, s, G);
  return launch_status("gemm_bf16_big_kernel");
}

}  // namespace yt8m
