// netvlad_fused.hip -- NetVLAD soft-assignment + aggregation straight from the RAW uint8 frame features (gfx950).
// (NetVLAD is NOT in the reference: SURVEY.md Appendix B defines it; the uint8 contract it folds in is
//  W/utils.py:23-38 Dequantize + W/readers.py:178-187 padding + W/all_feature_transform/default_transformer.py:7.)
//
//   x[f,:]   = l2_normalize(q[f,:]*(4/255) + (4/512-2))              never materialised
//   s[f,k]   = x[f,:].W_c[:,k] + b_c[k] ;  a = softmax_k(s) * [f < num_frames]          ("rows" kernel)
//   agg[k,:] = sum_f a[f,k] x[f,:]                                                        ("cols" kernel)
// and for the backward, with G = d(loss)/d(agg) [K,D] per video and dn = d(loss)/d(sum_f a):
//   da[f,k]  = x[f,:].G[k,:] + dn[k] ;  ds = a * (da - sum_k a da)                        ("rows" kernel, per-video weights)
//   dW_c     = sum_b sum_f x[f,:]^T ds[f,:] ;  db_c = sum ds                              ("cols" kernel over video groups)
//
// How the uint8 input meets the matrix cores: v_mfma_f32_16x16x32_f16.  Two bytes become two f16 values with ONE
// v_perm_b32 (the bit pattern 0x6400 | q is exactly 1024 + q) and half a v_pk_add_f16 (- 1152 -> q - 128, exact), i.e. the
// dequantise costs ~0.75 VALU op per element and no HBM traffic; the affine remainder (x = r_f (alpha (q - 128) + 0.0157))
// is a rank-1 correction in the epilogue: x.W = r_f ((alpha/S) acc + 0.0157 cs), cs = column sums of the packed
// weights, r_f from the integer row sums (sum q, sum q^2: v_dot4_u32_u8 on the bytes already in registers).  The fp32 operand (W_c, G, a*r,
// ds*r) is split into f16 hi + lo parts (2 MFMAs; 2^-21 relative) after a power-of-two scaling that keeps both parts
// in the f16 normal range -- fp32-class results at 1/8 of the f32-MFMA cost; nsplit = 1 is the f16-operand variant.
//
//   rows kernel : workgroup = (video, range of 64*NT frames); wave w owns row tiles w, w+4, ... (16 frames each) x all 64
//                 clusters; q goes HBM -> VGPR (16 B / lane, one 64-feature block ahead), the packed weights go L2 -> LDS
//                 by LDS-DMA in MFMA fragment order (2 stages, ds_read_b128 lane-linear = conflict-free); softmax over
//                 the 64 clusters = 4 accumulator tiles x 16 lanes, reduced with DPP row moves.  Output: the TRANSPOSED
//                 c = a*r [B,64,Fp] (the MFMA result layout holds 4 consecutive frames per lane, so the store is a
//                 float4; a itself is c / r and is never stored) and n = sum_f a.
//   cols kernel : the reduction runs over frames, the stride dimension of q.  A lane fetches one dword (4 features) for 8
//                 consecutive frames, transposes bytes -> f16 pairs in registers (v_perm_b32 + v_or_b32) and feeds FOUR
//                 MFMAs whose 16 columns are the features 4n+t: the result lands as float4 runs of agg[k, d..d+3].
//                 Workgroup = (384-feature slice, video group); 2x2 waves = 32 clusters x 192 features each; both
//                 operands stream through a 3-stage LDS-DMA ring with counted s_waitcnt vmcnt.
// Bound: HBM (345.6 KB of uint8 per video against 44 MFLOP per GEMM = 128 FLOP/B, below the f16 ridge of 312 FLOP/B).
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include <algorithm>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

constexpr int NK = 64;                       // clusters (fused path)
constexpr float DQ_A = 4.0f / 255.0f;        // W/utils.py:35-38
constexpr float DQ_B = 4.0f / 512.0f - 2.0f;
constexpr float QOFF = 128.0f;               // the MFMA operand is q - 128 (exact in f16): x = alpha (q - 128) + (beta + 128 alpha)
constexpr float DQ_C = DQ_B + QOFF * DQ_A;   // = 0.0156...: the affine remainder is ~1 % of the signal, no cancellation
constexpr uint32_t BIAS2 = 0x64006400u;
constexpr int TMAX = 5;                      // row tiles per wave in the rows kernel (5 * 4 waves * 16 = 320 frames)
constexpr int SCALE_TARGET = 12;             // scaled operands satisfy |v| < 2^12 (f16 max 65504)

// power-of-two scale that brings max |v| = m just below 2^SCALE_TARGET.  Vanishing operands (a saturated model's
// gradients underflow towards fp32 denormals) keep scale 1: they round to zero in f16, an absolute error < 2^-100.
__device__ __forceinline__ float pow2_scale(float m) {
  if (!(m > 7.9e-31f) || !(m < 3.0e38f)) return 1.f;                  // also catches NaN
  int e;
  frexpf(m, &e);
  return ldexpf(1.f, SCALE_TARGET - e);
}

// ---- per-video max |v| in MAXP partial maxima (grid (MAXP, B)); the pack kernel turns them into the power-of-two scale ----
constexpr int MAXP = 16;
__global__ __launch_bounds__(256) void vlad_absmax_part_kernel(const float* __restrict__ v, int64_t n, int64_t stride,
                                                               float* __restrict__ part) {
  __shared__ float red[4];
  const float* p = v + (int64_t)blockIdx.y * stride;
  const int64_t chunk = ((n + MAXP - 1) / MAXP + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  float m = 0.f;
  if ((((uintptr_t)p) & 15) == 0 && (n & 3) == 0) {
    for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i < hi; i += 1024) {
      const float4 x = *reinterpret_cast<const float4*>(p + i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(p[i]));
  }
  m = block_max_256(m, red);
  if (threadIdx.x == 0) part[(int64_t)blockIdx.y * MAXP + blockIdx.x] = m;
}

// ---- packed f16 weights in MFMA fragment order ----------------------------------------------------------------------
// Wp[b][j][ks][ct][sp][lane][i]  (j: block of 64 features, ks: K-step of 32 inside it, ct: 16-cluster tile, sp: hi/lo)
//   feature d = 64 j + 16 (lane / 16) + 8 ks + i      cluster k = 16 ct + lane % 16
// (the rows kernel's lane (m, kg) loads the 16 bytes q[row][64 j + 16 kg .. +16): bytes 0-7 feed K-step 0, 8-15 K-step 1)
// cs_part[b][j][k] = sum over the block's 64 features of the ROUNDED weights (hi + lo) / scale.
// p128 (round 6): the rows kernels walk D in 128-byte blocks -- lane (m, kg) loads the 32 bytes q[row][128 j' + 32 kg .. +32) once per PAIR
// of packed blocks and feeds bytes 0-15 to block 2 j', 16-31 to block 2 j' + 1 -- so packed block j = 2 j' + h holds
//   feature d = 128 j' + 32 (lane / 16) + 16 h + 8 ks + i.
// (With 64-byte blocks a wave touched half of every 128-byte line per block and the other half one block later, by then evicted from the
// L1: every line of the frames was requested twice -- 2.76 M of the rows kernel's 7.86 M requests at B = 1024, profiles/r6_netvlad_pmc.txt.)
template <bool SRC_KD>   // false: src[b][d][k] (W_c, [D,64]);  true: src[b][k][d] (G = d agg, [64,D])
__global__ __launch_bounds__(256) void vlad_pack_kernel(const float* __restrict__ src, int64_t bstride, int D,
                                                        const float* __restrict__ maxpart, float* __restrict__ scale,
                                                        int nsplit, _Float16* __restrict__ Wp, float* __restrict__ cs_part, int p128) {
  __shared__ float tile[64][65];
  const int j = blockIdx.x, b = blockIdx.y, nblk = gridDim.x;
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) mx = fmaxf(mx, maxpart[(int64_t)b * MAXP + i]);
  const float S = pow2_scale(mx);                                     // every thread derives the same scale
  if (j == 0 && threadIdx.x == 0) scale[b] = S;
  const float* sp = src + (int64_t)b * bstride;
  for (int e = threadIdx.x; e < 4096; e += 256) {
    int dl, k;
    float v;
    if (SRC_KD) { k = e >> 6; dl = e & 63; }
    else        { dl = e >> 6; k = e & 63; }
    const int d = p128 ? 128 * (j >> 1) + 32 * (dl >> 4) + 16 * (j & 1) + (dl & 15) : 64 * j + dl;   // packed slot dl -> source feature
    v = SRC_KD ? sp[(int64_t)k * D + d] : sp[(int64_t)d * NK + k];
    tile[dl][k] = v * S;
  }
  __syncthreads();
  _Float16* out = Wp + ((int64_t)b * nblk + j) * (4096 * nsplit);
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int i = e & 7, l = (e >> 3) & 63, ct = (e >> 9) & 3, ks = e >> 11;
    const int dl = 16 * (l >> 4) + 8 * ks + i, k = 16 * ct + (l & 15);
    const float v = tile[dl][k];
    const _Float16 hi = (_Float16)v;
    float eff = (float)hi;
    out[((((ks * 4 + ct) * nsplit + 0) * 64 + l) << 3) + i] = hi;
    if (nsplit == 2) {
      const _Float16 lo = (_Float16)(v - eff);
      out[((((ks * 4 + ct) * nsplit + 1) * 64 + l) << 3) + i] = lo;
      eff += (float)lo;
    }
    tile[dl][k] = eff;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int dl = 0; dl < 64; ++dl) t += tile[dl][threadIdx.x];
    cs_part[((int64_t)b * nblk + j) * NK + threadIdx.x] = t / S;
  }
}

// ---- rows kernel ----------------------------------------------------------------------------------------------------
struct RowsArgs {
  const uint8_t* q;        // [B,F,D]
  const int32_t* nf;       // [B] or null
  const _Float16* Wp;      // packed weights
  int64_t wp_bstride;      // halves between videos (0: shared W_c)
  const float* cs;         // [*, 64] column sums of the (rounded, unscaled) packed weights
  int64_t cs_bstride;      // floats between videos (0: shared)
  const float* wscale;     // [1] or [B]
  int wscale_bstride;      // 0 / 1
  const float* bias;       // fwd: b_c [64];  bwd: dn [B,64]
  int bias_bstride;        // 0 / 64
  const float* cfw;        // bwd: the forward's c = a*r [B,64,Fp] (a is recovered as c / r);  fwd: unused
  float* outT;             // [B,64,Fp]: fwd a*r, bwd ds*r  (frames contiguous: what the cols kernel consumes)
  float* wgmax;            // [B*ranges] max |outT| of the workgroup
  float* colpart;          // [B*ranges,64] sum over the workgroup's frames of a (fwd: n = sum_f a) or ds (bwd: db_c)
  int B, F, D, Fp;
  float eps;
};

// ---- LDS-DMA + inline-asm LDS reads ---------------------------------------------------------------------------------
// Both kernels stream EVERY operand HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) through a
// 3-stage ring with counted s_waitcnt vmcnt(N): two blocks stay in flight while one is consumed.  The LDS reads are
// inline asm on purpose: hipcc inserts s_waitcnt vmcnt(0) before any LDS load it can see while LDS-DMA is pending (and
// before any VGPR load result is used), which would drain the blocks that were only just issued.
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {      // + lane*16 bytes added by hardware
  __builtin_amdgcn_global_load_lds((const AS1 void*)src, (AS3 void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ h8 lds_read_h8(uint32_t addr) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ f4 lds_read_f4(uint32_t addr) {
  f4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// reductions over the 16 lanes of a DPP row (the 16 clusters of one accumulator tile): quad xor-1, quad xor-2, half-row
// mirror, row mirror -- four full-rate VALU DPP moves, no LDS traffic (ds_bpermute-based shuffles made the epilogue cost
// more than the main loop).  Every lane ends up with the reduction over its row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float grp16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_mov<0x140>(v));     // row_mirror
  return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// NT = row tiles per wave (rows per workgroup = 64 NT).
// Per 64-feature block: the packed W block goes L2 -> LDS by LDS-DMA (2 stages), the workgroup's q bytes go HBM -> VGPR
// (16 B per lane and tile) one block ahead.  hipcc drains vmcnt to 0 whenever a VGPR load is consumed while LDS-DMA is
// pending (it treats the DMA as a possibly out-of-order flat access), so the block's q bytes are converted to f16
// fragments FIRST -- that wait covers loads issued a whole block ago -- and only then W(j+1) / q(j+1) are issued; they fly
// during the block's MFMAs.  (Loads through inline asm with counted waits were tried: the register allocator is free to
// copy a not-yet-landed destination register, which it does.)
template <int NSPLIT, bool BWD, int NT, bool P128 = false>
__global__ __launch_bounds__(256, 2) void vlad_rows_kernel(RowsArgs g) {
  constexpr int WB = 8192 * NSPLIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [2][WB] + reduction scratch
  float (*red)[NK + 1] = reinterpret_cast<float (*)[NK + 1]>(smem + 2 * WB);
  const int b = blockIdx.y, range = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int f0 = range * 64 * NT;
  const int nblk = g.D >> 6;

  const char* wsrc = reinterpret_cast<const char*>(g.Wp + (int64_t)b * g.wp_bstride) + tid * 16;
  const uint8_t* qrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int f = f0 + 16 * (w + 4 * t) + m;
    f = f < g.F ? f : g.F - 1;                                       // rows beyond the video only feed outputs never stored
    qrow[t] = g.q + ((int64_t)b * g.F + f) * g.D + (P128 ? 32 : 16) * kg;
  }
  const int wave_off = (tid & ~63) * 16;
  auto issue_w = [&](int j, int stage) {
    char* st = smem + stage * WB + wave_off;
#pragma unroll
    for (int i = 0; i < 2 * NSPLIT; ++i) dma16(wsrc + (int64_t)j * WB + i * 4096, st + i * 4096);
  };
  const uint32_t wfrag = (uint32_t)(uintptr_t)((AS3 char*)smem) + (uint32_t)lane * 16u;

  f4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f4){0.f, 0.f, 0.f, 0.f};
  uint32_t s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0u;

#ifdef NV_TIMING
  const uint64_t tm0 = __builtin_amdgcn_s_memtime();
#endif
  // bytes of one packed block (16 per lane and tile) -> f16 A fragments of its two K steps + the integer row sums
  auto convert = [&](const u4 (&qsrc)[NT], h8 (&af)[NT][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint32_t d0 = qsrc[t][2 * ks], d1 = qsrc[t][2 * ks + 1];
        u4 av;
        av[0] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u);      // [0x64 b1 | 0x64 b0] = f16 (1024+b1, 1024+b0)
        av[1] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u);
        av[2] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u);
        av[3] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u);
        af[t][ks] = __builtin_bit_cast(h8, av) - (_Float16)1152.0f;       // (1024 + q) - 1152 = q - 128, exact
        s1[t] = __builtin_amdgcn_udot4(d0, 0x01010101u, s1[t], false);
        s1[t] = __builtin_amdgcn_udot4(d1, 0x01010101u, s1[t], false);
        s2[t] = __builtin_amdgcn_udot4(d0, d0, s2[t], false);
        s2[t] = __builtin_amdgcn_udot4(d1, d1, s2[t], false);
      }
    }
  };
  // the products of packed block j (its W fragments are in LDS stage j & 1)
  auto products = [&](int j, const h8 (&af)[NT][2]) __attribute__((always_inline)) {
    const uint32_t so = (uint32_t)((j & 1) * WB);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // W fragments in two halves of two cluster tiles when the 128-byte walk holds a second set of frame bytes in registers (NT = 5 ran
      // out of them with all four tiles' fragments live); one group of four otherwise, as before
      constexpr int CG = P128 ? 2 : 4;
#pragma unroll
      for (int c0 = 0; c0 < 4; c0 += CG) {
        h8 bf[CG][NSPLIT];
#pragma unroll
        for (int ct = 0; ct < CG; ++ct)
#pragma unroll
          for (int sp = 0; sp < NSPLIT; ++sp) bf[ct][sp] = lds_read_h8(wfrag + so + (uint32_t)(((ks * 4 + c0 + ct) * NSPLIT + sp) * 1024));
        // the compiler does not track asm loads: wait, and tie the registers to the wait so their uses stay below it
#pragma unroll
        for (int ct = 0; ct < CG; ++ct)
#pragma unroll
          for (int sp = 0; sp < NSPLIT; ++sp) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[ct][sp]) : : "memory");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int sp = 0; sp < NSPLIT; ++sp)
#pragma unroll
            for (int ct = 0; ct < CG; ++ct)
              acc[t][c0 + ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t][ks], bf[ct][sp], acc[t][c0 + ct], 0, 0, 0);
      }
    }
  };
  if constexpr (!P128) {
    u4 qc[NT];
    issue_w(0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) qc[t] = *reinterpret_cast<const u4*>(qrow[t]);
    for (int j = 0; j < nblk; ++j) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                 // W(j) landed for every wave; the other stage is free again
      h8 af[NT][2];
      convert(qc, af);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < nblk) {
        issue_w(j + 1, (j + 1) & 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) qc[t] = *reinterpret_cast<const u4*>(qrow[t] + 64 * (j + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      products(j, af);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // 128-byte blocks: the two 16-byte loads of a lane hit the same line back to back (one L2 request per line of the frames), the
    // pair's second half waits in registers for one block
    u4 q0[NT], q1[NT];
    issue_w(0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      q0[t] = *reinterpret_cast<const u4*>(qrow[t]);
      q1[t] = *reinterpret_cast<const u4*>(qrow[t] + 16);
    }
    for (int j = 0; j < nblk; j += 2) {             // (nblk is even: D % 128 == 0)
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      h8 af[NT][2];
      convert(q0, af);
      __builtin_amdgcn_sched_barrier(0);
      issue_w(j + 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      products(j, af);
      __builtin_amdgcn_sched_barrier(0);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      convert(q1, af);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < nblk) {
        issue_w(j + 2, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          q0[t] = *reinterpret_cast<const u4*>(qrow[t] + 64 * (j + 2));
          q1[t] = *reinterpret_cast<const u4*>(qrow[t] + 64 * (j + 2) + 16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      products(j + 1, af);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                    // LDS is re-used as reduction scratch below
#ifdef NV_TIMING
  const uint64_t tm1 = __builtin_amdgcn_s_memtime();
#endif

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  // ||dequantised frame||^2 from the integer row sums (lane (m, kg) holds a quarter of row m)
  float ssq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t a1 = s1[t], a2 = s2[t];
    a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
    a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
    ssq[t] = fmaxf((DQ_A * DQ_A) * (float)a2 + (2.0f * DQ_A * DQ_B) * (float)a1 + (float)g.D * (DQ_B * DQ_B), g.eps);
  }
  const float S = g.wscale[b * g.wscale_bstride];
  const float A1 = DQ_A / S;
  const int n = m;                                                    // result column of this lane inside a cluster tile
  float cb[4], bs[4];
  {
    const float* cp = g.cs + (int64_t)b * g.cs_bstride;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      cb[ct] = DQ_C * cp[16 * ct + n];
      bs[ct] = g.bias[b * g.bias_bstride + 16 * ct + n];
    }
  }
  const int nfb = g.nf ? min(max(g.nf[b], 0), g.F) : g.F;
  float vmax = 0.f;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  // results leave as float4 runs of 4 consecutive frames per (cluster, lane): 4 stores per tile (the store path is
  // issue-bound: one dword store per element made the epilogue cost more than the main loop)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int fb = f0 + 16 * (w + 4 * t) + 4 * kg;                    // first of this lane's 4 result frames
    f4 cin[4];
    if (BWD) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        cin[ct] = fb < g.Fp ? *reinterpret_cast<const f4*>(g.cfw + ((int64_t)b * NK + 16 * ct + n) * g.Fp + fb)
                            : (f4){0.f, 0.f, 0.f, 0.f};
    }
    f4 ov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = fb + i;
      const float ss = __shfl(ssq[t], 4 * kg + i, 64);
      const float r = rsqrtf(ss);
      float v[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) v[ct] = r * (A1 * acc[t][ct][i] + cb[ct]) + bs[ct];
      if (!BWD) {
        const float mx = grp16_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float e[4], sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { e[ct] = __expf(v[ct] - mx); sum += e[ct]; }
        sum = grp16_sum(sum);
        const float inv = f < nfb ? 1.0f / sum : 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float av = e[ct] * inv;
          csum[ct] += av;
          ov[ct][i] = av * r;
          vmax = fmaxf(vmax, av * r);
        }
      } else {
        const float rinv = sqrtf(ss);                                 // a = c / r
        float av[4], dot = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          av[ct] = cin[ct][i] * rinv;
          dot += av[ct] * v[ct];
        }
        dot = grp16_sum(dot);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float ds = av[ct] * (v[ct] - dot);
          csum[ct] += ds;
          const float e = ds * r;
          vmax = fmaxf(vmax, fabsf(e));
          ov[ct][i] = e;
        }
      }
    }
    if (fb < g.Fp) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        *reinterpret_cast<f4*>(g.outT + ((int64_t)b * NK + 16 * ct + n) * g.Fp + fb) = ov[ct];
    }
  }
  const int slot = b * gridDim.x + range;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float t = csum[ct];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (kg == 0) red[w][16 * ct + n] = t;
  }
  vmax = wave_max(vmax);
  if (lane == 0) red[w][NK] = vmax;
  __syncthreads();
  if (tid < NK) g.colpart[(int64_t)slot * NK + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (tid == 0) g.wgmax[slot] = fmaxf(fmaxf(red[0][NK], red[1][NK]), fmaxf(red[2][NK], red[3][NK]));
#ifdef NV_TIMING   // (timing experiments only) cycles of the main loop / the epilogue land in n_out[b, 0..1]
  __syncthreads();
  if (tid == 0) {
    const uint64_t tm2 = __builtin_amdgcn_s_memtime();
    g.colpart[(int64_t)slot * NK + 0] = (float)(tm1 - tm0);
    g.colpart[(int64_t)slot * NK + 1] = (float)(tm2 - tm1);
  }
#endif
}

// ---- cols kernel ----------------------------------------------------------------------------------------------------
struct ColsArgs {
  const uint8_t* q;        // [B,F,D]
  const float* cT;         // [B,64,Fp]  (a*r or ds*r, zero for frames >= num_frames and in the padding)
  const float* scale;      // device scalar: power of two with |cT| * scale < 2^SCALE_TARGET
  float* out;              // [groups,64,D]
  int B, F, D, Fp, vids;   // vids = videos per workgroup (consecutive)
};

// LDS stage = [q: 32 frames x 384 B of the slice, 16-byte chunks rotated by 4*(row>>3) positions (kg groups of the
// ds_read_b32 fetch land on different banks)][c: 64 cluster rows x 32 frames fp32, chunks XOR-swizzled by row & 7].
template <int NSPLIT>
__global__ __launch_bounds__(256) void vlad_cols_kernel(ColsArgs g) {
  constexpr int QB = 32 * 384, CBYTES = NK * 128, SB = QB + CBYTES;
  constexpr int PER = 5;                                             // DMA instructions per thread and step (3 q + 2 c)
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [3][SB]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const int d0 = blockIdx.x * 384;
  const int dbase = d0 + wn * 192;
  const int grp = blockIdx.y;
  const float S = g.scale[0];
  const int steps = g.Fp >> 5;

  // DMA slots of this thread
  int qrow[3], qd[3], crow[2], cc[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 256 * i, row = idx / 24, cpos = idx - row * 24;
    int c = cpos - 4 * ((row >> 3) & 3);
    c = c < 0 ? c + 24 : c;
    qrow[i] = row;
    qd[i] = min(d0 + 16 * c, g.D - 16);                               // chunks past the end of D (last slice) stay in bounds
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    crow[i] = idx >> 3;
    cc[i] = (idx & 7) ^ (crow[i] & 7);
  }
  const int wave_off = (tid & ~63) * 16;
  const int v0 = grp * g.vids;
  const int v1 = min(v0 + g.vids, g.B);
  const int total = (v1 - v0) * steps;                                // flattened (video, frame step) iterations
  auto issue = [&](int it, int stage) {
    const int vq = it / steps;
    const int v = v0 + vq, s = it - vq * steps;
    char* st = smem + stage * SB + wave_off;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int f = 32 * s + qrow[i];
      f = f < g.F ? f : g.F - 1;                                      // padded frames carry c = 0
      dma16(g.q + ((int64_t)v * g.F + f) * g.D + qd[i], st + i * 4096);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      dma16(g.cT + ((int64_t)v * NK + crow[i]) * g.Fp + 32 * s + 4 * cc[i], st + QB + i * 4096);
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)((AS3 char*)smem);
  uint32_t cfrag[2][2], qfrag[3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      cfrag[mt][h] = lds0 + (uint32_t)(QB + (32 * wm + 16 * mt + n) * 128 + (((2 * kg + h) ^ (n & 7)) << 4));
#pragma unroll
  for (int gq = 0; gq < 3; ++gq) {
    int pos = 12 * wn + 4 * gq + (n >> 2) + 4 * kg;
    pos = pos >= 24 ? pos - 24 : pos;
    qfrag[gq] = lds0 + (uint32_t)(8 * kg * 384 + pos * 16 + (n & 3) * 4);
  }

  f4 acc[2][3][4], acc1[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    acc1[mt] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[mt][gq][t] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  u4 onesv = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  const h8 ones = __builtin_bit_cast(h8, onesv);

  auto compute = [&](int stage) {
    const uint32_t so = (uint32_t)(stage * SB);
    f4 cr[2][2];
    uint32_t qr[3][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) cr[mt][h] = lds_read_f4(cfrag[mt][h] + so);
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
#pragma unroll
      for (int i = 0; i < 8; ++i) qr[gq][i] = lds_read_u32(qfrag[gq] + so + (uint32_t)(i * 384));
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cr[mt][h]) : : "memory");
#pragma unroll
    for (int gq = 0; gq < 3; ++gq)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(qr[gq][i]) : : "memory");
    // split the fp32 operand into scaled f16 hi (+ lo)
    h8 af[2][NSPLIT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      h8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = cr[mt][i >> 2][i & 3] * S;
        hi[i] = (_Float16)x;
        lo[i] = (_Float16)(x - (float)hi[i]);
      }
      af[mt][0] = hi;
      if (NSPLIT == 2) af[mt][NSPLIT - 1] = lo;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int sp = 0; sp < NSPLIT; ++sp)
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][sp], ones, acc1[mt], 0, 0, 0);
#pragma unroll
    for (int gq = 0; gq < 3; ++gq) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // byte t of 8 frame rows -> 8 halves (1024 + q) - 1152 = q - 128, frame pairs packed per register
        const uint32_t sel = 0x0c040c00u + (uint32_t)t * 0x00010001u;   // [0, S0.byte t, 0, S1.byte t]
        u4 bv;
#pragma unroll
        for (int p = 0; p < 4; ++p) bv[p] = __builtin_amdgcn_perm(qr[gq][2 * p + 1], qr[gq][2 * p], sel) | BIAS2;
        const h8 bq = __builtin_bit_cast(h8, bv) - (_Float16)1152.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int sp = 0; sp < NSPLIT; ++sp)
            acc[mt][gq][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][sp], bq, acc[mt][gq][t], 0, 0, 0);
      }
    }
  };

  if (total > 0) {
    issue(0, 0);
    if (total > 1) issue(1, 1);
    for (int it = 0, st = 0; it < total; ++it) {
      if (it + 1 < total) wait_vmcnt<PER>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (it + 2 < total) issue(it + 2, st >= 1 ? st - 1 : 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(st);
      __builtin_amdgcn_sched_barrier(0);
      st = st == 2 ? 0 : st + 1;
    }
  }

  // out[k, d] = sum_f c x = (alpha/S) acc + ((beta + 128 alpha)/S) m1,  m1 = sum_f (scaled, rounded) c
  const float A2 = DQ_A / S, CB = DQ_C / S;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = 32 * wm + 16 * mt + 4 * kg + i;
      const float corr = CB * acc1[mt][i];
#pragma unroll
      for (int gq = 0; gq < 3; ++gq) {
        if (dbase + 64 * gq < g.D) {
          f4 o;
#pragma unroll
          for (int t = 0; t < 4; ++t) o[t] = A2 * acc[mt][gq][t][i] + corr;
          *reinterpret_cast<f4*>(g.out + ((int64_t)grp * NK + k) * g.D + dbase + 64 * gq + 4 * n) = o;
        }
      }
    }
}

// ---- single-pass forward: ONE workgroup owns a video (round 5) ----------------------------
// The rows + cols pair reads the frames twice from HBM and round-trips the assignment through it.  Here a 512-thread workgroup (8
// waves, one workgroup per CU: the LDS is its working set) does both GEMMs of a video in one launch:
//   phase 1  s = x W_c + b_c, a = softmax_k(s) [f < num_frames]: the video's frames stream HBM -> LDS by LDS-DMA in 64-feature blocks
//            (384 frame rows x 64 B, 16-byte chunks swizzled so that the fragment fetch is conflict-free) next to the packed W block
//            (L2 -> LDS, fragment order), three stages with counted vmcnt; wave w owns the row tiles w, w + 8, w + 16.  The
//            softmax epilogue is the rows kernel's (DPP row reductions); c = a r goes to cT (the backward reads it) AND stays in LDS
//            ([64 clusters][320 frames] fp32, 80 KB) -- it never comes back from memory.
//   phase 2  agg[k, :] = sum_f c[k, f] x[f, :]: the video's own 346 KB of frames again, now from L2 / Infinity Cache (this CU read
//            them microseconds ago), 32 frames x D bytes per step through a two-stage LDS-DMA ring; wave (ct = w >> 1, fh = w & 1)
//            keeps its 16 clusters x D / 2 features of agg in registers (9 groups x 4 tiles x 4 = 144) for the whole video: the
//            frames are passed over ONCE for the aggregation too (the cols kernel needs three 384-feature slices).
// HBM sees: the frames once, cT + agg written once, the packed weights from L2.  Outputs and their meaning are exactly those of the
// rows + cols pair (the f16 scale of c is per video here, a power of two either way), so the finishing kernels and the backward are
// unchanged.  Cover: K = 64, D % 128 == 0, D <= 1152, F <= 320 (else the pair runs).
struct VideoArgs {
  const uint8_t* q;        // [B,F,D]
  const int32_t* nf;       // [B] or null
  const _Float16* Wp;      // packed W_c (vlad_pack_kernel<false>)
  const float* cs;         // [64] column sums of the packed weights
  const float* wscale;     // [1]
  const float* bias;       // [64]
  float* cT;               // [B,64,Fp]  a * r
  float* n_out;            // [B,64]     sum_f a
  float* agg;              // [B,64,D]
  int B, F, D, Fp;
  float eps;
};

constexpr int VF = 320;                        // frames of the LDS assignment buffer (10 steps of 32)
constexpr int VROWS = 384;                     // frame rows of a phase-1 block: 8 waves x 3 tiles x 16
constexpr int VC_PITCH = VF * 4;               // bytes per cluster row of the assignment buffer
constexpr int VRING = 2 * 32 * 1152;           // phase-2 ring (two stages of 32 frames x <= 1152 B) = 73728: phase 1's 3 stages alias it and c
constexpr int VGMAX = 9;                       // 64-feature groups per wave in phase 2 (D / 128)

// where (cluster row, frame f) of c lives in the LDS buffer: the 128-byte step of odd rows swaps with its neighbour and the 16-byte
// chunks are XOR-swizzled so that the A-fragment fetch (16 rows x 8 chunks per ds_read_b128 group) touches 16 distinct slots
__device__ __forceinline__ uint32_t vc_addr(int row, int f) {
  const int s = f >> 5, chunk = (f >> 2) & 7;
  const int sig = ((row >> 1) & 1) | (row & 4);
  return (uint32_t)(row * VC_PITCH + ((s ^ (row & 1)) << 7) + ((chunk ^ sig) << 4) + ((f & 3) << 2));
}
// phase-1 frame block: row r holds its four 16-byte chunks at positions chunk ^ g[(r >> 2) & 3], g = {0, 2, 3, 1}
__device__ __forceinline__ int vq_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // {0, 2, 3, 1} in 2-bit fields

template <int NSPLIT>
__global__ __launch_bounds__(512, 2) void vlad_video_kernel(VideoArgs g) {
  constexpr int WB = 8192 * NSPLIT;                                  // packed W block (64 features)
  constexpr int QB1 = VROWS * 64;                                    // phase-1 frame block (24 KB)
  constexpr int ST1 = QB1 + WB;
  constexpr int PER1 = 3 + NSPLIT;                                   // DMA instructions per thread and phase-1 stage
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [ring | c buffer | reduction scratch]
  char* const cbuf = smem + VRING;
  float (*red)[NK + 1] = reinterpret_cast<float (*)[NK + 1]>(smem + VRING + NK * VC_PITCH);
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int nblk = g.D >> 6;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((AS3 char*)smem);
  const int wave_off = (tid & ~63) * 16;
  const uint8_t* qv = g.q + (int64_t)b * g.F * g.D;

  // ---- phase 1 ------------------------------------------------------------------------------------------------------------------
  // DMA slots of this thread: three 16-byte chunks of the frame block (LDS position idx -> frame row idx / 4, chunk slot idx % 4)
  const uint8_t* qsrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 512 * i, row = idx >> 2, slot = idx & 3;
    const int f = row < g.F ? row : g.F - 1;                          // rows beyond the video only feed outputs never stored
    qsrc[i] = qv + (int64_t)f * g.D + 16 * (slot ^ vq_swz(row));
  }
  const char* wsrc = reinterpret_cast<const char*>(g.Wp) + tid * 16;
  auto issue1 = [&](int j, int stage) {
    char* st = smem + stage * ST1 + wave_off;
#pragma unroll
    for (int i = 0; i < 3; ++i) dma16(qsrc[i] + 64 * j, st + i * 8192);
#pragma unroll
    for (int i = 0; i < NSPLIT; ++i) dma16(wsrc + (int64_t)j * WB + i * 8192, st + QB1 + i * 8192);
  };
  uint32_t afrag[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int row = 16 * (w + 8 * t) + m;
    afrag[t] = lds0 + (uint32_t)(row * 64 + ((kg ^ vq_swz(row)) << 4));
  }
  const uint32_t wfrag = lds0 + (uint32_t)QB1 + (uint32_t)lane * 16u;

  f4 acc[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f4){0.f, 0.f, 0.f, 0.f};
  uint32_t s1[3], s2[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) s1[t] = s2[t] = 0u;

#ifdef NV_TIMING
  const uint64_t vt0 = __builtin_amdgcn_s_memtime();
#endif
  issue1(0, 0);
  if (nblk > 1) issue1(1, 1);
  for (int j = 0, st = 0; j < nblk; ++j) {
    if (j + 1 < nblk) wait_vmcnt<PER1>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                                     // stage j landed for every wave; stage j - 1 is free again
    if (j + 2 < nblk) issue1(j + 2, st >= 1 ? st - 1 : 2);
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t so = (uint32_t)(st * ST1);
    u4 qc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) asm volatile("ds_read_b128 %0, %1" : "=v"(qc[t]) : "v"(afrag[t] + so) : "memory");
#pragma unroll
    for (int t = 0; t < 3; ++t) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qc[t]) : : "memory");
    h8 af[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint32_t d0 = qc[t][2 * ks], d1 = qc[t][2 * ks + 1];
        u4 av;
        av[0] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u);      // [0x64 b1 | 0x64 b0] = f16 (1024+b1, 1024+b0)
        av[1] = __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u);
        av[2] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u);
        av[3] = __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u);
        af[t][ks] = __builtin_bit_cast(h8, av) - (_Float16)1152.0f;       // (1024 + q) - 1152 = q - 128, exact
        s1[t] = __builtin_amdgcn_udot4(d0, 0x01010101u, s1[t], false);
        s1[t] = __builtin_amdgcn_udot4(d1, 0x01010101u, s1[t], false);
        s2[t] = __builtin_amdgcn_udot4(d0, d0, s2[t], false);
        s2[t] = __builtin_amdgcn_udot4(d1, d1, s2[t], false);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      h8 bf[4][NSPLIT];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int sp = 0; sp < NSPLIT; ++sp) bf[ct][sp] = lds_read_h8(wfrag + so + (uint32_t)(((ks * 4 + ct) * NSPLIT + sp) * 1024));
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int sp = 0; sp < NSPLIT; ++sp) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[ct][sp]) : : "memory");
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int sp = 0; sp < NSPLIT; ++sp)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t][ks], bf[ct][sp], acc[t][ct], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    st = st == 2 ? 0 : st + 1;
  }
  __syncthreads();                                                    // the ring is free: phase 2's first frames can start flying
#ifdef NV_TIMING
  const uint64_t vt1 = __builtin_amdgcn_s_memtime();
#endif

  // ---- phase 2 DMA plumbing (step 0 is issued before the softmax epilogue and lands while it runs) -------------------------------
  const int Dc = g.D >> 4;                                            // 16-byte chunks per frame row
  const int SB2 = 32 * g.D;                                           // bytes of a phase-2 stage
  const int rounds2 = (2 * g.D + 511) >> 9;                           // DMA instructions per thread and stage (the last may cover
  const int steps = g.Fp >> 5;                                        // only the first waves: 2 D chunks, a multiple of 256)
  int r2[5], c2[5];                                                   // this thread's DMA slots of a stage: frame row, source chunk
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int idx = tid + 512 * i;
    const int r = idx / Dc, pc = idx - r * Dc;
    const int c = (pc + 4 * Dc - 4 * (r >> 3)) % Dc;                  // the rotation that spreads the kg groups over the banks
    r2[i] = idx < 2 * g.D ? r : -1;                                   // (wave-uniform: 2 D % 64 == 0)
    c2[i] = 16 * c;
  }
  (void)rounds2;
  auto issue2 = [&](int s, int stage) {
    char* st = smem + stage * SB2 + wave_off;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (r2[i] >= 0) {
        int f = 32 * s + r2[i];
        f = f < g.F ? f : g.F - 1;                                    // padded frames carry c = 0
        dma16(qv + (int64_t)f * g.D + c2[i], st + i * 8192);
      }
    }
  };
  if (steps > 0) issue2(0, 0);

  // ---- phase-1 epilogue: softmax over the 64 clusters, c = a r -> cT and the LDS buffer, n = sum_f a -----------------------------
  float ssq[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    uint32_t a1 = s1[t], a2 = s2[t];
    a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
    a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
    ssq[t] = fmaxf((DQ_A * DQ_A) * (float)a2 + (2.0f * DQ_A * DQ_B) * (float)a1 + (float)g.D * (DQ_B * DQ_B), g.eps);
  }
  const float SW = g.wscale[0];
  const float A1 = DQ_A / SW;
  const int n = m;
  float cb[4], bs[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    cb[ct] = DQ_C * g.cs[16 * ct + n];
    bs[ct] = g.bias[16 * ct + n];
  }
  const int nfb = g.nf ? min(max(g.nf[b], 0), g.F) : g.F;
  float vmax = 0.f;
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int fb = 16 * (w + 8 * t) + 4 * kg;                         // first of this lane's 4 result frames
    f4 ov[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = fb + i;
      const float ss = __shfl(ssq[t], 4 * kg + i, 64);
      const float r = rsqrtf(ss);
      float v[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) v[ct] = r * (A1 * acc[t][ct][i] + cb[ct]) + bs[ct];
      const float mx = grp16_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
      float e[4], sum = 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) { e[ct] = __expf(v[ct] - mx); sum += e[ct]; }
      sum = grp16_sum(sum);
      const float inv = f < nfb ? 1.0f / sum : 0.f;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float av = e[ct] * inv;
        csum[ct] += av;
        ov[ct][i] = av * r;
        vmax = fmaxf(vmax, av * r);
      }
    }
    if (fb < VF) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        *reinterpret_cast<f4*>(cbuf + vc_addr(16 * ct + n, fb)) = ov[ct];
        if (fb < g.Fp) *reinterpret_cast<f4*>(g.cT + ((int64_t)b * NK + 16 * ct + n) * g.Fp + fb) = ov[ct];
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float t = csum[ct];
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (kg == 0) red[w][16 * ct + n] = t;
  }
  vmax = wave_max(vmax);
  if (lane == 0) red[w][NK] = vmax;
  __syncthreads();                                                    // c buffer + partial sums complete
  if (tid < NK) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][tid];
    g.n_out[(int64_t)b * NK + tid] = t;
  }
  float vm = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) vm = fmaxf(vm, red[i][NK]);
  const float S = pow2_scale(vm);                                     // |c| * S < 2^12: both f16 parts stay normal
#ifdef NV_TIMING
  const uint64_t vt2 = __builtin_amdgcn_s_memtime();
#endif

  // ---- phase 2 ------------------------------------------------------------------------------------------------------------------
  const int ct2 = w >> 1, fh = w & 1;
  const int G = g.D >> 7;                                             // 64-feature groups of this wave's half
  uint32_t cfrag[2];
  {
    const int row = 16 * ct2 + n;
    const int sig = ((row >> 1) & 1) | (row & 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) cfrag[h] = lds0 + (uint32_t)(VRING + row * VC_PITCH + (((2 * kg + h) ^ sig) << 4));
  }
  const int crow1 = (16 * ct2 + n) & 1;
  f4 acc2[VGMAX][4], acc1 = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int gq = 0; gq < VGMAX; ++gq)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc2[gq][t] = (f4){0.f, 0.f, 0.f, 0.f};
  u4 onesv = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};
  const h8 ones = __builtin_bit_cast(h8, onesv);
  // byte address (inside a stage) of this lane's dword of group gq in frame row 8 kg: chunk 4 (G fh + gq) + n / 4, rotated by 4 kg
  uint32_t qoff[VGMAX];
#pragma unroll
  for (int gq = 0; gq < VGMAX; ++gq)
    qoff[gq] = (uint32_t)(8 * kg * g.D + (n & 3) * 4 + (((4 * (G * fh + gq) + (n >> 2) + 4 * kg) % Dc) << 4));

  for (int s = 0; s < steps; ++s) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                                     // step s landed for every wave, and every wave is done with
    if (s + 1 < steps) issue2(s + 1, (s + 1) & 1);                    // step s - 1: its stage takes step s + 1 while s is consumed
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t so = lds0 + (uint32_t)((s & 1) * SB2);
    f4 cr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) cr[h] = lds_read_f4(cfrag[h] + (uint32_t)(((s ^ crow1)) << 7));
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cr[h]) : : "memory");
    h8 af2[NSPLIT];
    {
      h8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = cr[i >> 2][i & 3] * S;
        hi[i] = (_Float16)x;
        lo[i] = (_Float16)(x - (float)hi[i]);
      }
      af2[0] = hi;
      if (NSPLIT == 2) af2[NSPLIT - 1] = lo;
    }
#pragma unroll
    for (int sp = 0; sp < NSPLIT; ++sp) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af2[sp], ones, acc1, 0, 0, 0);
#pragma unroll
    for (int gq = 0; gq < VGMAX; ++gq) {
      if (gq < G) {
        const uint32_t qa = so + qoff[gq];
        uint32_t qr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[i] = lds_read_u32(qa + (uint32_t)(i * g.D));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qr[i]) : : "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t sel = 0x0c040c00u + (uint32_t)t * 0x00010001u;     // [0, S0.byte t, 0, S1.byte t]
          u4 bv;
#pragma unroll
          for (int p = 0; p < 4; ++p) bv[p] = __builtin_amdgcn_perm(qr[2 * p + 1], qr[2 * p], sel) | BIAS2;
          const h8 bq = __builtin_bit_cast(h8, bv) - (_Float16)1152.0f;
#pragma unroll
          for (int sp = 0; sp < NSPLIT; ++sp) acc2[gq][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af2[sp], bq, acc2[gq][t], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

#ifdef NV_TIMING
  const uint64_t vt3 = __builtin_amdgcn_s_memtime();
#endif
  // agg[k, d] = sum_f c x = (alpha / S) acc + ((beta + 128 alpha) / S) m1,  m1 = sum_f (scaled, rounded) c
  const float A2 = DQ_A / S, CB = DQ_C / S;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 16 * ct2 + 4 * kg + i;
    const float corr = CB * acc1[i];
    float* orow = g.agg + ((int64_t)b * NK + k) * g.D + (int64_t)G * 64 * fh + 4 * n;
#pragma unroll
    for (int gq = 0; gq < VGMAX; ++gq) {
      if (gq < G) {
        f4 o;
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = A2 * acc2[gq][t][i] + corr;
        *reinterpret_cast<f4*>(orow + 64 * gq) = o;
      }
    }
  }
#ifdef NV_TIMING   // (timing experiments only) cycles of phase 1 / its epilogue / phase 2 / the store tail land in n_out[b, 0..3]
  __syncthreads();
  if (tid == 0) {
    const uint64_t vt4 = __builtin_amdgcn_s_memtime();
    g.n_out[(int64_t)b * NK + 0] = (float)(vt1 - vt0);
    g.n_out[(int64_t)b * NK + 1] = (float)(vt2 - vt1);
    g.n_out[(int64_t)b * NK + 2] = (float)(vt3 - vt2);
    g.n_out[(int64_t)b * NK + 3] = (float)(vt4 - vt3);
  }
#endif
}

// first level of the dW reduction: out[r][k][d] = sum over groups g = r, r + RED2, ... of part[g][k][d]  (fixed order)
constexpr int RED2 = 16;
__global__ __launch_bounds__(256) void vlad_part_reduce_kernel(const float* __restrict__ part, int groups, int64_t n,
                                                               float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const int r = blockIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int gI = r; gI < groups; gI += RED2) {
    const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)gI * n + i);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(out + (int64_t)r * n + i) = s;
}

// dW[d,k] (+)= sum_g part[g][k][d]   (transposing, fixed order);  one workgroup per 64 features
__global__ __launch_bounds__(256) void vlad_dw_reduce_kernel(const float* __restrict__ part, int groups, int D,
                                                             float* __restrict__ dW, int accumulate) {
  __shared__ float tile[NK][65];
  const int d0 = blockIdx.x * 64;
  const int dl = threadIdx.x & 63, kq = threadIdx.x >> 6;             // thread: feature dl, clusters kq, kq+4, ...
  float s[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
  for (int gI = 0; gI < groups; ++gI) {
    const float* p = part + (int64_t)gI * NK * D + d0 + dl;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] += p[(int64_t)(kq + 4 * r) * D];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[kq + 4 * r][dl] = s[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int k = e & 63, d = e >> 6;
    float* o = dW + (int64_t)(d0 + d) * NK + k;
    *o = accumulate ? *o + tile[k][d] : tile[k][d];
  }
}

__global__ __launch_bounds__(256) void vlad_max_to_scale_kernel(const float* __restrict__ v, int64_t n, float* __restrict__ scale) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, v[i]);
  m = block_max_256(m, red);
  if (threadIdx.x == 0) {
    const float s = pow2_scale(m);
    scale[0] = s;
  }
}

// cs[b,k] = sum_j cs_part[b,j,k]  (the rows kernel used to do this sum itself: 18 dependent L2 round trips per cluster
// tile in its epilogue, ~35k cycles per workgroup)
__global__ __launch_bounds__(256) void vlad_cs_sum_kernel(const float* __restrict__ part, int nblk, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (b, k)
  if (i >= n) return;
  const int64_t b = i / NK, k = i - b * NK;
  float t = 0.f;
  for (int j = 0; j < nblk; ++j) t += part[(b * nblk + j) * NK + k];
  out[i] = t;
}

// n[b,k] = sum over the row ranges of a video of the per-workgroup partial sums (fixed order)
__global__ __launch_bounds__(256) void vlad_sum_parts_kernel(const float* __restrict__ part, int ranges, int64_t n,
                                                             float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (b, k)
  if (i >= n) return;
  const int64_t b = i / NK, k = i - b * NK;
  float t = 0.f;
  for (int r = 0; r < ranges; ++r) t += part[(b * ranges + r) * NK + k];
  out[i] = t;
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// rows per workgroup = 64 nt: the tile count that wastes the fewest padded frames while giving the chip >= 2 workgroups
// per CU when the batch allows it
inline int pick_nt(int64_t B, int64_t F) {
  int best = 1;
  double best_score = -1.0;
  for (int nt = 1; nt <= TMAX; ++nt) {
    const int64_t ranges = (F + 64 * nt - 1) / (64 * nt);
    const double eff = (double)F / (double)(ranges * 64 * nt);
    const double fill = std::min(1.0, (double)(B * ranges) / 512.0);
    const double score = eff * fill * (1.0 + 0.02 * nt);             // ties: larger tiles re-read the weights less often
    if (score > best_score) { best_score = score; best = nt; }
  }
  return best;
}

struct Layout {
  int64_t nblk, Fp, ranges, groups, vids;
  int nt;
  // byte offsets into the workspace
  int64_t o_scale, o_maxpart, o_wp, o_cs, o_cssum, o_cT, o_wgmax, o_escale, o_colpart, o_part, o_part2, total;
};

Layout make_layout(int64_t B, int64_t F, int64_t D) {
  Layout L;
  L.nblk = D / 64;
  L.Fp = align_up(F, 32);
  L.nt = pick_nt(B, F);
  L.ranges = (F + 64 * L.nt - 1) / (64 * L.nt);
  const int64_t slices = (D + 383) / 384;
  L.groups = std::min<int64_t>(B, std::max<int64_t>(1, 384 / slices));
  L.vids = (B + L.groups - 1) / L.groups;
  L.groups = (B + L.vids - 1) / L.vids;
  int64_t o = 0;
  L.o_scale = o;   o += align_up((B + 1) * 4, 256);
  L.o_maxpart = o; o += align_up(B * MAXP * 4, 256);
  L.o_wp = o;      o += align_up(B * L.nblk * 4096 * 2 * 2, 256);     // nsplit = 2, per-video weights (backward)
  L.o_cs = o;      o += align_up(B * L.nblk * NK * 4, 256);
  L.o_cssum = o;   o += align_up(B * NK * 4, 256);
  L.o_cT = o;      o += align_up(B * NK * L.Fp * 4, 256);
  L.o_wgmax = o;   o += align_up(B * L.ranges * 4, 256);
  L.o_escale = o;  o += 256;
  L.o_colpart = o; o += align_up(B * L.ranges * NK * 4, 256);
  L.o_part = o;    o += align_up(L.groups * NK * D * 4, 256);
  L.o_part2 = o;   o += align_up(RED2 * NK * D * 4, 256);
  L.total = o;
  return L;
}

// dynamic LDS above 64 KiB has to be allowed per kernel
template <typename K, typename A>
void launch_lds(K kernel, dim3 grid, int lds_bytes, hipStream_t s, const A& args) {
  // (a refused attribute shows up as the launch error launch_status reports right after: hipGetLastError is sticky)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return;
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds_bytes, s, args);
}

// D in 128-byte blocks (see vlad_pack_kernel): OPT-IN, YT8M_NETVLAD_K128=1.  Measured at B = 1024 (profiles/r6_netvlad_pmc.txt): the rows
// kernel's L2 requests drop from 7.85 M to 5.11 M as predicted -- and its duration does not move (2.149 M against 2.148 M GRBM cycles; 50
// instead of 56 requests outstanding per CU): the window was full but not what the kernel waits for.  Forward call 0.36-0.37 against
// 0.38 ms, backward call 0.58 against 0.55 (the per-video pack of dagg reads 64-byte runs in this order), configs[2] step unchanged.
bool rows_p128(int64_t D) {
  static const bool on = getenv("YT8M_NETVLAD_K128") != nullptr && atoi(getenv("YT8M_NETVLAD_K128")) != 0;
  return on && (D % 128) == 0;
}

template <int NSPLIT, bool BWD>
void launch_rows(const RowsArgs& a, int nt, int64_t ranges, hipStream_t s) {
  const dim3 grid((unsigned)ranges, (unsigned)a.B);
  const int lds = 2 * 8192 * NSPLIT + 4 * (NK + 1) * (int)sizeof(float);
  (void)nt;
  if (rows_p128(a.D)) {
    switch (nt) {
      case 1: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 1, true>, grid, lds, s, a); break;
      case 2: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 2, true>, grid, lds, s, a); break;
      case 3: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 3, true>, grid, lds, s, a); break;
      case 4: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 4, true>, grid, lds, s, a); break;
      default: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 5, true>, grid, lds, s, a); break;
    }
    return;
  }
  switch (nt) {
    case 1: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 1>, grid, lds, s, a); break;
    case 2: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 2>, grid, lds, s, a); break;
    case 3: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 3>, grid, lds, s, a); break;
    case 4: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 4>, grid, lds, s, a); break;
    default: launch_lds(vlad_rows_kernel<NSPLIT, BWD, 5>, grid, lds, s, a); break;
  }
}

void launch_cols(const ColsArgs& a, int nsplit, dim3 grid, hipStream_t s) {
  const int lds = 3 * (32 * 384 + NK * 128);
  if (nsplit == 2) launch_lds(vlad_cols_kernel<2>, grid, lds, s, a);
  else launch_lds(vlad_cols_kernel<1>, grid, lds, s, a);
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_netvlad_supported(int64_t B, int64_t F, int64_t D, int64_t K) {
  return (K == NK && D >= 64 && (D % 64) == 0 && B >= 1 && F >= 1 && B <= 65535 && F * D < (int64_t)1 << 31) ? 1 : 0;
}

// 1 when yt8m_netvlad_fwd_u8 runs the single-pass kernel (one workgroup per video: vlad_video_kernel) for this shape, 0 when it runs
// the rows + cols pair.  Default: the pair (the single pass measured slower, DESIGN_LOG 10.3); YT8M_NETVLAD_SINGLE=1 opts in.
static std::atomic<int> g_single_mode{-1};
extern "C" int yt8m_netvlad_set_single(int mode) {               // -1: environment / default (OFF: measured slower), 0: rows + cols pair, 1: single pass
  g_single_mode.store(mode < 0 ? -1 : (mode ? 1 : 0));
  return YT8M_OK;
}
extern "C" int yt8m_netvlad_single_pass(int64_t B, int64_t F, int64_t D, int64_t K) {
  static const int env = getenv("YT8M_NETVLAD_SINGLE") ? atoi(getenv("YT8M_NETVLAD_SINGLE")) : 0;   // measured slower than the pair: opt-in
  const int mode = g_single_mode.load();
  const int on = mode < 0 ? env : mode;
  return (on != 0 && yt8m_netvlad_supported(B, F, D, K) && (D % 128) == 0 && D <= 1152 && F <= VF) ? 1 : 0;
}

extern "C" int64_t yt8m_netvlad_workspace_bytes(int64_t B, int64_t F, int64_t D, int64_t K) {
  if (!yt8m_netvlad_supported(B, F, D, K)) return 0;
  return make_layout(B, F, D).total;
}

extern "C" int yt8m_netvlad_fwd_u8(const uint8_t* q, const int32_t* num_frames, const float* Wc, const float* bc, int64_t B,
                                   int64_t F, int64_t D, int64_t K, int nsplit, float eps, float* cT_out, float* n_out,
                                   float* agg_out, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(yt8m_netvlad_supported(B, F, D, K), YT8M_E_SHAPE, "fused NetVLAD needs K == 64, D % 64 == 0");
  YT8M_REQUIRE(nsplit == 1 || nsplit == 2, YT8M_E_BADARG, "nsplit must be 1 or 2");
  YT8M_REQUIRE(q && Wc && bc && cT_out && n_out && agg_out && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((((uintptr_t)q | (uintptr_t)agg_out | (uintptr_t)cT_out | (uintptr_t)workspace) & 15) == 0, YT8M_E_BADARG,
               "operands must be 16-byte aligned");
  const Layout L = make_layout(B, F, D);
  YT8M_REQUIRE(workspace_bytes >= L.total, YT8M_E_BADARG, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  // "NetVLAD: bytes = 345 600 B + params" -- the frames once, W_c / b_c (the outputs are the next kernel's inputs)
  ProfScope prof8d(F_NETVLAD_FWD, s, 4.0 * (double)B * (double)F * (double)D * NK, (double)B * (double)F * (double)D + 4.0 * (D * NK + NK));
  char* ws = static_cast<char*>(workspace);
  float* scale = reinterpret_cast<float*>(ws + L.o_scale);
  _Float16* Wp = reinterpret_cast<_Float16*>(ws + L.o_wp);
  float* cs = reinterpret_cast<float*>(ws + L.o_cs);
  float* cT = cT_out;
  float* escale = reinterpret_cast<float*>(ws + L.o_escale);
  float* maxpart = reinterpret_cast<float*>(ws + L.o_maxpart);
  float* colpart = reinterpret_cast<float*>(ws + L.o_colpart);
  hipLaunchKernelGGL(vlad_absmax_part_kernel, dim3(MAXP, 1), dim3(256), 0, s, Wc, D * NK, (int64_t)0, maxpart);
  // (the single pass DMAs its frame blocks in the 64-byte order)
  const int p128 = (rows_p128(D) && !yt8m_netvlad_single_pass(B, F, D, K)) ? 1 : 0;
  hipLaunchKernelGGL(vlad_pack_kernel<false>, dim3((unsigned)L.nblk, 1), dim3(256), 0, s, Wc, (int64_t)0, (int)D, maxpart, scale,
                     nsplit, Wp, cs, p128);
  float* wgmax = reinterpret_cast<float*>(ws + L.o_wgmax);
  float* cssum = reinterpret_cast<float*>(ws + L.o_cssum);
  hipLaunchKernelGGL(vlad_cs_sum_kernel, dim3(1), dim3(256), 0, s, cs, (int)L.nblk, (int64_t)NK, cssum);
  const double qbytes = (double)B * (double)F * (double)D, ctbytes = (double)B * NK * (double)L.Fp * 4.0;
  if (yt8m_netvlad_single_pass(B, F, D, K)) {                      // one workgroup per video, the frames read from HBM once
    VideoArgs va;
    va.q = q; va.nf = num_frames; va.Wp = Wp; va.cs = cssum; va.wscale = scale; va.bias = bc; va.cT = cT; va.n_out = n_out;
    va.agg = agg_out; va.B = (int)B; va.F = (int)F; va.D = (int)D; va.Fp = (int)L.Fp; va.eps = eps;
    const int lds = VRING + NK * VC_PITCH + 8 * (NK + 1) * (int)sizeof(float);
    // SURVEY.md 8(d) bytes: the uint8 frames once + the parameters (the packed W_c)
    ProfScope pv(F_VLAD_ROWS, s, 4.0 * qbytes * NK, qbytes + (double)L.nblk * 4096 * 2 * nsplit);
    if (nsplit == 2) {
      YT8M_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_video_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(vlad_video_kernel<2>, dim3((unsigned)B), dim3(512), lds, s, va);
    } else {
      YT8M_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_video_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(vlad_video_kernel<1>, dim3((unsigned)B), dim3(512), lds, s, va);
    }
    return launch_status("yt8m_netvlad_fwd_u8 (single pass)");
  }
  RowsArgs ra;
  ra.q = q; ra.nf = num_frames; ra.Wp = Wp; ra.wp_bstride = 0; ra.cs = cssum; ra.cs_bstride = 0; ra.wscale = scale;
  ra.wscale_bstride = 0; ra.bias = bc; ra.bias_bstride = 0; ra.cfw = nullptr; ra.outT = cT; ra.wgmax = wgmax; ra.colpart = colpart;
  ra.B = (int)B; ra.F = (int)F; ra.D = (int)D; ra.Fp = (int)L.Fp; ra.eps = eps;
  {                                                              // algorithmic bytes: the frames once + the transposed assignment
    ProfScope pr(F_VLAD_ROWS, s, 2.0 * qbytes * NK, qbytes + ctbytes + (double)L.nblk * 4096 * 2 * nsplit);
    if (nsplit == 2) launch_rows<2, false>(ra, L.nt, L.ranges, s);
    else launch_rows<1, false>(ra, L.nt, L.ranges, s);
  }
  hipLaunchKernelGGL(vlad_max_to_scale_kernel, dim3(1), dim3(256), 0, s, wgmax, B * L.ranges, escale);
  hipLaunchKernelGGL(vlad_sum_parts_kernel, dim3((unsigned)((B * NK + 255) / 256)), dim3(256), 0, s, colpart, (int)L.ranges, B * NK,
                     n_out);
  ColsArgs ca;
  ca.q = q; ca.cT = cT; ca.scale = escale; ca.out = agg_out; ca.B = (int)B; ca.F = (int)F; ca.D = (int)D; ca.Fp = (int)L.Fp;
  ca.vids = 1;
  const dim3 cgrid((unsigned)((D + 383) / 384), (unsigned)B);
  {
    ProfScope pc(F_VLAD_COLS, s, 2.0 * qbytes * NK, qbytes + ctbytes + (double)B * NK * (double)D * 4.0);
    launch_cols(ca, nsplit, cgrid, s);
  }
  return launch_status("yt8m_netvlad_fwd_u8");
}

extern "C" int yt8m_netvlad_bwd_u8(const uint8_t* q, const int32_t* num_frames, const float* cT, const float* dagg,
                                   const float* dn, int64_t B, int64_t F, int64_t D, int64_t K, int nsplit, float eps,
                                   float* dWc, float dWc_beta, float* dbc, float dbc_beta, void* workspace,
                                   int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(yt8m_netvlad_supported(B, F, D, K), YT8M_E_SHAPE, "fused NetVLAD needs K == 64, D % 64 == 0");
  YT8M_REQUIRE(nsplit == 1 || nsplit == 2, YT8M_E_BADARG, "nsplit must be 1 or 2");
  YT8M_REQUIRE(q && cT && dagg && dn && dWc && dbc && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((dWc_beta == 0.f || dWc_beta == 1.f) && (dbc_beta == 0.f || dbc_beta == 1.f), YT8M_E_BADARG, "beta must be 0 or 1");
  YT8M_REQUIRE((((uintptr_t)q | (uintptr_t)cT | (uintptr_t)workspace) & 15) == 0, YT8M_E_BADARG,
               "operands must be 16-byte aligned");
  const Layout L = make_layout(B, F, D);
  YT8M_REQUIRE(workspace_bytes >= L.total, YT8M_E_BADARG, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  char* ws = static_cast<char*>(workspace);
  float* scale = reinterpret_cast<float*>(ws + L.o_scale);
  _Float16* Wp = reinterpret_cast<_Float16*>(ws + L.o_wp);
  float* cs = reinterpret_cast<float*>(ws + L.o_cs);
  float* eT = reinterpret_cast<float*>(ws + L.o_cT);
  float* wgmax = reinterpret_cast<float*>(ws + L.o_wgmax);
  float* escale = reinterpret_cast<float*>(ws + L.o_escale);
  float* colpart = reinterpret_cast<float*>(ws + L.o_colpart);
  float* part = reinterpret_cast<float*>(ws + L.o_part);
  // per-video weights G[b] = dagg[b] ([64, D]): scale, pack, row pass with the softmax backward in the epilogue
  float* maxpart = reinterpret_cast<float*>(ws + L.o_maxpart);
  float* part2 = reinterpret_cast<float*>(ws + L.o_part2);
  hipLaunchKernelGGL(vlad_absmax_part_kernel, dim3(MAXP, (unsigned)B), dim3(256), 0, s, dagg, NK * D, NK * D, maxpart);
  hipLaunchKernelGGL(vlad_pack_kernel<true>, dim3((unsigned)L.nblk, (unsigned)B), dim3(256), 0, s, dagg, NK * D, (int)D, maxpart,
                     scale, nsplit, Wp, cs, rows_p128(D) ? 1 : 0);
  float* cssum = reinterpret_cast<float*>(ws + L.o_cssum);
  hipLaunchKernelGGL(vlad_cs_sum_kernel, dim3((unsigned)((B * NK + 255) / 256)), dim3(256), 0, s, cs, (int)L.nblk, B * NK, cssum);
  RowsArgs ra;
  ra.q = q; ra.nf = num_frames; ra.Wp = Wp; ra.wp_bstride = L.nblk * 4096 * nsplit; ra.cs = cssum; ra.cs_bstride = NK;
  ra.wscale = scale; ra.wscale_bstride = 1; ra.bias = dn; ra.bias_bstride = NK; ra.cfw = cT; ra.outT = eT;
  ra.wgmax = wgmax; ra.colpart = colpart;
  ra.B = (int)B; ra.F = (int)F; ra.D = (int)D; ra.Fp = (int)L.Fp; ra.eps = eps;
  const double qbytes = (double)B * (double)F * (double)D, ctbytes = (double)B * NK * (double)L.Fp * 4.0;
  {                                                              // frames + per-video packed weights + c^T read, e^T written
    ProfScope pr(F_VLAD_ROWS, s, 2.0 * qbytes * NK, qbytes + 2.0 * ctbytes + (double)B * (double)L.nblk * 4096 * 2 * nsplit);
    if (nsplit == 2) launch_rows<2, true>(ra, L.nt, L.ranges, s);
    else launch_rows<1, true>(ra, L.nt, L.ranges, s);
  }
  hipLaunchKernelGGL(vlad_max_to_scale_kernel, dim3(1), dim3(256), 0, s, wgmax, B * L.ranges, escale);
  ColsArgs ca;
  ca.q = q; ca.cT = eT; ca.scale = escale; ca.out = part; ca.B = (int)B; ca.F = (int)F; ca.D = (int)D; ca.Fp = (int)L.Fp;
  ca.vids = (int)L.vids;
  const dim3 cgrid((unsigned)((D + 383) / 384), (unsigned)L.groups);
  {
    ProfScope pc(F_VLAD_COLS, s, 2.0 * qbytes * NK, qbytes + ctbytes + (double)L.groups * NK * (double)D * 4.0);
    launch_cols(ca, nsplit, cgrid, s);
  }
  if (L.groups > RED2) {
    const int64_t n = NK * D;
    hipLaunchKernelGGL(vlad_part_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256), RED2), dim3(256), 0, s, part, (int)L.groups, n,
                       part2);
    hipLaunchKernelGGL(vlad_dw_reduce_kernel, dim3((unsigned)L.nblk), dim3(256), 0, s, part2, RED2, (int)D, dWc,
                       dWc_beta != 0.f ? 1 : 0);
  } else {
    hipLaunchKernelGGL(vlad_dw_reduce_kernel, dim3((unsigned)L.nblk), dim3(256), 0, s, part, (int)L.groups, (int)D, dWc,
                       dWc_beta != 0.f ? 1 : 0);
  }
  const int st = launch_status("yt8m_netvlad_bwd_u8");
  if (st != YT8M_OK) return st;
  // db_c = sum of the per-workgroup partial sums of ds (fixed order)
  return yt8m_colsum_f32(colpart, B * L.ranges, NK, NK, dbc, dbc_beta, nullptr, 0, stream);
}
