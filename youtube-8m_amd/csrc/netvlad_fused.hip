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
//   rows kernel : workgroup = (video, range of 64*t frames); wave w owns row tiles w, w+4, ... (16 frames each) x all 64
//                 clusters; q goes HBM -> VGPR (16 B / lane, one K-block ahead), the packed weights go L2 -> LDS by
//                 LDS-DMA in MFMA fragment order (3 stages, ds_read_b128 lane-linear = conflict-free); softmax over the
//                 64 clusters = 4 accumulator tiles x 16 lanes, reduced with wave shuffles.  Output: a [B,F,64] and the
//                 TRANSPOSED c = a*r [B,64,Fp] -- the MFMA result layout holds 4 consecutive frames per lane, so the
//                 transposed store is a float4.
//   cols kernel : the reduction runs over frames, the stride dimension of q.  A lane loads one dword (4 features) for 8
//                 consecutive frames, transposes bytes -> f16 pairs in registers (v_perm_b32 + v_or_b32) and feeds FOUR
//                 MFMAs whose 16 columns are the features 4n+t: the result lands as float4 runs of agg[k, d..d+3].
//                 Workgroup = (384-feature slice, video group); 2x2 waves = 32 clusters x 192 features each.
// Bound: HBM (345.6 KB of uint8 per video against 44 MFLOP per GEMM = 128 FLOP/B, below the f16 ridge of 312 FLOP/B).
#include "common.h"
#include <type_traits>
#include <algorithm>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

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
template <bool SRC_KD>   // false: src[b][d][k] (W_c, [D,64]);  true: src[b][k][d] (G = d agg, [64,D])
__global__ __launch_bounds__(256) void vlad_pack_kernel(const float* __restrict__ src, int64_t bstride, int D,
                                                        const float* __restrict__ maxpart, float* __restrict__ scale,
                                                        int nsplit, _Float16* __restrict__ Wp, float* __restrict__ cs_part) {
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
    if (SRC_KD) { k = e >> 6; dl = e & 63; v = sp[(int64_t)k * D + 64 * j + dl]; }
    else        { dl = e >> 6; k = e & 63; v = sp[(int64_t)(64 * j + dl) * NK + k]; }
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
  const float* cs_part;    // [*, nblk, 64]
  int64_t cs_bstride;      // floats between videos (0: shared)
  const float* wscale;     // [1] or [B]
  int wscale_bstride;      // 0 / 1
  const float* bias;       // fwd: b_c [64];  bwd: dn [B,64]
  int bias_bstride;        // 0 / 64
  float* a;                // fwd: out [B,F,64];  bwd: in
  float* outT;             // [B,64,Fp]: fwd a*r, bwd ds*r
  float* wgmax;            // bwd: [B*ranges] max |outT| of the workgroup
  float* colpart;          // bwd: [B*ranges,64] sum over the workgroup's frames of ds
  int B, F, D, Fp;
  float eps;
};

template <int NSPLIT>
__device__ __forceinline__ void rows_fill(const _Float16* src, _Float16* stage, int tid) {
#pragma unroll
  for (int i = 0; i < 2 * NSPLIT; ++i) {
    const int idx = tid + i * 256;                                  // 16-byte slot
    const _Float16* s = src + idx * 8;
    _Float16* d = stage + (idx & ~63) * 8;                          // wave-uniform base; hardware adds lane*16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)d, 16, 0, 0);
  }
}

// fragment reads + MFMAs of one 64-feature block.  The LDS reads are inline asm on purpose: hipcc waits for ALL pending
// LDS-DMA (s_waitcnt vmcnt(0)) before any LDS load it can see -- that would drain the W block / q rows that were only just
// issued for the NEXT block.  The stage being read here was completed by the explicit wait + barrier at the block's top.
template <int OFF>
__device__ __forceinline__ h8 lds_read_b128(uint32_t addr) {
  h8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

template <int NSPLIT, int NT>
__device__ __forceinline__ void rows_mma(uint32_t st /* LDS byte address of the stage + lane*16 */, const h8 (&af)[NT][2],
                                         f4 (&acc)[NT][4]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    h8 bf[4][NSPLIT];
    if (ks == 0) {
      bf[0][0] = lds_read_b128<((0 * 4 + 0) * NSPLIT + 0) * 1024>(st);
      bf[1][0] = lds_read_b128<((0 * 4 + 1) * NSPLIT + 0) * 1024>(st);
      bf[2][0] = lds_read_b128<((0 * 4 + 2) * NSPLIT + 0) * 1024>(st);
      bf[3][0] = lds_read_b128<((0 * 4 + 3) * NSPLIT + 0) * 1024>(st);
      if (NSPLIT == 2) {
        bf[0][NSPLIT - 1] = lds_read_b128<((0 * 4 + 0) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[1][NSPLIT - 1] = lds_read_b128<((0 * 4 + 1) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[2][NSPLIT - 1] = lds_read_b128<((0 * 4 + 2) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[3][NSPLIT - 1] = lds_read_b128<((0 * 4 + 3) * NSPLIT + NSPLIT - 1) * 1024>(st);
      }
    } else {
      bf[0][0] = lds_read_b128<((1 * 4 + 0) * NSPLIT + 0) * 1024>(st);
      bf[1][0] = lds_read_b128<((1 * 4 + 1) * NSPLIT + 0) * 1024>(st);
      bf[2][0] = lds_read_b128<((1 * 4 + 2) * NSPLIT + 0) * 1024>(st);
      bf[3][0] = lds_read_b128<((1 * 4 + 3) * NSPLIT + 0) * 1024>(st);
      if (NSPLIT == 2) {
        bf[0][NSPLIT - 1] = lds_read_b128<((1 * 4 + 0) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[1][NSPLIT - 1] = lds_read_b128<((1 * 4 + 1) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[2][NSPLIT - 1] = lds_read_b128<((1 * 4 + 2) * NSPLIT + NSPLIT - 1) * 1024>(st);
        bf[3][NSPLIT - 1] = lds_read_b128<((1 * 4 + 3) * NSPLIT + NSPLIT - 1) * 1024>(st);
      }
    }
    // the compiler does not track asm loads: wait here, and tie the fragments to the wait so the MFMAs stay below it
    if (NSPLIT == 2)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(bf[0][0]), "+v"(bf[1][0]), "+v"(bf[2][0]), "+v"(bf[3][0]), "+v"(bf[0][NSPLIT - 1]),
                     "+v"(bf[1][NSPLIT - 1]), "+v"(bf[2][NSPLIT - 1]), "+v"(bf[3][NSPLIT - 1])
                   :
                   : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[0][0]), "+v"(bf[1][0]), "+v"(bf[2][0]), "+v"(bf[3][0]) : : "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sp = 0; sp < NSPLIT; ++sp)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t][ks], bf[ct][sp], acc[t][ct], 0, 0, 0);
  }
}

__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NT = row tiles per wave (rows per workgroup = 64 NT).  Everything that issues memory operations is free of run-time
// branches (NT is a template parameter, the pipeline tail is peeled) so that the compiler's s_waitcnt counts are exact:
// a conservative count would wait for the loads issued one block AHEAD and serialise the pipeline.
template <int NSPLIT, bool BWD, int NT>
__global__ __launch_bounds__(256) void vlad_rows_kernel(RowsArgs g) {
  constexpr int BLK = 4096 * NSPLIT;                                 // halves per 64-feature block of packed weights
  __shared__ __attribute__((aligned(16))) _Float16 Ws[2][BLK];
  __shared__ float red[4][NK + 1];
  const int b = blockIdx.y, range = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const int f0 = range * 64 * NT;
  const int nblk = g.D >> 6;

  const uint8_t* qrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int f = f0 + 16 * (w + 4 * t) + m;
    f = f < g.F ? f : g.F - 1;                                       // rows beyond the video only feed outputs never stored
    qrow[t] = g.q + ((int64_t)b * g.F + f) * g.D + 16 * kg;
  }
  const _Float16* Wsrc = g.Wp + (int64_t)b * g.wp_bstride;

  f4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = (f4){0.f, 0.f, 0.f, 0.f};
  uint32_t s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0u;

  // Pipeline (2 LDS stages, ONE barrier per 64-feature block).  hipcc drains vmcnt to 0 whenever a VGPR load is consumed
  // while LDS-DMA is pending (it treats the DMA as a possibly out-of-order flat access), so counted waits are not
  // available here; instead the block's q bytes are converted to f16 fragments FIRST (that wait covers loads issued a
  // whole block ago: q(j) and the DMA of W(j)), then W(j+1) / q(j+1) are issued and fly during the block's MFMAs.
  u4 qc[NT];
  const uint32_t lds_base =
      (uint32_t)(uintptr_t)((__attribute__((address_space(3))) _Float16*)&Ws[0][0]) + (uint32_t)lane * 16u;
  rows_fill<NSPLIT>(Wsrc, Ws[0], tid);
#pragma unroll
  for (int t = 0; t < NT; ++t) qc[t] = *reinterpret_cast<const u4*>(qrow[t]);

  auto block = [&](int j, auto prefetch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // W(j) landed for every wave; stage (j+1)&1 is free again
    h8 af[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
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
    __builtin_amdgcn_sched_barrier(0);
    if (decltype(prefetch)::value) {
      rows_fill<NSPLIT>(Wsrc + (int64_t)(j + 1) * BLK, Ws[(j + 1) & 1], tid);
#pragma unroll
      for (int t = 0; t < NT; ++t) qc[t] = *reinterpret_cast<const u4*>(qrow[t] + 64 * (j + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    rows_mma<NSPLIT, NT>(lds_base + (uint32_t)(j & 1) * (uint32_t)(BLK * 2), af, acc);
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int j = 0;
    for (; j + 1 < nblk; ++j) block(j, std::true_type());
    block(j, std::false_type());
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------------
  // 1/||dequantised frame|| from the integer row sums (lane (m, kg) holds a quarter of row m)
  float rr[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint32_t a1 = s1[t], a2 = s2[t];
    a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
    a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
    const float ss = (DQ_A * DQ_A) * (float)a2 + (2.0f * DQ_A * DQ_B) * (float)a1 + (float)g.D * (DQ_B * DQ_B);
    rr[t] = rsqrtf(fmaxf(ss, g.eps));
  }
  const float S = g.wscale[b * g.wscale_bstride];
  const float A1 = DQ_A / S;
  const int n = m;                                                    // result column of this lane inside a cluster tile
  float cb[4], bs[4];
  {
    const float* cp = g.cs_part + (int64_t)b * g.cs_bstride;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float t = 0.f;
      for (int j = 0; j < nblk; ++j) t += cp[j * NK + 16 * ct + n];
      cb[ct] = DQ_C * t;
      bs[ct] = g.bias[b * g.bias_bstride + 16 * ct + n];
    }
  }
  const int nfb = g.nf ? min(max(g.nf[b], 0), g.F) : g.F;
  float vmax = 0.f;
  float dsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    {
      const int fb = f0 + 16 * (w + 4 * t) + 4 * kg;                  // first of this lane's 4 result frames
      f4 ov[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = fb + i;
        const float r = __shfl(rr[t], 4 * kg + i, 64);
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
            if (f < g.F) g.a[((int64_t)b * g.F + f) * NK + 16 * ct + n] = av;
            ov[ct][i] = av * r;
            vmax = fmaxf(vmax, av * r);
          }
        } else {
          float av[4], dot = 0.f;
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            av[ct] = f < g.F ? g.a[((int64_t)b * g.F + f) * NK + 16 * ct + n] : 0.f;
            dot += av[ct] * v[ct];
          }
          dot = grp16_sum(dot);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const float ds = av[ct] * (v[ct] - dot);
            dsum[ct] += ds;
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
  }
  const int slot = b * gridDim.x + range;
  if (BWD) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      float t = dsum[ct];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (kg == 0) red[w][16 * ct + n] = t;
    }
  }
  vmax = wave_max(vmax);
  if (lane == 0) red[w][NK] = vmax;
  __syncthreads();
  if (BWD && tid < NK) g.colpart[(int64_t)slot * NK + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  if (tid == 0) g.wgmax[slot] = fmaxf(fmaxf(red[0][NK], red[1][NK]), fmaxf(red[2][NK], red[3][NK]));
}

// ---- cols kernel ----------------------------------------------------------------------------------------------------
struct ColsArgs {
  const uint8_t* q;        // [B,F,D]
  const float* cT;         // [B,64,Fp]  (a*r or ds*r, zero for frames >= num_frames and in the padding)
  const float* scale;      // device scalar: power of two with |cT| * scale < 2^SCALE_TARGET
  float* out;              // [groups,64,D]
  int B, F, D, Fp, vids;   // vids = videos per workgroup (consecutive)
};

template <int NSPLIT>
__global__ __launch_bounds__(256) void vlad_cols_kernel(ColsArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const int dbase = blockIdx.x * 384 + wn * 192;
  const int grp = blockIdx.y;
  // 64-feature groups past the end of D (last slice of e.g. D = 1024) are computed on clamped, in-bounds addresses and
  // not stored: the load / MFMA stream stays free of run-time branches (exact s_waitcnt counts, see the rows kernel)
  int goff[3];
#pragma unroll
  for (int gq = 0; gq < 3; ++gq) goff[gq] = min(dbase + 64 * gq, g.D - 64) - dbase;
  const float S = g.scale[0];
  const int steps = g.Fp >> 5;

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

  const int v0 = grp * g.vids;
  const int v1 = min(v0 + g.vids, g.B);
  const int total = (v1 - v0) * steps;                                // flattened (video, frame step) iterations

  // operand registers of one step: c rows (2 cluster tiles x 8 frames) and q (3 groups x 8 frames x 4 features)
  f4 cA[2][2], cN[2][2];
  uint32_t qA[3][8], qN[3][8];
  auto load_step = [&](int it, f4 (&cr)[2][2], uint32_t (&qr)[3][8]) {
    const int v = v0 + it / steps, s = it - (it / steps) * steps;
    const float* cp = g.cT + ((int64_t)v * NK + 32 * wm + n) * g.Fp + 32 * s + 8 * kg;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      cr[mt][0] = *reinterpret_cast<const f4*>(cp + (int64_t)16 * mt * g.Fp);
      cr[mt][1] = *reinterpret_cast<const f4*>(cp + (int64_t)16 * mt * g.Fp + 4);
    }
    const uint8_t* qv = g.q + (int64_t)v * g.F * g.D + dbase + 4 * n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int f = 32 * s + 8 * kg + i;
      f = f < g.F ? f : g.F - 1;                                      // padded frames carry c = 0
      const uint8_t* qp = qv + (int64_t)f * g.D;
#pragma unroll
      for (int gq = 0; gq < 3; ++gq) qr[gq][i] = *reinterpret_cast<const uint32_t*>(qp + goff[gq]);
    }
  };

  auto compute = [&]() {
    // split the fp32 operand into scaled f16 hi (+ lo)
    h8 af[2][NSPLIT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      h8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = cA[mt][i >> 2][i & 3] * S;
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
      {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // byte t of 8 frame rows -> 8 halves (1024 + q), frame pairs packed per register
          const uint32_t sel = 0x0c040c00u + (uint32_t)t * 0x00010001u;   // [0, S0.byte t, 0, S1.byte t]
          u4 bv;
#pragma unroll
          for (int p = 0; p < 4; ++p) bv[p] = __builtin_amdgcn_perm(qA[gq][2 * p + 1], qA[gq][2 * p], sel) | BIAS2;
          const h8 bq = __builtin_bit_cast(h8, bv) - (_Float16)1152.0f;   // q - 128
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int sp = 0; sp < NSPLIT; ++sp)
              acc[mt][gq][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][sp], bq, acc[mt][gq][t], 0, 0, 0);
        }
      }
    }
  };

  if (total > 0) {
    load_step(0, cA, qA);
    for (int it = 0; it + 1 < total; ++it) {
      load_step(it + 1, cN, qN);
      __builtin_amdgcn_sched_barrier(0);
      compute();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { cA[mt][0] = cN[mt][0]; cA[mt][1] = cN[mt][1]; }
#pragma unroll
      for (int gq = 0; gq < 3; ++gq)
#pragma unroll
        for (int i = 0; i < 8; ++i) qA[gq][i] = qN[gq][i];
    }
    compute();
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
  int64_t o_scale, o_maxpart, o_wp, o_cs, o_cT, o_wgmax, o_escale, o_colpart, o_part, o_part2, total;
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
  L.o_cT = o;      o += align_up(B * NK * L.Fp * 4, 256);
  L.o_wgmax = o;   o += align_up(B * L.ranges * 4, 256);
  L.o_escale = o;  o += 256;
  L.o_colpart = o; o += align_up(B * L.ranges * NK * 4, 256);
  L.o_part = o;    o += align_up(L.groups * NK * D * 4, 256);
  L.o_part2 = o;   o += align_up(RED2 * NK * D * 4, 256);
  L.total = o;
  return L;
}

template <int NSPLIT, bool BWD>
void launch_rows(const RowsArgs& a, int nt, int64_t ranges, hipStream_t s) {
  const dim3 grid((unsigned)ranges, (unsigned)a.B), block(256);
  switch (nt) {
    case 1: hipLaunchKernelGGL((vlad_rows_kernel<NSPLIT, BWD, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((vlad_rows_kernel<NSPLIT, BWD, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((vlad_rows_kernel<NSPLIT, BWD, 3>), grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL((vlad_rows_kernel<NSPLIT, BWD, 4>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((vlad_rows_kernel<NSPLIT, BWD, 5>), grid, block, 0, s, a); break;
  }
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_netvlad_supported(int64_t B, int64_t F, int64_t D, int64_t K) {
  return (K == NK && D >= 64 && (D % 64) == 0 && B >= 1 && F >= 1 && B <= 65535 && F * D < (int64_t)1 << 31) ? 1 : 0;
}

extern "C" int64_t yt8m_netvlad_workspace_bytes(int64_t B, int64_t F, int64_t D, int64_t K) {
  if (!yt8m_netvlad_supported(B, F, D, K)) return 0;
  return make_layout(B, F, D).total;
}

extern "C" int yt8m_netvlad_fwd_u8(const uint8_t* q, const int32_t* num_frames, const float* Wc, const float* bc, int64_t B,
                                   int64_t F, int64_t D, int64_t K, int nsplit, float eps, float* a_out, float* agg_out,
                                   void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(yt8m_netvlad_supported(B, F, D, K), YT8M_E_SHAPE, "fused NetVLAD needs K == 64, D % 64 == 0");
  YT8M_REQUIRE(nsplit == 1 || nsplit == 2, YT8M_E_BADARG, "nsplit must be 1 or 2");
  YT8M_REQUIRE(q && Wc && bc && a_out && agg_out && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((((uintptr_t)q | (uintptr_t)agg_out | (uintptr_t)workspace) & 15) == 0, YT8M_E_BADARG, "operands must be 16-byte aligned");
  const Layout L = make_layout(B, F, D);
  YT8M_REQUIRE(workspace_bytes >= L.total, YT8M_E_BADARG, "workspace too small");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_NETVLAD, s);
  char* ws = static_cast<char*>(workspace);
  float* scale = reinterpret_cast<float*>(ws + L.o_scale);
  _Float16* Wp = reinterpret_cast<_Float16*>(ws + L.o_wp);
  float* cs = reinterpret_cast<float*>(ws + L.o_cs);
  float* cT = reinterpret_cast<float*>(ws + L.o_cT);
  float* escale = reinterpret_cast<float*>(ws + L.o_escale);
  float* maxpart = reinterpret_cast<float*>(ws + L.o_maxpart);
  hipLaunchKernelGGL(vlad_absmax_part_kernel, dim3(MAXP, 1), dim3(256), 0, s, Wc, D * NK, (int64_t)0, maxpart);
  hipLaunchKernelGGL(vlad_pack_kernel<false>, dim3((unsigned)L.nblk, 1), dim3(256), 0, s, Wc, (int64_t)0, (int)D, maxpart, scale,
                     nsplit, Wp, cs);
  float* wgmax = reinterpret_cast<float*>(ws + L.o_wgmax);
  RowsArgs ra;
  ra.q = q; ra.nf = num_frames; ra.Wp = Wp; ra.wp_bstride = 0; ra.cs_part = cs; ra.cs_bstride = 0; ra.wscale = scale;
  ra.wscale_bstride = 0; ra.bias = bc; ra.bias_bstride = 0; ra.a = a_out; ra.outT = cT; ra.wgmax = wgmax; ra.colpart = nullptr;
  ra.B = (int)B; ra.F = (int)F; ra.D = (int)D; ra.Fp = (int)L.Fp; ra.eps = eps;
  if (nsplit == 2) launch_rows<2, false>(ra, L.nt, L.ranges, s);
  else launch_rows<1, false>(ra, L.nt, L.ranges, s);
  hipLaunchKernelGGL(vlad_max_to_scale_kernel, dim3(1), dim3(256), 0, s, wgmax, B * L.ranges, escale);
  ColsArgs ca;
  ca.q = q; ca.cT = cT; ca.scale = escale; ca.out = agg_out; ca.B = (int)B; ca.F = (int)F; ca.D = (int)D; ca.Fp = (int)L.Fp;
  ca.vids = 1;
  const dim3 cgrid((unsigned)((D + 383) / 384), (unsigned)B);
  if (nsplit == 2) hipLaunchKernelGGL(vlad_cols_kernel<2>, cgrid, dim3(256), 0, s, ca);
  else hipLaunchKernelGGL(vlad_cols_kernel<1>, cgrid, dim3(256), 0, s, ca);
  return launch_status("yt8m_netvlad_fwd_u8");
}

extern "C" int yt8m_netvlad_bwd_u8(const uint8_t* q, const int32_t* num_frames, const float* a, const float* dagg,
                                   const float* dn, int64_t B, int64_t F, int64_t D, int64_t K, int nsplit, float eps,
                                   float* dWc, float dWc_beta, float* dbc, float dbc_beta, void* workspace,
                                   int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(yt8m_netvlad_supported(B, F, D, K), YT8M_E_SHAPE, "fused NetVLAD needs K == 64, D % 64 == 0");
  YT8M_REQUIRE(nsplit == 1 || nsplit == 2, YT8M_E_BADARG, "nsplit must be 1 or 2");
  YT8M_REQUIRE(q && a && dagg && dn && dWc && dbc && workspace, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((dWc_beta == 0.f || dWc_beta == 1.f) && (dbc_beta == 0.f || dbc_beta == 1.f), YT8M_E_BADARG, "beta must be 0 or 1");
  YT8M_REQUIRE((((uintptr_t)q | (uintptr_t)workspace) & 15) == 0, YT8M_E_BADARG, "operands must be 16-byte aligned");
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
                     scale, nsplit, Wp, cs);
  RowsArgs ra;
  ra.q = q; ra.nf = num_frames; ra.Wp = Wp; ra.wp_bstride = L.nblk * 4096 * nsplit; ra.cs_part = cs; ra.cs_bstride = L.nblk * NK;
  ra.wscale = scale; ra.wscale_bstride = 1; ra.bias = dn; ra.bias_bstride = NK; ra.a = const_cast<float*>(a); ra.outT = eT;
  ra.wgmax = wgmax; ra.colpart = colpart;
  ra.B = (int)B; ra.F = (int)F; ra.D = (int)D; ra.Fp = (int)L.Fp; ra.eps = eps;
  if (nsplit == 2) launch_rows<2, true>(ra, L.nt, L.ranges, s);
  else launch_rows<1, true>(ra, L.nt, L.ranges, s);
  hipLaunchKernelGGL(vlad_max_to_scale_kernel, dim3(1), dim3(256), 0, s, wgmax, B * L.ranges, escale);
  ColsArgs ca;
  ca.q = q; ca.cT = eT; ca.scale = escale; ca.out = part; ca.B = (int)B; ca.F = (int)F; ca.D = (int)D; ca.Fp = (int)L.Fp;
  ca.vids = (int)L.vids;
  const dim3 cgrid((unsigned)((D + 383) / 384), (unsigned)L.groups);
  if (nsplit == 2) hipLaunchKernelGGL(vlad_cols_kernel<2>, cgrid, dim3(256), 0, s, ca);
  else hipLaunchKernelGGL(vlad_cols_kernel<1>, cgrid, dim3(256), 0, s, ca);
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
