// gemm_x3.hip -- fp32 GEMM on the bf16 matrix pipe: three-plane bf16 split of both operands, six MFMA products; gfx950, wave64.
//
// C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias) for fp32 A, B.  Every operand element is split once into three bf16 terms
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (exact: 3 x 8 significand bits)
// and the product is accumulated in fp32 from the six partial products whose weight is >= 2^-16 of a1 b1
//     a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)
// Each partial product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 (8 x 8 bits); the three dropped terms are
// <= 2^-23 |a b| together, the size of the rounding an fp32 FMA commits on the product itself.  tests/test_gpu_round2.py holds
// the comparison with the fp32-MFMA kernel of gemm_f32.hip against an fp64 product (same error, both ~1e-7 relative).
// The bf16 pipe is 16x the fp32 one on this chip (2.5 PFLOP/s vs 157 TFLOP/s dense), so six products cost 0.375 of the fp32
// instruction time: the ceiling is 2.5 / 6 = 417 "fp32-equivalent" TFLOP/s.
//
// Operand image ("x3 image", written by yt8m_x3_split): [rows / 32][K / 16][plane 0..2][32 rows][2 halves][8] bf16 -- one 1 KiB
// block per 32 rows, 16-wide K block and plane, stored exactly as a wave's LDS-DMA instruction lays it into LDS (the half-swap
// that makes the fragment fetch conflict-free included), so every DMA instruction reads 1 KiB of consecutive memory and a wave
// streams 3 KiB per K-step.  (A [row][K/16][3][16] image with 32-byte pieces per lane pair ran the L2 -> LDS path at 7.7 TB/s
// and bound the kernel.)  K is zero-padded to a multiple of 16 (no K tail in the GEMM), rows to a multiple of 32 (never stored).
// Both operands K-contiguous ("NT"): an operand needed with the other orientation is split with the transposing variant.
//
// Kernel: 256 x 256 tile per workgroup, 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 MFMA tiles (128 accumulator registers);
// K-step = 16 (one MFMA K): 6 planes x 256 rows x 32 B = 48 KiB per step by LDS-DMA into a 3-stage ring (two steps on the wire);
// per step and wave 18 ds_read_b128 feed 48 MFMAs (LDS busy ~50 %: the matrix pipe is the bound, unlike the plain bf16 kernel of
// gemm_bf16.hip whose 12 reads feed 16 MFMAs), and the fragment fetch of the next step is interleaved with the products of the
// current one.  Tile order and the float4 epilogue follow gemm_bf16.hip; the tiles of the last partial round are split along K.
#include "common.h"
#include "x3_image.h"
#include "philox.h"
#include <type_traits>
#include <algorithm>
#include <mutex>
#include <stdlib.h>

namespace {

constexpr int TM = 256, TN = 256;
constexpr int PLANE_F = 256 * 8;                      // floats of one plane tile: 256 rows x 32 B
constexpr int OP_F = 3 * PLANE_F;                     // one operand, three planes (24 KiB)
constexpr int NST = 3;
using yt8m_x3::RG_F;
using yt8m_x3::store_block;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct XArgs {
  const float* A;       // x3 image of A ([M rows, K])
  const float* B;       // x3 image of B ([N rows, K])
  float* C;
  const float* bias;
  int64_t ldc;
  int M, N, KB;         // KB: 16-wide K blocks of THIS product
  int ska, skb;         // K blocks between consecutive 32-row groups of the A / B image (= KB unless the product reads a K range
                        // of a larger image: the weight-gradient products of one time chunk out of whole-sequence images)
  int tiles_m, tiles_n;
  int accumulate;
  // optional affine epilogue (the uint8 operand paths): C = alpha * rscale[m] * (acc + cs_scale * cs[n]) + bias[n]
  // (rscale NULL: 1; cs NULL: no rank-1 term).  Forward projection: rscale = 1 / ||x_m||, cs = colsum(W), cs_scale = beta;
  // layer-0 weight gradient: alpha = 4/255, cs = colsum(r (.) dz), cs_scale = beta / alpha.
  const float* rscale;
  const float* cs;
  float cs_scale;
  float alpha;
  // device words with max |operand| (float bits; NULL: none): h2 images whose scale was chosen on the device (round 5) -- see
  // fold_device_scales
  const float* dsa;
  const float* dsb;
  // round 6: C is a bf16 matrix (ldc in bf16 elements; beta = 0 only): the MoE logits of the bf16 configuration leave the b1 kernel at the
  // configuration's own precision -- half the write here, half the read of the two mixing passes behind it (yt8m_gemm_b1_nt_grouped_bf16c)
  int c_bf16;
};
// round-to-nearest-even bf16 bits of a float (NaN stays NaN)
__device__ __forceinline__ unsigned c_bf16_bits(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
// four consecutive elements of a row of C at element offset `off` (fp32: 16-byte store; bf16: 8-byte store)
__device__ __forceinline__ void store_c4(const XArgs& g, int64_t off, float4 v) {
  if (g.c_bf16) {
    uint2 w;
    w.x = c_bf16_bits(v.x) | (c_bf16_bits(v.y) << 16);
    w.y = c_bf16_bits(v.z) | (c_bf16_bits(v.w) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(g.C) + off) = w;
  } else {
    *reinterpret_cast<float4*>(g.C + off) = v;
  }
}
__device__ __forceinline__ void store_c1(const XArgs& g, int64_t off, float v) {
  if (g.c_bf16) reinterpret_cast<unsigned short*>(g.C)[off] = (unsigned short)c_bf16_bits(v);
  else g.C[off] = v;
}
__device__ __forceinline__ float4 affine(const XArgs& g, float4 v, int row, int col) {
  if (g.cs) {                                                        // (N % 4 == 0 and 16-byte aligned cs: checked by the host)
    const float4 cv = *reinterpret_cast<const float4*>(g.cs + col);
    v.x += g.cs_scale * cv.x; v.y += g.cs_scale * cv.y; v.z += g.cs_scale * cv.z; v.w += g.cs_scale * cv.w;
  }
  if (g.rscale) { const float rs = g.rscale[row]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
  if (g.alpha != 1.0f) { v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha; }
  return v;
}
__device__ __forceinline__ bool has_affine(const XArgs& g) { return g.rscale || g.cs || g.alpha != 1.0f; }
// dsa / dsb: device words holding max |operand| as float bits (yt8m_h2_absmax): the operand's h2 image was made under the power of two
// S = pow2_scale_for(max, 14), so the product is scaled back by 1 / (S_a S_b) -- folded into alpha ONCE per workgroup (exact: powers of two)
__device__ __forceinline__ XArgs fold_device_scales(const XArgs& g) {
  XArgs r = g;
  float S = 1.0f;
  if (g.dsa) S *= yt8m_x3::pow2_scale_for(__uint_as_float(reinterpret_cast<const unsigned*>(g.dsa)[0]), 14);
  if (g.dsb) S *= yt8m_x3::pow2_scale_for(__uint_as_float(reinterpret_cast<const unsigned*>(g.dsb)[0]), 14);
  r.alpha *= 1.0f / S;
  r.cs_scale *= S;                                                 // the rank-1 term joins the accumulator BEFORE alpha: it must carry the scale too
  r.dsa = r.dsb = nullptr;
  return r;
}
// Work items of a launch: problem q contributes its first full[q] tiles (whole rounds of 256 workgroups) unsplit, then the
// rem[q] tiles of its last, partial round as rem[q] * S[q] K-part items, part major -- the wave-quantisation tail costs a fraction
// of a round instead of a whole one.  full / rem / S come from the shape of problem q ALONE, so a product is summed in the same
// order whether it is launched by itself or inside a group.  Parts park raw accumulators in ws[slot_base[q] + item][256][256]
// with write-through stores and take a ticket on the tile's arrival counter; the part that arrives LAST sums all S slabs in the
// fixed order 0 .. S-1 (so the result does not depend on which part that was) and runs the epilogue -- the combine can ride on the
// GEMM launch instead of a kernel of its own (MI355X_MICROARCH.md "splitk-seam": write-through slabs + ticket), or x3_fixup_kernel
// sums the slabs in the same order as a separate pass.
// Built, parity-checked and measured in round 4 -- and left OFF by default (YT8M_X3_FUSED_COMBINE=1 turns it on): see TileCounters.
struct XGroup {
  XArgs p[4];
  int full_base[5];     // unsplit tiles, cumulative: workgroups [0, full_base[4]) in XCD-contiguous order
  int part_base[5];     // K-part items, cumulative: the workgroups behind them, XCD-contiguous among themselves
  int slot_base[4];
  int fix_base[5];      // split tiles, cumulative (grid of the fixup pass)
  int full[4], rem[4], S[4];
  int nprob;
  float* ws;
  unsigned* cnt;        // arrival counters of the split tiles (one per tile, zero between launches); NULL: separate fix-up pass
};

__device__ __forceinline__ int xcd_remap(int wg, int n) {
  const int xcd = wg & 7, slot = wg >> 3;
  const int q = n >> 3, rem = n & 7;
  return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + slot;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int lt, int& tm, int& tn) {
  constexpr int GM = 4;
  const int band_tiles = GM * tiles_n;
  const int band = lt / band_tiles;
  const int first = band * GM;
  const int rows = min(GM, tiles_m - first);
  const int in = lt - band * band_tiles;
  tn = in / rows;
  tm = first + (in - tn * rows);
}

// which (problem, tile, K part, workspace slot) a workgroup of a launch works on
__device__ __forceinline__ void x_work_item(const XGroup& G, int& q, int& nparts, int& part, int& slot, int& lt) {
  q = 0; nparts = 1; part = 0; slot = 0;
  if ((int)blockIdx.x < G.full_base[4]) {
    const int item = xcd_remap(blockIdx.x, G.full_base[4]);
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < G.nprob && item >= G.full_base[i]) q = i;
    lt = item - G.full_base[q];
  } else {
    // the K-part items are all equally long: each XCD takes a contiguous run of them too (32 neighbouring tiles of one K range
    // share 4 + 8 operand panels in that XCD's L2)
    const int item = xcd_remap(blockIdx.x - G.full_base[4], G.part_base[4]);
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < G.nprob && item >= G.part_base[i]) q = i;
    const int l2 = item - G.part_base[q];
    nparts = G.S[q];
    part = l2 / G.rem[q];
    lt = G.full[q] + (l2 - part * G.rem[q]);
    slot = G.slot_base[q] + l2;
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_WT = 17;                                         // sc0 sc1: write-through store / fabric-coherent load
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, TM * TN * (int)sizeof(float), 0x00020000);
}
// the summed value of four consecutive elements of a split tile -> C (affine / bias / accumulate), exactly as the unsplit epilogue
__device__ __forceinline__ void finish_store(const XArgs& g, float4 v, int row, int col) {
  if (row >= g.M) return;
  if (has_affine(g) && col + 3 < g.N) v = affine(g, v, row, col);
  float* c = g.C + (int64_t)row * g.ldc + col;
  if ((g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0 && col + 3 < g.N) {
    if (g.bias) {                                                  // the same additions in the same order as the scalar path
      const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (g.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(c);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    store_c4(g, (int64_t)row * g.ldc + col, v);
    return;
  }
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (col + k < g.N) {
      float o = vv[k] + (g.bias ? g.bias[col + k] : 0.f);
      if (g.accumulate) o += c[k];
      store_c1(g, (int64_t)row * g.ldc + col + k, o);
    }
  }
}

// epilogue of the image kernels: accumulators -> wave-private LDS image [32][68] -> 16-byte stores (split-K parts: raw
// accumulators to the workspace slab; the last part to arrive combines)
__device__ __forceinline__ void x_epilogue(const XGroup& G, const XArgs& g, f32x16 (&acc)[4][2], float* smem, int m0, int n0, int wm, int wn,
                                           int lane, int wave, int li, int lk, int nparts, int slot, int q, int lt) {
  constexpr int P = 68;
  float* st = smem + wave * (32 * P);
  if (nparts > 1) {                                                // split-K part: raw accumulators to the workspace slab
    const __amdgpu_buffer_rsrc_t wr = slab_rsrc(G.ws + (int64_t)slot * (TM * TN));
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
        const u32x4 v = *reinterpret_cast<const u32x4*>(&st[rr * P + c4]);
        // write-through: a part of this launch reads the slab back (the guide's publish-large row: also the faster form)
        __builtin_amdgcn_raw_buffer_store_b128(v, wr, ((wm + i * 32 + rr) * TN + wn + c4) * 4, 0, AUX_WT);
      }
    }
    if (!G.cnt) return;                                            // x3_fixup_kernel combines
    // ticket: every slab store of this workgroup has left the CU (vmcnt) before the arrival; the LAST part of the tile combines
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int rt = lt - G.full[q];                                 // which split tile of problem q
    unsigned* flag = reinterpret_cast<unsigned*>(smem + 8 * 32 * P);
    if (threadIdx.x == 0) {
      unsigned* c = G.cnt + G.fix_base[q] + rt;
      const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)nparts - 1u) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      *flag = old;
    }
    __syncthreads();
    if (*flag != (unsigned)nparts - 1u) return;
    // slab of part s of this tile: slot_base + s * rem + rt  (the fixed order 0 .. S-1 of x3_fixup_kernel: bitwise the same sums)
    const float* base = G.ws + (int64_t)(G.slot_base[q] + rt) * (TM * TN);
    const int64_t pstride = (int64_t)G.rem[q] * (TM * TN);
    // 32 float4 per thread and part.  One workgroup combines a whole tile, so the loads must be deep in flight: eight per part and
    // up to four parts at once = 32 x 16 B per lane (the accumulators are dead here), summed in part order afterwards.  (The first
    // form -- four loads, one part at a time -- made the combine a 250 us latency chain per tile: slower than the fix-up kernel.)
    const bool cvec = (g.ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(g.C) | reinterpret_cast<uintptr_t>(g.bias)) & 15) == 0;
    for (int i0 = 0; i0 < TM * TN / 4 / 512; i0 += 8) {
      float4 v[8], oldc[8];
      bool fast[8];
      // the values C already holds (beta = 1: weight gradients accumulate over time parts) are requested FIRST, with the slabs: read
      // one by one inside the store loop they were eight dependent round trips per chunk (the second form of this combine: 23.7 ms
      // per headline step against 23.0 with the separate fix-up kernel)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = ((int)threadIdx.x + 512 * (i0 + u)) * 4;
        const int row = m0 + e / TN, col = n0 + (e % TN);
        fast[u] = cvec && row < g.M && col + 3 < g.N;
        oldc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fast[u] && g.accumulate) oldc[u] = *reinterpret_cast<const float4*>(g.C + (int64_t)row * g.ldc + col);
      }
      for (int sp0 = 0; sp0 < nparts; sp0 += 4) {
        float4 t[4][8];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (sp0 + pp < nparts) {
            const __amdgpu_buffer_rsrc_t rr = slab_rsrc(base + (int64_t)(sp0 + pp) * pstride);
#pragma unroll
            for (int u = 0; u < 8; ++u)
              t[pp][u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, ((int)threadIdx.x + 512 * (i0 + u)) * 16, 0, AUX_WT));
          }
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
          if (sp0 + pp < nparts) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (sp0 + pp == 0) v[u] = t[pp][u];
              else { v[u].x += t[pp][u].x; v[u].y += t[pp][u].y; v[u].z += t[pp][u].z; v[u].w += t[pp][u].w; }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = ((int)threadIdx.x + 512 * (i0 + u)) * 4;
        const int row = m0 + e / TN, col = n0 + (e % TN);
        if (!fast[u]) { finish_store(g, v[u], row, col); continue; }      // edges / unaligned C: the general path
        float4 w = v[u];
        if (has_affine(g)) w = affine(g, w, row, col);
        if (g.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
          w.x += bv.x; w.y += bv.y; w.z += bv.z; w.w += bv.w;
        }
        if (g.accumulate) { w.x += oldc[u].x; w.y += oldc[u].y; w.z += oldc[u].z; w.w += oldc[u].w; }
        store_c4(g, (int64_t)row * g.ldc + col, w);
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
        if (has_affine(g)) v = affine(g, v, row, col);
        if (vec && col + 3 < g.N) {
          if (g.bias) {
            const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          if (g.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(c);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          store_c4(g, (int64_t)row * g.ldc + col, v);
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && col + e < g.N; ++e) {
            float t = vv[e] + (g.bias ? g.bias[col + e] : 0.f);
            if (g.accumulate) t += c[e];
            store_c1(g, (int64_t)row * g.ldc + col + e, t);
          }
        }
      }
    }
  }
}

// One operand, one K block: three DMA instructions (one per plane), each 512 lanes x 16 B = a [256 rows][2 slots] plane tile;
// slot s of row x holds half (s ^ ((x >> 3) & 1)) of the 16-wide block (the image is stored that way), which makes the
// ds_read_b128 fragment fetch conflict-free.  src: this wave's row group at K block 0, + lane * 16 B.
template <int NP>
__device__ __forceinline__ void fill_op(const float* __restrict__ src, int kb, float* S, int tid) {
  src += (int64_t)kb * (NP * RG_F);
  float* dst = S + (tid & ~63) * 4;                    // wave-uniform base; the hardware adds lane * 16 bytes
#pragma unroll
  for (int p = 0; p < NP; ++p)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * RG_F),
                                     (__attribute__((address_space(3))) void*)(dst + p * PLANE_F), 16, 0, 0);
}
// PA = planes of the A image: 3 (general fp32 operand, six products) or 1 (an operand whose elements are exact in bf16 -- the
// uint8 frames minus 128 -- : three products a b1 + a b2 + a b3, exact up to the 2^-26 of the split of b).
template <int PA>
__global__ __launch_bounds__(512) void gemm_x3_kernel(const XGroup G) {
  constexpr int OPA_F = PA * PLANE_F;                              // A planes of a stage, then the three B planes
  constexpr int STAGE_F = OPA_F + OP_F;
  extern __shared__ __attribute__((aligned(16))) float smem[];    // NST * STAGE_F floats (144 KiB at PA = 3)
  int q, nparts, part, slot, lt;
  x_work_item(G, q, nparts, part, slot, lt);
  const XArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int kb0 = (int)((int64_t)g.KB * part / nparts), kb1 = (int)((int64_t)g.KB * (part + 1) / nparts);
  const int nk = kb1 - kb0;
  // this wave's 32-row group of either operand tile (groups beyond the matrix only feed outputs that are never stored)
  const float* pa = g.A + (int64_t)min(m0 / 32 + wave, (g.M + 31) / 32 - 1) * g.ska * (PA * RG_F) + lane * 4;
  const float* pb = g.B + (int64_t)min(n0 / 32 + wave, (g.N + 31) / 32 - 1) * g.skb * (3 * RG_F) + lane * 4;
#define X3_FILL(KBI, STAGE) { fill_op<PA>(pa, KBI, (STAGE), tid); fill_op<3>(pb, KBI, (STAGE) + OPA_F, tid); }
  constexpr int DMA = PA + 3;                                      // LDS-DMA instructions per thread and K-step

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Software pipeline: the fragments of step kt+1 are fetched from LDS while the MFMAs of step kt run, each plane into the
  // registers its last product has just released (the product order below is the one for which every fetch is issued >= 1
  // product = 256 matrix cycles before its first use), so the matrix pipe never waits for LDS.  Ring: at the top of iteration
  // kt the stages hold steps kt+1 (landed), kt+2 (on the wire) and -- refilled there -- kt+3.
  const int pro = nk < 3 ? nk : 3;
  for (int s = 0; s < pro; ++s) X3_FILL(kb0 + s, smem + s * STAGE_F)
  if (pro == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");
  else if (pro == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (wm + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));      // + t * 256 floats per 32 rows, + p * PLANE_F
  const int fb = OPA_F + (wn + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  bf16x8 a2[4], a1[4], a0[4], b0[2], b1[2], b2[2];
  auto load_a = [&](bf16x8 (&d)[4], const float* S, int p) __attribute__((always_inline)) {
#ifdef YT8M_X3_NO_LDS
    if (S != smem) return;
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fa + p * PLANE_F + t * 256]));
  };
  auto load_b = [&](bf16x8 (&d)[2], const float* S, int p) __attribute__((always_inline)) {
#ifdef YT8M_X3_NO_LDS
    if (S != smem) return;
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t) d[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fb + p * PLANE_F + t * 256]));
  };
  auto term = [&](const bf16x8 (&x)[4], const bf16x8 (&y)[2]) __attribute__((always_inline)) {
#ifdef YT8M_X3_NO_MFMA                                            // tuning variant: DMA + LDS traffic only (wrong results)
    acc[0][0][0] += (float)x[0][0] + (float)y[0][0];
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[i], y[j], acc[i][j], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (PA == 3) { load_a(a2, smem, 2); load_a(a1, smem, 1); }
  load_b(b0, smem, 0); load_b(b1, smem, 1); load_a(a0, smem, 0); load_b(b2, smem, 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // stage 0 is refilled right behind the first barrier
  int cur = 0;                                                     // stage of step kt
  // One step.  Top: step kt+1 landed (step kt+2 may stay on the wire); behind the barrier every wave holds step kt in registers,
  // so its stage takes step kt+3 (the fragment fetches in flight read stage kt+1, not the one refilled: no LDS wait).
  // The six LDS-DMA instructions of the refill are spread over the products (an LDS-DMA issue holds the wave for ~60-100 cycles:
  // six in a row at the top of the step idle the matrix pipe of both waves of a SIMD).
  const int wbase = (tid & ~63) * 4;                               // wave-uniform LDS base; the hardware adds lane * 16 bytes
  auto dma = [&](bool on, const float* src, float* dst) __attribute__((always_inline)) {
#ifndef YT8M_X3_NO_DMA                                             // tuning variant: products + LDS traffic only (wrong results)
    if (on)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int kt = 0; kt < nk; ++kt) {
    const int nxt = cur + 1 == NST ? 0 : cur + 1;
    // step kt+1 landed (step kt+2 may stay on the wire) and this wave's fragment fetches of step kt -- the last of them issued at
    // the end of the previous iteration -- have returned: behind the barrier the stage they read is refilled
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef YT8M_X3_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    const bool rf = kt + 3 < nk;
    const float* qa = pa + (int64_t)(kb0 + kt + 3) * (PA * RG_F);
    const float* qb = pb + (int64_t)(kb0 + kt + 3) * (3 * RG_F);
    float* Sc = smem + cur * STAGE_F + wbase;
    const float* Sn = smem + nxt * STAGE_F;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PA == 3) {
      term(a2, b0);
      load_a(a2, Sn, 2);
      dma(rf, qa, Sc);
      term(a1, b0);
      dma(rf, qa + RG_F, Sc + PLANE_F);
      term(a1, b1);
      load_a(a1, Sn, 1);
      dma(rf, qa + 2 * RG_F, Sc + 2 * PLANE_F);
      term(a0, b0);
      load_b(b0, Sn, 0);
      dma(rf, qb, Sc + OPA_F);
      term(a0, b1);
      load_b(b1, Sn, 1);
      dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
      dma(rf, qb + 2 * RG_F, Sc + OPA_F + 2 * PLANE_F);
      term(a0, b2);
      load_a(a0, Sn, 0);
      load_b(b2, Sn, 2);
    } else {
      term(a0, b0);
      load_b(b0, Sn, 0);
      dma(rf, qa, Sc);
      dma(rf, qb, Sc + OPA_F);
      term(a0, b1);
      load_b(b1, Sn, 1);
      dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
      dma(rf, qb + 2 * RG_F, Sc + OPA_F + 2 * PLANE_F);
      term(a0, b2);
      load_a(a0, Sn, 0);
      load_b(b2, Sn, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
#undef X3_FILL
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                    // the epilogue reuses the ring

  x_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot, q, lt);
}

// ---- gemm_x3_kernel with every LDS read / LDS-DMA request placed singly behind an MFMA (round 4) --------------------------------------
// Same ring (three one-block stages, step kt+3 requested into the stage step kt was read from, counted vmcnt), same product order
// and therefore the same sums bit for bit.  What changes is issue placement, after the counters on the one-plane kernel (DESIGN.md
// 9.9): four reads or a request issued in a row outlast the 32 cycles the last MFMA keeps the pipe busy, and the two waves of a
// SIMD, aligned by the barrier, sit in that gap together.  Here each of the step's 18 (10) reads and 6 (4) requests follows its
// own MFMA; a register is re-read as soon as the LAST MFMA that uses it has been issued (i-major MFMA order inside a product
// frees the A fragments one by one); and the two planes the last product holds to its end (a0, b2) alternate between two
// register sets, so the step's last read sits five MFMAs before the barrier instead of right in front of the lgkmcnt(0) the
// refill needs.  The two wave groups (M halves, one wave per SIMD each) take their requests in different slots.
template <int PA>
__global__ __launch_bounds__(512) void gemm_x3q_kernel(const XGroup G) {
  constexpr int OPA_F = PA * PLANE_F;
  constexpr int STAGE_F = OPA_F + OP_F;
  extern __shared__ __attribute__((aligned(16))) float smem[];    // NST * STAGE_F floats
  int q, nparts, part, slot, lt;
  x_work_item(G, q, nparts, part, slot, lt);
  const XArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int wm = grp * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int kb0 = (int)((int64_t)g.KB * part / nparts), kb1 = (int)((int64_t)g.KB * (part + 1) / nparts);
  const int nk = kb1 - kb0;
  const float* pa = g.A + (int64_t)min(m0 / 32 + wave, (g.M + 31) / 32 - 1) * g.ska * (PA * RG_F) + lane * 4;
  const float* pb = g.B + (int64_t)min(n0 / 32 + wave, (g.N + 31) / 32 - 1) * g.skb * (3 * RG_F) + lane * 4;
  constexpr int DMA = PA + 3;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int pro = nk < 3 ? nk : 3;
  for (int s = 0; s < pro; ++s) { fill_op<PA>(pa, kb0 + s, smem + s * STAGE_F, tid); fill_op<3>(pb, kb0 + s, smem + s * STAGE_F + OPA_F, tid); }
  if (pro == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");
  else if (pro == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (wm + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  const int fb = OPA_F + (wn + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  auto rd_a = [&](const float* S, int p, int t) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fa + p * PLANE_F + t * 256]));
  };
  auto rd_b = [&](const float* S, int p, int t) __attribute__((always_inline)) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fb + p * PLANE_F + t * 256]));
  };
  bf16x8 a2[4], a1[4], b0[2], b1[2], a0x[4], b2x[2], a0y[4], b2y[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (PA == 3) { a2[t] = rd_a(smem, 2, t); a1[t] = rd_a(smem, 1, t); }
    a0x[t] = rd_a(smem, 0, t);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) { b0[t] = rd_b(smem, 0, t); b1[t] = rd_b(smem, 1, t); b2x[t] = rd_b(smem, 2, t); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int wbase = (tid & ~63) * 4;
  auto dma = [&](bool on, const float* src, float* dst) __attribute__((always_inline)) {
    if (on)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // one product: eight MFMAs, i-major, `behind(m)` issued after the m-th
  auto term = [&](const bf16x8 (&x)[4], const bf16x8 (&y)[2], auto behind) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[m >> 1], y[m & 1], acc[m >> 1][m & 1], 0, 0, 0);
      behind(m);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int cur = 0;
  // one step: products of step kt from (a2, a1, b0, b1, A0, B2); step kt+1 read into (a2, a1, b0, b1, A0n, B2n)
  auto step = [&](int kt, bf16x8 (&A0)[4], bf16x8 (&B2)[2], bf16x8 (&A0n)[4], bf16x8 (&B2n)[2], auto variant) __attribute__((always_inline)) {
    constexpr int V = decltype(variant)::value;
    constexpr int RQ = V == 0 ? 1 : 5;                             // the request slot of this wave group
    const int nxt = cur + 1 == NST ? 0 : cur + 1;
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool rf = kt + 3 < nk;
    const float* qa = pa + (int64_t)(kb0 + kt + 3) * (PA * RG_F);
    const float* qb = pb + (int64_t)(kb0 + kt + 3) * (3 * RG_F);
    float* Sc = smem + cur * STAGE_F + wbase;
    const float* Sn = smem + nxt * STAGE_F;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PA == 3) {
      term(a2, b0, [&](int m) __attribute__((always_inline)) {
        if (m == RQ) dma(rf, qa, Sc);
        if (m == 2) a2[0] = rd_a(Sn, 2, 0);
        if (m == 4) a2[1] = rd_a(Sn, 2, 1);
        if (m == 6) a2[2] = rd_a(Sn, 2, 2);
      });
      term(a1, b0, [&](int m) __attribute__((always_inline)) {
        if (m == 0) a2[3] = rd_a(Sn, 2, 3);
        if (m == RQ) dma(rf, qa + RG_F, Sc + PLANE_F);
        if (m == 2) A0n[0] = rd_a(Sn, 0, 0);
        if (m == 4) A0n[1] = rd_a(Sn, 0, 1);
        if (m == 6) A0n[2] = rd_a(Sn, 0, 2);
      });
      term(a1, b1, [&](int m) __attribute__((always_inline)) {
        if (m == 0) A0n[3] = rd_a(Sn, 0, 3);
        if (m == RQ) dma(rf, qa + 2 * RG_F, Sc + 2 * PLANE_F);
        if (m == 2) a1[0] = rd_a(Sn, 1, 0);
        if (m == 4) a1[1] = rd_a(Sn, 1, 1);
        if (m == 6) a1[2] = rd_a(Sn, 1, 2);
      });
      term(A0, b0, [&](int m) __attribute__((always_inline)) {
        if (m == 0) a1[3] = rd_a(Sn, 1, 3);
        if (m == RQ) dma(rf, qb, Sc + OPA_F);
        if (m == 2) B2n[0] = rd_b(Sn, 2, 0);
        if (m == 4) B2n[1] = rd_b(Sn, 2, 1);
      });
      term(A0, b1, [&](int m) __attribute__((always_inline)) {
        if (m == 0) b0[0] = rd_b(Sn, 0, 0);
        if (m == RQ) dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
        if (m == 2) b0[1] = rd_b(Sn, 0, 1);
      });
      term(A0, B2, [&](int m) __attribute__((always_inline)) {
        if (m == 0) b1[0] = rd_b(Sn, 1, 0);
        if (m == RQ) dma(rf, qb + 2 * RG_F, Sc + OPA_F + 2 * PLANE_F);
        if (m == 2) b1[1] = rd_b(Sn, 1, 1);
      });
    } else {
      term(A0, b0, [&](int m) __attribute__((always_inline)) {
        if (m == 0) A0n[0] = rd_a(Sn, 0, 0);
        if (m == RQ) dma(rf, qa, Sc);
        if (m == 2) A0n[1] = rd_a(Sn, 0, 1);
        if (m == 4) A0n[2] = rd_a(Sn, 0, 2);
        if (m == 6) A0n[3] = rd_a(Sn, 0, 3);
      });
      term(A0, b1, [&](int m) __attribute__((always_inline)) {
        if (m == 0) b0[0] = rd_b(Sn, 0, 0);
        if (m == RQ) dma(rf, qb, Sc + OPA_F);
        if (m == 2) b0[1] = rd_b(Sn, 0, 1);
        if (m == RQ + 2) dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
        if (m == 4) B2n[0] = rd_b(Sn, 2, 0);
        if (m == 6) B2n[1] = rd_b(Sn, 2, 1);
      });
      term(A0, B2, [&](int m) __attribute__((always_inline)) {
        if (m == 0) b1[0] = rd_b(Sn, 1, 0);
        if (m == RQ) dma(rf, qb + 2 * RG_F, Sc + OPA_F + 2 * PLANE_F);
        if (m == 2) b1[1] = rd_b(Sn, 1, 1);
      });
    }
    cur = nxt;
  };
  auto run = [&](auto variant) __attribute__((always_inline)) {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(kt, a0x, b2x, a0y, b2y, variant);
      step(kt + 1, a0y, b2y, a0x, b2x, variant);
    }
    if (kt < nk) step(kt, a0x, b2x, a0y, b2y, variant);
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                    // the epilogue reuses the ring

  x_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot, q, lt);
}

// ---- two f16 planes x two f16 planes, three products ("h2", round 5) ----------------------------------------------------------------
// C = alpha . A . B^T for fp32 operands given as h2 images (csrc/x3_image.h: a S_a = hi + lo in IEEE half): the products hi hi,
// hi lo, lo hi accumulate in fp32 on v_mfma_f32_32x32x16_f16 -- the same pipe and rate as the bf16 form, HALF the matrix
// instructions of the six-product split, 4 instead of 6 bytes per operand element.  Error: 2^-21 |a b| per term (the dropped lo lo
// product 2^-22, the two representations 2^-23 each), i.e. three fp32 roundings; an fp32 FMA chain over K terms commits ~sqrt(K) of
// them, so for K >= 16 this is inside the error of the exact-fp32 MFMA kernel (tests/test_gpu_h2.py measures both against fp64).
// What it needs that the bf16 split does not: the operands' magnitudes (half has 5 exponent bits) -- alpha carries 1 / (S_a S_b), from
// the host for operands with known bounds (l2-normalised inputs, LSTM outputs) and / or from device words for gradients whose scale
// is measured on the device (dsa / dsb).  Use: products whose result is a SUM over the operand's rows (weight gradients) or whose
// operand rows are uniformly scaled; a per-row dynamic range (dx of vanishing time steps) stays on the bf16 split.
// Schedule: gemm_x3q_kernel's (three one-block stages, step kt + 3 requested into the stage step kt was read from, every LDS read and
// DMA request behind its own MFMA); per step 24 MFMAs, 12 fragment reads, 4 requests.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// PA = 1: the A operand is ONE half plane whose elements are exact in half -- the uint8 frames minus 128 -- : two products a b_hi + a b_lo
// (the uint8 layer-0 projection and weight gradient of the recurrent stack: "h1x2").
template <int PA>
__global__ __launch_bounds__(512) void gemm_h2q_kernel(const XGroup G) {
  constexpr int OPA_F = PA * PLANE_F;
  constexpr int H2_STAGE_F = (PA + 2) * PLANE_F;                   // A planes, then B hi, B lo
  extern __shared__ __attribute__((aligned(16))) float smem[];    // NST * H2_STAGE_F floats (96 KiB at PA = 2)
  int q, nparts, part, slot, lt;
  x_work_item(G, q, nparts, part, slot, lt);
  const XArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int wm = grp * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int kb0 = (int)((int64_t)g.KB * part / nparts), kb1 = (int)((int64_t)g.KB * (part + 1) / nparts);
  const int nk = kb1 - kb0;
  const float* pa = g.A + (int64_t)min(m0 / 32 + wave, (g.M + 31) / 32 - 1) * g.ska * (PA * RG_F) + lane * 4;
  const float* pb = g.B + (int64_t)min(n0 / 32 + wave, (g.N + 31) / 32 - 1) * g.skb * (2 * RG_F) + lane * 4;
  constexpr int DMA = PA + 2;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int pro = nk < 3 ? nk : 3;
  for (int s = 0; s < pro; ++s) { fill_op<PA>(pa, kb0 + s, smem + s * H2_STAGE_F, tid); fill_op<2>(pb, kb0 + s, smem + s * H2_STAGE_F + OPA_F, tid); }
  if (pro == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA) : "memory");
  else if (pro == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (wm + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  const int fb = OPA_F + (wn + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  auto rd_a = [&](const float* S, int p, int t) __attribute__((always_inline)) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(&S[fa + p * PLANE_F + t * 256]));
  };
  auto rd_b = [&](const float* S, int p, int t) __attribute__((always_inline)) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const float4*>(&S[fb + p * PLANE_F + t * 256]));
  };
  // hi planes of A and lo planes of B are held to the end of a step: two register sets in alternation (as a0 / b2 of gemm_x3q_kernel)
  f16x8 al[4], bh[2], ahx[4], blx[2], ahy[4], bly[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (PA == 2) al[t] = rd_a(smem, 1, t);
    ahx[t] = rd_a(smem, 0, t);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) { bh[t] = rd_b(smem, 0, t); blx[t] = rd_b(smem, 1, t); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int wbase = (tid & ~63) * 4;
  auto dma = [&](bool on, const float* src, float* dst) __attribute__((always_inline)) {
    if (on)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto term = [&](const f16x8 (&x)[4], const f16x8 (&y)[2], auto behind) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[m >> 1], y[m & 1], acc[m >> 1][m & 1], 0, 0, 0);
      behind(m);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  int cur = 0;
  // one step: lo hi, hi hi, hi lo of step kt from (al, bh, AH, BL); step kt + 1 read into (al, bh, AHn, BLn)
  auto step = [&](int kt, f16x8 (&AH)[4], f16x8 (&BL)[2], f16x8 (&AHn)[4], f16x8 (&BLn)[2], auto variant) __attribute__((always_inline)) {
    constexpr int V = decltype(variant)::value;
    constexpr int RQ = V == 0 ? 1 : 5;                             // the request slots of this wave group (odd: reads take the even ones)
    const int nxt = cur + 1 == NST ? 0 : cur + 1;
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool rf = kt + 3 < nk;
    const float* qa = pa + (int64_t)(kb0 + kt + 3) * (PA * RG_F);
    const float* qb = pb + (int64_t)(kb0 + kt + 3) * (2 * RG_F);
    float* Sc = smem + cur * H2_STAGE_F + wbase;
    const float* Sn = smem + nxt * H2_STAGE_F;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PA == 1) {
      term(AH, bh, [&](int m) __attribute__((always_inline)) {        // a b_hi; meanwhile the next step's A into the other set
        if (m == 0) AHn[0] = rd_a(Sn, 0, 0);
        if (m == RQ) dma(rf, qa, Sc);
        if (m == 2) AHn[1] = rd_a(Sn, 0, 1);
        if (m == 4) AHn[2] = rd_a(Sn, 0, 2);
        if (m == 6) AHn[3] = rd_a(Sn, 0, 3);
      });
      term(AH, BL, [&](int m) __attribute__((always_inline)) {        // a b_lo; B hi is free: the next step's, and B lo into the other set
        if (m == 0) bh[0] = rd_b(Sn, 0, 0);
        if (m == RQ) dma(rf, qb, Sc + OPA_F);
        if (m == 2) bh[1] = rd_b(Sn, 0, 1);
        if (m == RQ + 2) dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
        if (m == 4) BLn[0] = rd_b(Sn, 1, 0);
        if (m == 6) BLn[1] = rd_b(Sn, 1, 1);
      });
      cur = nxt;
      return;
    }
    term(al, bh, [&](int m) __attribute__((always_inline)) {          // lo hi; meanwhile the next step's A hi
      if (m == 0) AHn[0] = rd_a(Sn, 0, 0);
      if (m == RQ) dma(rf, qa, Sc);
      if (m == 2) AHn[1] = rd_a(Sn, 0, 1);
      if (m == 4) AHn[2] = rd_a(Sn, 0, 2);
      if (m == 6) AHn[3] = rd_a(Sn, 0, 3);
    });
    term(AH, bh, [&](int m) __attribute__((always_inline)) {          // hi hi; A lo is free: the next step's
      if (m == 0) al[0] = rd_a(Sn, 1, 0);
      if (m == RQ) dma(rf, qa + RG_F, Sc + PLANE_F);
      if (m == 2) al[1] = rd_a(Sn, 1, 1);
      if (m == RQ + 2) dma(rf, qb, Sc + OPA_F);
      if (m == 4) al[2] = rd_a(Sn, 1, 2);
      if (m == 6) al[3] = rd_a(Sn, 1, 3);
    });
    term(AH, BL, [&](int m) __attribute__((always_inline)) {          // hi lo; B hi is free: the next step's, and B lo into the other set
      if (m == 0) bh[0] = rd_b(Sn, 0, 0);
      if (m == RQ) dma(rf, qb + RG_F, Sc + OPA_F + PLANE_F);
      if (m == 2) bh[1] = rd_b(Sn, 0, 1);
      if (m == 4) BLn[0] = rd_b(Sn, 1, 0);
      if (m == 6) BLn[1] = rd_b(Sn, 1, 1);
    });
    cur = nxt;
  };
  auto run = [&](auto variant) __attribute__((always_inline)) {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      step(kt, ahx, blx, ahy, bly, variant);
      step(kt + 1, ahy, bly, ahx, blx, variant);
    }
    if (kt < nk) step(kt, ahx, blx, ahy, bly, variant);
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                    // the epilogue reuses the ring

  const XArgs ge = fold_device_scales(g);
  x_epilogue(G, ge, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot, q, lt);
}

// ---- one plane x one plane: the plain bf16 product on operand images ("b1") -------------------------------------------------------
// C = A . B^T for operands that ARE bfloat16 (--compute_dtype=bfloat16: BASELINE configs[4] and the bf16 variants), both given as
// ONE-plane images -- [rows / 32][K / 16][32 rows][2 halves][8] bf16, what yt8m_bf16_image writes straight from the fp32 source.
// Why images: the row-major kernels of gemm_bf16.hip are bound by L2 -> LDS operand delivery (~6.5 TB/s: 16 rows x 64 bytes, or 8
// rows x 128 bytes, per wave instruction; with the MFMAs removed they take as long as with them), while a wave
// instruction on an image moves 1 KiB of consecutive memory and four K blocks of a row group are 4 KiB in a row.
// Step = four K blocks (K = 64): 64 KiB per stage, two stages (the refill of a stage is issued block by block between the MFMA
// groups of the step after its last read), one barrier per step; per block and wave 6 ds_read_b128 feed 8 MFMAs.
constexpr int B1_KBS = 4;                                           // K blocks per step
constexpr int B1_STAGE_F = 2 * B1_KBS * PLANE_F;                    // A blocks, then B blocks (64 KiB)

__global__ __launch_bounds__(512) void gemm_b1_kernel(const XGroup G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // 2 * B1_STAGE_F floats (128 KiB)
  int q, nparts, part, slot, lt;
  x_work_item(G, q, nparts, part, slot, lt);
  const XArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int kb0 = (int)((int64_t)g.KB * part / nparts), kb1 = (int)((int64_t)g.KB * (part + 1) / nparts);
  const int nkb = kb1 - kb0, nst = (nkb + B1_KBS - 1) / B1_KBS;
  // this wave's 32-row group of either operand tile, at K block kb0 (groups beyond the matrix only feed outputs never stored)
  const float* pa = g.A + ((int64_t)min(m0 / 32 + wave, (g.M + 31) / 32 - 1) * g.ska + kb0) * RG_F + lane * 4;
  const float* pb = g.B + ((int64_t)min(n0 / 32 + wave, (g.N + 31) / 32 - 1) * g.skb + kb0) * RG_F + lane * 4;
  const int wbase = (tid & ~63) * 4;                               // wave-uniform LDS base of the wave's row group inside a block
  auto dma = [&](const float* src, float* dst) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // prologue: step 0
  for (int j = 0; j < B1_KBS && j < nkb; ++j) {
    dma(pa + (int64_t)j * RG_F, smem + j * PLANE_F + wbase);
    dma(pb + (int64_t)j * RG_F, smem + (B1_KBS + j) * PLANE_F + wbase);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (wm + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));        // + t * 256 floats per 32 rows, + block * PLANE_F
  const int fb = B1_KBS * PLANE_F + (wn + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  for (int st = 0; st < nst; ++st) {
    const float* S = smem + (st & 1) * B1_STAGE_F;
    float* N = smem + ((st + 1) & 1) * B1_STAGE_F + wbase;
    const int here = min(B1_KBS, nkb - st * B1_KBS);                 // K blocks of this step (the last one may be short)
    const int next = min(B1_KBS, nkb - (st + 1) * B1_KBS);           // ... of the next one (<= 0: none)
    const float* qa = pa + (int64_t)(st + 1) * B1_KBS * RG_F;
    const float* qb = pb + (int64_t)(st + 1) * B1_KBS * RG_F;
#pragma unroll
    for (int j = 0; j < B1_KBS; ++j) {
      if (j < here) {
        bf16x8 a[4], b[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fa + j * PLANE_F + t * 256]));
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fb + j * PLANE_F + t * 256]));
        if (j < next) {
          dma(qa + (int64_t)j * RG_F, N + j * PLANE_F);
          dma(qb + (int64_t)j * RG_F, N + (B1_KBS + j) * PLANE_F);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      } else if (j < next) {                                       // (a short step followed by blocks cannot happen; kept for safety)
        dma(qa + (int64_t)j * RG_F, N + j * PLANE_F);
        dma(qb + (int64_t)j * RG_F, N + (B1_KBS + j) * PLANE_F);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the next step landed; this wave's fragment reads are done
    __builtin_amdgcn_s_barrier();
  }
  x_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot, q, lt);
}

// ---- the same product as gemm_b1_kernel with a deeper request ring and fragments carried across the barrier (round 4) ------------------
// What the counters said about gemm_b1_kernel: its K step ends on vmcnt(0) for requests the wave issued one
// block earlier, so every step exposes most of an L2 / fabric round trip; and each block's MFMAs wait for fragment reads issued
// just before them.  Here LDS is a ring of eight 16 KiB slots (A block, B block), a step is TWO blocks, and:
//   * the requests of step s + 3 are issued in step s (into the slots step s - 1 just vacated), and the counted vmcnt at the end of
//     step s retires step s + 2 -- two steps of landing time, never vmcnt(0) in steady state;
//   * a step's blocks were therefore published one barrier EARLIER than they are read, so the fragments of block n + 1 are read
//     during block n's MFMAs, across the step boundary too: no wave waits on LDS after a barrier.
// (A phased variant -- two barriers per block, the two M-half wave groups one barrier apart, after the hardware guide's "eight
// phase" description -- measured 8-15 % SLOWER than gemm_b1_kernel on every head shape and on 8192^3: 921 vs 1096 TFLOP/s; removed.)
constexpr int BQ_SLOTS = 8;
constexpr int BQ_SLOT_F = 2 * PLANE_F;                               // A block (8 KiB), B block (8 KiB)

__global__ __launch_bounds__(512) void gemm_b1q_kernel(const XGroup G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // BQ_SLOTS * BQ_SLOT_F floats (128 KiB)
  int q, nparts, part, slot, lt;
  x_work_item(G, q, nparts, part, slot, lt);
  const XArgs& g = G.p[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, lt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int wm = grp * 128, wn = (wave & 3) * 64;
  const int li = lane & 31, lk = lane >> 5;
  const int kb0 = (int)((int64_t)g.KB * part / nparts), kb1 = (int)((int64_t)g.KB * (part + 1) / nparts);
  const int nkb = kb1 - kb0;
  const float* pa = g.A + ((int64_t)min(m0 / 32 + wave, (g.M + 31) / 32 - 1) * g.ska + kb0) * RG_F + lane * 4;
  const float* pb = g.B + ((int64_t)min(n0 / 32 + wave, (g.N + 31) / 32 - 1) * g.skb + kb0) * RG_F + lane * 4;
  const int wbase = (tid & ~63) * 4;
  auto dma = [&](const float* src, float* dst) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto request = [&](int n) __attribute__((always_inline)) {         // block n of both operands -> slot n & 7 (this wave's row groups)
    float* S = smem + (n & (BQ_SLOTS - 1)) * BQ_SLOT_F + wbase;
    dma(pa + (int64_t)n * RG_F, S);
    dma(pb + (int64_t)n * RG_F, S + PLANE_F);
  };
  const int fa = (wm + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  const int fb = PLANE_F + (wn + li) * 8 + 4 * (lk ^ ((li >> 3) & 1));
  auto frags = [&](int n, bf16x8 (&a)[4], bf16x8 (&b)[2]) __attribute__((always_inline)) {
    const float* S = smem + (n & (BQ_SLOTS - 1)) * BQ_SLOT_F;
#pragma unroll
    for (int t = 0; t < 2; ++t) b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fb + t * 256]));
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(&S[fa + t * 256]));
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // prologue: steps 0..2 requested; steps 0 and 1 retired by this wave, then by every wave
  for (int n = 0; n < 6 && n < nkb; ++n) request(n);
  if (nkb >= 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (nkb == 5) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 a0[4], b0[2], a1[4], b1[2];
  // One block = eight MFMAs with ONE other instruction behind each: the six fragment reads of the next block and the block's two
  // requests.  (Issued as a group, six reads or two requests outlast the 32 cycles the last-issued MFMA keeps the pipe busy, and
  // the two waves of a SIMD -- aligned by the barriers -- both sit in that gap at once: the counters showed 31 % of the pipe idle
  // even with the requests compiled out.)  The two wave groups take their requests in different slots (V = 0: behind MFMAs 6, 7;
  // V = 1: behind MFMAs 0, 1).  The reads are inline asm with hand-counted lgkmcnt: hipcc waits lgkmcnt(0) before the first MFMA
  // that uses ANY fragment of a set, i.e. for the read issued one instruction earlier.  LDS returns in order; fragment order
  // b0 a0 b1 a1 a2 a3, MFMA m uses a[m / 2], b[m % 2]: the waits below allow exactly the reads issued after the needed one.
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  auto frag1 = [&](int n, int t, bf16x8 (&a)[4], bf16x8 (&b)[2]) __attribute__((always_inline)) {   // t: b0 a0 b1 a1 a2 a3
    const uint32_t S = lds0 + (uint32_t)((n & (BQ_SLOTS - 1)) * BQ_SLOT_F) * 4u;
    if (t == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(b[0]) : "v"(S + (uint32_t)fb * 4u) : "memory");
    else if (t == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(S + (uint32_t)fa * 4u) : "memory");
    else if (t == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(b[1]) : "v"(S + (uint32_t)(fb + 256) * 4u) : "memory");
    else asm volatile("ds_read_b128 %0, %1" : "=v"(a[t - 2]) : "v"(S + (uint32_t)(fa + (t - 2) * 256) * 4u) : "memory");
  };
#pragma unroll
  for (int t = 0; t < 6; ++t) frag1(0, t, a0, b0);
  auto block = [&](bf16x8 (&ca)[4], bf16x8 (&cb)[2], bf16x8 (&na)[4], bf16x8 (&nb)[2], int nread, int nreq, auto variant, auto dma_on)
                   __attribute__((always_inline)) {
    constexpr int V = decltype(variant)::value;
    constexpr bool D = decltype(dma_on)::value;
    float* R = smem + (nreq & (BQ_SLOTS - 1)) * BQ_SLOT_F + wbase;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      // the fragment this MFMA is the first to use (index into b0 a0 b1 a1 a2 a3), and the reads of the NEXT set issued so far
      const int need = m == 0 ? 1 : m == 1 ? 2 : m == 2 ? 3 : m == 4 ? 4 : m == 6 ? 5 : -1;
      const int newer = V == 0 ? (m < 6 ? m : 6) : (m < 2 ? 0 : m - 2);
      if (need >= 0) {
        const int allow = (5 - need) + newer;
        if (m == 0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ca[0]), "+v"(cb[0]), "+v"(ca[1]), "+v"(cb[1]) : "n"(allow) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ca[m >> 1]), "+v"(cb[m & 1]), "+v"(ca[3]), "+v"(ca[2]) : "n"(allow) : "memory");
      }
      acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[m >> 1], cb[m & 1], acc[m >> 1][m & 1], 0, 0, 0);
      const int rd = V == 0 ? m : m - 2;                            // which fragment read sits behind this MFMA (0..5), if any
      if (rd >= 0 && rd < 6) frag1(nread, rd, na, nb);
      if (D) {
        const int rq = V == 0 ? m - 6 : m;                          // which request (0: A, 1: B), if any
        if (rq == 0) dma(pa + (int64_t)nreq * RG_F, R);
        if (rq == 1) dma(pb + (int64_t)nreq * RG_F, R + PLANE_F);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto run = [&](auto variant) __attribute__((always_inline)) {
    int n = 0;
    for (; n + 7 < nkb; n += 2) {                                   // steady state: every step requests step s + 3
      block(a0, b0, a1, b1, n + 1, n + 6, variant, std::true_type{});
      block(a1, b1, a0, b0, n + 2, n + 7, variant, std::true_type{});
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");              // step s + 2 retired by this wave; this step's four may fly
      __builtin_amdgcn_s_barrier();
    }
    if (n + 6 < nkb) request(n + 6);                                 // (odd K-block count: one block left to request)
    for (; n + 1 < nkb; n += 2) {                                   // last steps: nothing left to request
      block(a0, b0, a1, b1, n + 1, 0, variant, std::false_type{});
      block(a1, b1, a0, b0, n + 2, 0, variant, std::false_type{});  // (the read of block n + 2 past the end fetches an unused slot)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (n < nkb) {                                                   // odd tail block (its fragments are in a0 / b0)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(b0[0]), "+v"(b0[1])::"memory");
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[m >> 1], b0[m & 1], acc[m >> 1][m & 1], 0, 0, 0);
    }
  };
  if (grp == 0) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                      // the epilogue reuses the ring
  x_epilogue(G, g, acc, smem, m0, n0, wm, wn, lane, wave, li, lk, nparts, slot, q, lt);
}

// sums the S parts of every remainder tile in a fixed order (deterministic) and applies bias / accumulate.
// Footprint matters more than speed-of-light here: with no LDS and <= 32 VGPRs a workgroup of this kernel fits on a CU that a
// persistent recurrence occupies (768 threads x 160 VGPRs leave 32 per lane), so in the LSTM step it runs BESIDE the recurrences
// instead of waiting for a CU like the GEMM it follows -- but alone on its CU, where loads in flight are what it lives on: two
// positions x two parts are requested at a time (one load at a time made a fix-up 150-280 us beside two recurrences, 1.2 ms of the
// weight-gradient stream per headline step).
__global__ __launch_bounds__(256) void x3_fixup_kernel(const XGroup G) {
  const int ft = blockIdx.x >> 4, sixteenth = blockIdx.x & 15;      // 16 workgroups per tile, 16 rows each
  int q = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < G.nprob && ft >= G.fix_base[i]) q = i;
  const XArgs g = fold_device_scales(G.p[q]);
  const int rt = ft - G.fix_base[q], S = G.S[q];
  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, G.full[q] + rt, tm, tn);
  const int m0 = tm * TM, n0 = tn * TN;
  const float* base = G.ws + (int64_t)(G.slot_base[q] + rt) * (TM * TN);
  const int64_t pstride = (int64_t)G.rem[q] * (TM * TN);
  for (int e = sixteenth * (TM * TN / 16) + threadIdx.x * 4; e < (sixteenth + 1) * (TM * TN / 16); e += 2 * 256 * 4) {
    const int e1 = e + 256 * 4;                                      // (4096 floats per slice: both positions are inside it)
    float4 v0 = *reinterpret_cast<const float4*>(base + e), v1 = *reinterpret_cast<const float4*>(base + e1);
    int s = 1;
    for (; s + 1 < S; s += 2) {
      const float4 a0 = *reinterpret_cast<const float4*>(base + (int64_t)s * pstride + e);
      const float4 a1 = *reinterpret_cast<const float4*>(base + (int64_t)s * pstride + e1);
      const float4 b0 = *reinterpret_cast<const float4*>(base + (int64_t)(s + 1) * pstride + e);
      const float4 b1 = *reinterpret_cast<const float4*>(base + (int64_t)(s + 1) * pstride + e1);
      v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
      v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
      v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
      v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
    }
    if (s < S) {
      const float4 a0 = *reinterpret_cast<const float4*>(base + (int64_t)s * pstride + e);
      const float4 a1 = *reinterpret_cast<const float4*>(base + (int64_t)s * pstride + e1);
      v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
      v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
    }
    finish_store(g, v0, m0 + e / TN, n0 + (e % TN));
    finish_store(g, v1, m0 + e1 / TN, n0 + (e1 % TN));
  }
}

// ---- the split pass -----------------------------------------------------------------------------------------------------
// src [R, Cc] fp32 (row stride ld), 64 x 64 tiles through LDS; plain image: rows = R, K = Cc; trans image: rows = Cc, K = R.
// Either destination may be null.  scale multiplies every element before the split (1.0f: none).
// rowscale / trans_s (both or neither): a second transposed image whose element (c, r) is rowscale[r] * scale * src[r][c] -- the
// operand r (.) dz of the layer-0 weight gradient on uint8 frames -- from the same pass over src.
// NP = 1: the ONE-plane image of the bfloat16 roundings (operands of the b1 kernel: yt8m_bf16_image).
// colpart / colpart_s (either may be null): per-tile column sums of the (scaled) source -- row blockIdx.y of a [ceil(R / 64), Cc]
// matrix of partial sums, plain and rowscale-weighted -- so that the bias gradient colsum(dz) and the rank-1 remainder
// colsum(r (.) dz) of the recurrent layers ride on the pass that reads dz anyway instead of two more passes over it at the very end
// of the backward pass (a fixed-order sum of the partials follows: deterministic).
// NP = 2: h2 images (two IEEE-half planes, csrc/x3_image.h) of S_d . scale . src, S_d = the power of two that brings the max |src| held
// (as float bits) by the device word `dscale` into [2^13, 2^14) (NULL: 1).
template <int NP>
__global__ __launch_bounds__(256) void x3_split_kernel(const float* __restrict__ src, int64_t ld, int R, int Cc, float* __restrict__ plain,
                                                       float* __restrict__ trans, float scale, const float* __restrict__ rowscale,
                                                       float* __restrict__ trans_s, float* __restrict__ colpart = nullptr,
                                                       float* __restrict__ colpart_s = nullptr, const float* __restrict__ dscale = nullptr,
                                                       const float* __restrict__ rowS = nullptr, const unsigned* __restrict__ rowmaxw = nullptr,
                                                       float* __restrict__ rowinv = nullptr, unsigned long long* __restrict__ degraded = nullptr,
                                                       float drop_keep = 0.f, unsigned long long drop_seed = 0ull, long long drop_off = 0) {
  // drop_keep in (0, 1): the source is tf.nn.dropout(src, keep) -- element (r, c) is element drop_off + r Cc + c of the logical tensor
  // whose Philox stream (csrc/philox.h, key drop_seed) csrc/random.hip's yt8m_dropout_f32 draws: the image is that of the dropped
  // tensor, bit for bit, without the tensor ever being written (DropoutWrapper(input_keep_prob) inside the native recurrent stack).
  // degraded (NP = 2): the process-wide sticky counters of yt8m_h2_degraded -- [0] elements CLAMPED (|x S| beyond the largest half: the
  // operand outgrew its scale), [1] elements FLUSHED (x != 0 whose scaled value rounds to a zero half: more than 2^-38 below the scale's
  // maximum).  One wave-uniform test per 64 x 64 tile; atomics only when something was counted.
  // rowmaxw (NP = 2, plain image only; instead of rowS): max |src[r, :]| as float bits (measured by the producer of src: yt8m_lstm_persist_bwd_ex);
  // the row's power of two is derived here, the workgroups of the first tile column write its inverse to rowinv (the product's rowscale)
  // rowS (NP = 2, plain image only): row r of the source is scaled by rowS[r] -- one power of two PER ROW (yt8m_h2_rowscales): the
  // operand of a product whose rows must each keep their own precision (dx = dz . W^T of time steps whose gradients differ by decades)
  __shared__ float T[64][65];
  if (dscale) scale *= yt8m_x3::pow2_scale_for(__uint_as_float(reinterpret_cast<const unsigned*>(dscale)[0]), 14);
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  unsigned n_clamp = 0, n_flush = 0;
  {
    const int c4 = (t & 15) * 4, rr = t >> 4;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 16 * i;
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < R) {
        const float* p = src + (int64_t)(r0 + r) * ld + c0 + c4;
        if (vec && c0 + c4 + 3 < Cc) v = *reinterpret_cast<const float4*>(p);
        else {
          if (c0 + c4 + 0 < Cc) v.x = p[0];
          if (c0 + c4 + 1 < Cc) v.y = p[1];
          if (c0 + c4 + 2 < Cc) v.z = p[2];
          if (c0 + c4 + 3 < Cc) v.w = p[3];
        }
      }
      if (drop_keep > 0.f && r0 + r < R) {
        const long long e0 = drop_off + (long long)(r0 + r) * Cc + c0 + c4;
        float* vv = reinterpret_cast<float*>(&v);
        if ((e0 & 3) == 0) {                                // one Philox block = this float4
          const yt8m_rng::U4 rn = yt8m_rng::philox4x32_10((unsigned long long)(e0 >> 2), drop_seed);
          const unsigned rw[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = (drop_keep + yt8m_rng::u01(rw[e])) >= 1.0f ? __fdiv_rn(vv[e], drop_keep) : 0.0f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            vv[e] = (drop_keep + yt8m_rng::uniform_at((unsigned long long)(e0 + e), drop_seed)) >= 1.0f ? __fdiv_rn(vv[e], drop_keep) : 0.0f;
        }
      }
      float sr = (rowS && r0 + r < R) ? scale * rowS[r0 + r] : scale;
      if (rowmaxw && r0 + r < R) {
        const float pr = yt8m_x3::pow2_scale_for(__uint_as_float(rowmaxw[r0 + r]), 14);
        sr = scale * pr;
        if (rowinv && blockIdx.x == 0 && (t & 15) == 0) rowinv[r0 + r] = 1.0f / pr;
      }
      T[r][c4 + 0] = v.x * sr; T[r][c4 + 1] = v.y * sr; T[r][c4 + 2] = v.z * sr; T[r][c4 + 3] = v.w * sr;
#ifndef YT8M_H2_NO_COUNT
      if constexpr (NP == 2) {
        const float sv[4] = {v.x * sr, v.y * sr, v.z * sr, v.w * sr};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m = fabsf(sv[e]);
          n_clamp += (m > 65504.f && m < 3.0e38f) ? 1u : 0u;                 // (inf / nan are not the scale's fault)
          n_flush += (m != 0.f && m < 2.98023224e-8f) ? 1u : 0u;             // nonzero and < 2^-25: rounds to a zero half
        }
      }
#endif
    }
  }
  if constexpr (NP == 2) {
    if (degraded && __any((n_clamp | n_flush) != 0u)) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { n_clamp += __shfl_xor(n_clamp, o, 64); n_flush += __shfl_xor(n_flush, o, 64); }
      if ((t & 63) == 0) {                                  // 256 shards, one 128-byte line each: [shard][0] clamped, [shard][1] flushed
        unsigned long long* w = degraded + (size_t)((blockIdx.x + blockIdx.y * 7u + (unsigned)(t >> 6) * 61u) & 255u) * 16;
        if (n_clamp) atomicAdd(w, (unsigned long long)n_clamp);
        if (n_flush) atomicAdd(w + 1, (unsigned long long)n_flush);
      }
    }
  }
  __syncthreads();
  if ((colpart || colpart_s) && t < 128) {                          // threads 0-63: plain sums, 64-127: weighted sums (rows >= R are 0)
    const int c = t & 63;
    const bool weighted = t >= 64;
    float* dst = weighted ? colpart_s : colpart;
    if (dst && c0 + c < Cc) {
      float acc = 0.f;
      if (weighted) {
        for (int r = 0; r < 64; ++r) acc += (r0 + r < R ? rowscale[r0 + r] : 0.f) * T[r][c];
      } else {
        for (int r = 0; r < 64; ++r) acc += T[r][c];
      }
      if constexpr (NP == 2) acc *= 1.0f / scale;                    // h2: T holds scale . src; the scale is a power of two, so this
      dst[(int64_t)blockIdx.y * Cc + c0 + c] = acc;                  // is the sum of the unscaled column, bit for bit
    }
  }
  const int a = t & 63, blk = t >> 6;                               // row of the image within the tile, K block within the tile
  if (plain) {
    const int KB = (Cc + 15) >> 4;
    const int row = r0 + a, kb = (c0 >> 4) + blk;
    if (row < ((R + 31) & ~31) && kb < KB) {                        // rows of the last 32-row group beyond R: zeros
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = T[a][blk * 16 + j];
      if constexpr (NP == 2) yt8m_x3::store_block_h2(v, plain + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
      else store_block<NP>(v, plain + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
    }
  }
  if (trans || trans_s) {
    const int KB = (R + 15) >> 4;
    const int row = c0 + a, kb = (r0 >> 4) + blk;
    if (row < ((Cc + 31) & ~31) && kb < KB) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = T[blk * 16 + j][a];
      if (trans) {
        if constexpr (NP == 2) yt8m_x3::store_block_h2(v, trans + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
        else store_block<NP>(v, trans + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
      }
      if (trans_s) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int r = r0 + blk * 16 + j;
          v[j] *= r < R ? rowscale[r] : 0.f;
        }
        if constexpr (NP == 2) yt8m_x3::store_block_h2(v, trans_s + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
        else store_block<NP>(v, trans_s + ((int64_t)(row >> 5) * KB + kb) * (NP * RG_F), row);
      }
    }
  }
}

}  // namespace

using namespace yt8m;

// Sticky counters of the h2 split passes ("nothing at run time reports a clamp"): 256 shards of two 64-bit words per device, allocated on
// first use and never freed -- [0] elements clamped, [1] nonzero elements flushed to a zero half, summed over every h2 image the
// process has split on that device since the last reset.  A failed allocation turns the counting off (NULL), never the split.
namespace {
constexpr size_t H2_DEGRADED_BYTES = 256 * 128;           // 256 shards x one 128-byte line (a single pair of words serialised the
                                                          // atomics of every split workgroup: +1.8 ms on the headline step, measured)
unsigned long long* h2_degraded_words() {
  static std::mutex mu;
  static unsigned long long* words[64] = {};
  static bool tried[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tried[dev]) {
    tried[dev] = true;
    void* p = nullptr;
    if (hipMalloc(&p, H2_DEGRADED_BYTES) == hipSuccess && hipMemset(p, 0, H2_DEGRADED_BYTES) == hipSuccess) words[dev] = static_cast<unsigned long long*>(p);
    else (void)hipGetLastError();
  }
  return words[dev];
}
}  // namespace

// counts[0] = elements CLAMPED by an h2 split on the current device since the last reset (|x . S| beyond the largest half: an operand
// outgrew a static or stale scale -- results are degraded), counts[1] = nonzero elements FLUSHED to a zero half (more than 2^-38
// below their scale's maximum: their contribution is below one ulp of the product's largest terms).  Waits for `stream`;
// reset != 0 zeroes the words afterwards.  Covers yt8m_h2_split / _ex / _rows / _rowmax (every operand image of the h2 products); the
// persistent recurrences scale each producer's values by their own measured maxima and cannot clamp.
extern "C" int yt8m_h2_degraded(uint64_t* counts, int reset, yt8m_stream_t stream) {
  YT8M_REQUIRE(counts, YT8M_E_BADARG, "counts is NULL");
  counts[0] = counts[1] = 0;
  unsigned long long* w = h2_degraded_words();
  if (!w) return YT8M_OK;
  YT8M_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  static thread_local unsigned long long host[H2_DEGRADED_BYTES / 8];
  YT8M_HIP_CHECK(hipMemcpy(host, w, H2_DEGRADED_BYTES, hipMemcpyDeviceToHost));
  for (size_t sh = 0; sh < 256; ++sh) {
    counts[0] += host[sh * 16];
    counts[1] += host[sh * 16 + 1];
  }
  if (reset) YT8M_HIP_CHECK(hipMemset(w, 0, H2_DEGRADED_BYTES));
  return YT8M_OK;
}

extern "C" int64_t yt8m_x3_image_bytes(int64_t rows, int64_t K) { return ((rows + 31) / 32) * ((K + 15) / 16) * 3072; }

// fp32 src [R, C] -> x3 images.  plain: rows = R, K = C (operand used K-contiguous as stored); trans: rows = C, K = R (operand
// used transposed).  One pass over src feeds both (either may be NULL).
extern "C" int yt8m_x3_split(const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans,
                             yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans)) & 15) == 0, YT8M_E_BADARG,
               "x3 images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  if (plain) yt8m_wimg_note_demand(src, R, C, ld, 0, 3, scale);     // (only while the owner of a parameter arena records: wimg.hip)
  if (trans) yt8m_wimg_note_demand(src, R, C, ld, 1, 3, scale);
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<3>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  return launch_status("x3_split_kernel");
}

// fp32 src [R, C] -> ONE-plane images of its bfloat16 rounding (round to nearest even): plain [R rows, K = C] and / or trans
// [C rows, K = R]; yt8m_x3_image_bytes(rows, K) / 3 bytes each.  The operands of yt8m_gemm_b1_nt_grouped; replaces the row-major
// casts (yt8m_cast_f32_bf16) of the bf16 configuration at the same cost: one pass over the source.
extern "C" int yt8m_bf16_image(const float* src, int64_t R, int64_t C, int64_t ld, float scale, void* plain, void* trans,
                               yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans)) & 15) == 0, YT8M_E_BADARG,
               "images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  if (plain) yt8m_wimg_note_demand(src, R, C, ld, 0, 1, scale);     // (only while the owner of a parameter arena records: wimg.hip)
  if (trans) yt8m_wimg_note_demand(src, R, C, ld, 1, 1, scale);
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<1>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  return launch_status("x3_split_kernel");
}

// fp32 src [R, C] -> h2 images (two IEEE-half planes, yt8m_x3_image_bytes(rows, K) * 2 / 3 bytes each): plain [R rows, K = C] and / or
// trans [C rows, K = R] of S_d . scale . src (S_d from the absmax word `dscale`, NULL: 1).  The caller owns `scale`: |scale . src| must stay below 65504 (it is clamped),
// and the product's alpha carries its inverse.  colpart (may be NULL): per-64-row-tile column sums of the SCALED source, as
// yt8m_x3_split_colsum (divide by the scale to use them).
extern "C" int yt8m_h2_split(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* dscale, void* plain, void* trans,
                             float* colpart, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans)) & 15) == 0, YT8M_E_BADARG,
               "images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  if (!dscale) {
    if (plain) yt8m_wimg_note_demand(src, R, C, ld, 0, 2, scale);
    if (trans) yt8m_wimg_note_demand(src, R, C, ld, 1, 2, scale);
  }
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<2>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, (const float*)nullptr, (float*)nullptr, colpart, (float*)nullptr, dscale,
                     (const float*)nullptr, (const unsigned*)nullptr, (float*)nullptr, h2_degraded_words());
  return launch_status("x3_split_kernel<2>");
}

// yt8m_h2_split of tf.nn.dropout(src, keep_prob): the h2 image(s) of the dropped tensor -- x / keep_prob where the Philox stream of
// (seed, offset + row * C + col) keeps the element, else 0: exactly what yt8m_dropout_f32(src, ., R * C, keep_prob, seed, offset)
// followed by yt8m_h2_split would produce -- in one pass, without the dropped copy (tf.contrib.rnn.DropoutWrapper(cell,
// input_keep_prob), W/all_frame_models/lstm_memory_model.py:36-44, inside yt8m_lstm_stack_fwd / _bwd).  ld == C.
extern "C" int yt8m_h2_split_dropout(const float* src, int64_t R, int64_t C, float scale, const float* dscale, void* plain, void* trans,
                                     float keep_prob, uint64_t seed, int64_t offset, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && (plain || trans), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f && offset >= 0, YT8M_E_BADARG, "keep_prob in (0, 1], offset >= 0");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans)) & 15) == 0, YT8M_E_BADARG,
               "images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<2>, grid, dim3(256), 0, as_stream(stream), src, C, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, dscale,
                     (const float*)nullptr, (const unsigned*)nullptr, (float*)nullptr, h2_degraded_words(),
                     keep_prob < 1.f ? keep_prob : 0.f, (unsigned long long)seed, (long long)offset);
  return launch_status("x3_split_kernel<2>");
}

// yt8m_h2_split with the outputs of yt8m_x3_split_colsum: trans_scaled = the h2 image of (diag(rowscale) . S_d . scale . src)^T, the
// per-tile column sums plain and rowscale-weighted (of the UNscaled source); any output may be NULL (rowscale comes with its two).
extern "C" int yt8m_h2_split_ex(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const void* dscale, const float* rowscale,
                                void* plain, void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans || trans_scaled), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE((rowscale != nullptr) == (trans_scaled != nullptr || colpart_scaled != nullptr), YT8M_E_BADARG,
               "rowscale comes with trans_scaled / colpart_scaled");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans) | reinterpret_cast<uintptr_t>(trans_scaled)) & 15) == 0,
               YT8M_E_BADARG, "images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<2>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, rowscale, static_cast<float*>(trans_scaled), colpart, colpart_scaled,
                     static_cast<const float*>(dscale), (const float*)nullptr, (const unsigned*)nullptr, (float*)nullptr, h2_degraded_words());
  return launch_status("x3_split_kernel<2>");
}

namespace {
// max |src| -> the bit pattern of a non-negative float orders like the float: one atomicMax per workgroup, order independent
__global__ __launch_bounds__(256) void h2_absmax_kernel(const float* __restrict__ src, int64_t ld, int R, int Cc, unsigned* __restrict__ word) {
  __shared__ float red[4];
  float m = 0.f;
  const int64_t n = (int64_t)R * Cc;
  if (ld == Cc && (Cc & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
      const float4 x = *reinterpret_cast<const float4*>(src + i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
      const int64_t r = i / Cc, c = i - r * Cc;
      m = fmaxf(m, fabsf(src[r * ld + c]));
    }
  }
  m = block_max_256(m, red);
  if (threadIdx.x == 0 && m > 0.f && m < 3.0e38f) atomicMax(word, __float_as_uint(m));
}
}  // namespace

namespace {
// one wave per row: S[r] = the power of two with max |src[r, :]| S in [2^13, 2^14) (1 for an all-zero row), inv[r] = 1 / S[r]
__global__ __launch_bounds__(256) void h2_rowscale_kernel(const float* __restrict__ src, int64_t ld, int R, int Cc, float* __restrict__ S,
                                                          float* __restrict__ inv) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* p = src + (int64_t)row * ld;
  float m = 0.f;
  if ((ld & 3) == 0 && (Cc & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int c = lane * 4; c < Cc; c += 256) {
      const float4 x = *reinterpret_cast<const float4*>(p + c);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));
    }
  } else {
    for (int c = lane; c < Cc; c += 64) m = fmaxf(m, fabsf(p[c]));
  }
  m = wave_max(m);
  if (lane == 0) {
    const float s = yt8m_x3::pow2_scale_for(m, 14);
    S[row] = s;
    inv[row] = 1.0f / s;
  }
}
}  // namespace

// Per-ROW scales of an h2 operand: S[r] = the power of two with max |src[r, :]| S[r] in [2^13, 2^14), inv[r] = 1 / S[r] (what the
// product's rowscale takes); yt8m_h2_split_rows writes the plain h2 image [R rows, K = C] of diag(S) . src.
extern "C" int yt8m_h2_rowscales(const float* src, int64_t R, int64_t C, int64_t ld, float* S, float* inv, yt8m_stream_t stream) {
  YT8M_REQUIRE(src && S && inv && R >= 1 && C >= 1 && ld >= C && R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "bad arguments");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(h2_rowscale_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, S, inv);
  return launch_status("h2_rowscale_kernel");
}
extern "C" int yt8m_h2_split_rows(const float* src, int64_t R, int64_t C, int64_t ld, const float* S, void* plain, yt8m_stream_t stream) {
  YT8M_REQUIRE(src && S && plain && R >= 1 && C >= 1 && ld >= C && R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "bad arguments");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(plain) & 15) == 0, YT8M_E_BADARG, "images must be 16-byte aligned");
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<2>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     (float*)nullptr, 1.0f, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, S,
                     (const unsigned*)nullptr, (float*)nullptr, h2_degraded_words());
  return launch_status("x3_split_kernel<2>");
}
// yt8m_h2_rowscales + yt8m_h2_split_rows in ONE pass when the row maxima are already known: rowmax[r] = max |src[r, :]| as float bits
// (yt8m_lstm_persist_bwd_ex measures them while it writes dz); writes the plain h2 image of diag(S) . src and inv[r] = 1 / S[r].
extern "C" int yt8m_h2_split_rowmax(const float* src, int64_t R, int64_t C, int64_t ld, const void* rowmax, float* inv, void* plain,
                                    yt8m_stream_t stream) {
  YT8M_REQUIRE(src && rowmax && inv && plain && R >= 1 && C >= 1 && ld >= C && R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "bad arguments");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(plain) & 15) == 0, YT8M_E_BADARG, "images must be 16-byte aligned");
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<2>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     (float*)nullptr, 1.0f, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, static_cast<const unsigned*>(rowmax), inv, h2_degraded_words());
  return launch_status("x3_split_kernel<2>");
}
// C[M,N] (+)= alpha . rowscale[m] / (S_a S_b) . A . B^T (+ bias): yt8m_gemm_h2_nt_grouped for one product with a per-row factor
// (rowscale = inv of yt8m_h2_rowscales when A was split by rows; dsa / dsb = absmax words or NULL; ska / skb as the x3 forms).
extern "C" int yt8m_gemm_h2_nt_ex(int64_t M, int64_t N, int64_t K, const void* A2, int64_t ska, const void* B2, int64_t skb, float* C,
                                  int64_t ldc, const float* bias, float alpha, const void* dsa, const void* dsb, const float* rowscale,
                                  float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream);

// max |src| into a device word (float bits, atomicMax: order independent; the word must be zero before, several calls may share it):
// what yt8m_h2_split (dscale) and yt8m_gemm_h2_nt_grouped (dsa / dsb) turn into the image's scale S = 2^(14 - exponent) and its inverse.
extern "C" int yt8m_h2_absmax(const float* src, int64_t R, int64_t C, int64_t ld, void* word, yt8m_stream_t stream) {
  YT8M_REQUIRE(src && word && R >= 1 && C >= 1 && ld >= C && R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "bad arguments");
  const int64_t n = R * C;
  const unsigned blocks = (unsigned)std::min<int64_t>(2048, (n + 4095) / 4096);
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(h2_absmax_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<unsigned*>(word));
  return launch_status("h2_absmax_kernel");
}

// The same pass with a third output: trans_scaled = the x3 image of (diag(rowscale) . scale . src)^T ([C rows, K = R]); any of the
// three images may be NULL (rowscale and trans_scaled come together).
extern "C" int yt8m_x3_split_ex(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                                void* trans, void* trans_scaled, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans || trans_scaled), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE((rowscale != nullptr) == (trans_scaled != nullptr), YT8M_E_BADARG, "rowscale and trans_scaled come together");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans) | reinterpret_cast<uintptr_t>(trans_scaled)) & 15) == 0,
               YT8M_E_BADARG, "x3 images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<3>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, rowscale, static_cast<float*>(trans_scaled));
  return launch_status("x3_split_kernel");
}

// yt8m_x3_split_ex plus per-tile column sums from the same pass: colpart / colpart_scaled (either may be NULL) receive rows
// [0, ceil(R / 64)) of a [*, C] matrix of partial sums over 64-row tiles of `src` (x scale; the scaled one weighted by rowscale).
// A caller that splits a tall matrix in several row ranges (the recurrent stack's time parts, each a multiple of 64 rows) points
// colpart at the range's first tile row and finishes with ONE yt8m_colsum_f32 over the whole partial matrix.
extern "C" int yt8m_x3_split_colsum(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                                    void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans || trans_scaled), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE((rowscale != nullptr) == (trans_scaled != nullptr || colpart_scaled != nullptr), YT8M_E_BADARG,
               "rowscale comes with trans_scaled / colpart_scaled");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans) | reinterpret_cast<uintptr_t>(trans_scaled)) & 15) == 0,
               YT8M_E_BADARG, "x3 images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<3>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, rowscale, static_cast<float*>(trans_scaled), colpart, colpart_scaled);
  return launch_status("x3_split_kernel");
}

// The one-plane (bf16 rounding) form of yt8m_x3_split_colsum: plain / trans / trans_scaled are ONE-plane images (operands of
// yt8m_gemm_b1_nt_grouped / _ex), the column sums are those of the fp32 source.  The recurrent stack in bf16-operand mode.
extern "C" int yt8m_bf16_image_colsum(const float* src, int64_t R, int64_t C, int64_t ld, float scale, const float* rowscale, void* plain,
                                      void* trans, void* trans_scaled, float* colpart, float* colpart_scaled, yt8m_stream_t stream) {
  YT8M_REQUIRE(R >= 0 && C >= 0 && ld >= C && (plain || trans || trans_scaled), YT8M_E_BADARG, "bad split arguments");
  YT8M_REQUIRE((rowscale != nullptr) == (trans_scaled != nullptr || colpart_scaled != nullptr), YT8M_E_BADARG,
               "rowscale comes with trans_scaled / colpart_scaled");
  YT8M_REQUIRE(R < (1LL << 31) && C < (1LL << 31), YT8M_E_BADARG, "matrix too large");
  YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(plain) | reinterpret_cast<uintptr_t>(trans) | reinterpret_cast<uintptr_t>(trans_scaled)) & 15) == 0,
               YT8M_E_BADARG, "images must be 16-byte aligned");
  if (R == 0 || C == 0) return YT8M_OK;
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
  YT8M_REQUIRE(grid.y < 65536, YT8M_E_BADARG, "too many rows for one split launch");
  ProfScope prof(F_ELEMENTWISE, as_stream(stream));
  hipLaunchKernelGGL(x3_split_kernel<1>, grid, dim3(256), 0, as_stream(stream), src, ld, (int)R, (int)C, static_cast<float*>(plain),
                     static_cast<float*>(trans), scale, rowscale, static_cast<float*>(trans_scaled), colpart, colpart_scaled);
  return launch_status("x3_split_kernel");
}

namespace {
// Arrival counters of split tiles: one block of 64 Ki words per device, zeroed once (every completed tile leaves its counter at
// zero again), handed out to launches as a ring -- two launches can only share a counter if more than 64 Ki split tiles lie between
// them, i.e. never while the first one is still running.
constexpr uint32_t CNT_CAP = 1u << 16;
// yt8m_x3_set_combine: per-thread override of where the K parts of the following image-GEMM launches are summed
// (0 = the process default: separate fix-up pass unless YT8M_X3_FUSED_COMBINE=1; 1 = inside the launch; 2 = separate pass).
thread_local int g_combine_mode = 0;
// yt8m_x3_set_schedule: per-thread choice of the main-loop schedule of the following image-GEMM launches (0 = the process default:
// the interleaved kernels gemm_x3q_kernel / gemm_b1q_kernel unless YT8M_X3_PIPE=0 / YT8M_B1_PIPE=0; 1 = interleaved; 2 = the
// round-3 kernels).  Same products in the same order per accumulator: bit-identical results (tests/test_gpu_round4.py).
thread_local int g_schedule_mode = 0;
struct TileCounters {
  std::mutex mu;
  unsigned* base[16] = {nullptr};
  bool failed[16] = {false};
  uint32_t next[16] = {0};
  unsigned* take(int n) {
    // Measured (profiles/r4_sched_knobs.md): correct and bitwise equal, but NOT faster than the separate pass where it was meant to
    // pay -- 23.67 against 23.36 ms per headline step, 1.00 / 1.01 ms configs[1], 5.68 / 5.64 configs[2].  The likely reason (not
    // isolated by a counter): a 256-thread fix-up workgroup needs no LDS and 20 VGPRs, so it fits on CUs the persistent recurrences
    // occupy (they leave ~32 VGPRs per lane and ~7 KB of LDS) and runs beside them, while a GEMM workgroup that stays to combine
    // keeps its 144 KB of LDS -- and the recurrence launch that is waiting for that CU.  Opt-in: YT8M_X3_FUSED_COMBINE=1.
    static const bool off = getenv("YT8M_X3_FUSED_COMBINE") == nullptr || atoi(getenv("YT8M_X3_FUSED_COMBINE")) == 0;
    int dev = 0;
    const bool fused = g_combine_mode == 1 || (g_combine_mode == 0 && !off);
    if (!fused || n <= 0 || (uint32_t)n > CNT_CAP || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (failed[dev]) return nullptr;
    if (!base[dev]) {
      unsigned* p = nullptr;
      if (hipMalloc(reinterpret_cast<void**>(&p), CNT_CAP * sizeof(unsigned)) != hipSuccess ||
          hipMemset(p, 0, CNT_CAP * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError();
        failed[dev] = true;                                        // the separate fix-up pass still works
        return nullptr;
      }
      base[dev] = p;
    }
    if (next[dev] + (uint32_t)n > CNT_CAP) next[dev] = 0;
    unsigned* r = base[dev] + next[dev];
    next[dev] += (uint32_t)n;
    return r;
  }
};
TileCounters g_cnt;

// PA: 3 / 1 = planes of the A image of the bf16 split kernels, 0 = the one-plane bf16 kernels, 2 = the h2 kernel (two f16 planes each),
// 4 = the h2 kernel with a ONE-plane exact A operand ("h1x2")
template <int PA>
int x3_launch(int nprob, const yt8m_gemm_problem* probs, const float* rscale, const float* cs, float cs_scale, float alpha,
              void* workspace, int64_t workspace_bytes, yt8m_stream_t stream, const float* const* dsa = nullptr,
              const float* const* dsb = nullptr, const float* alphas = nullptr, unsigned c_bf16_mask = 0) {
  XGroup G;
  G.nprob = 0;
  // one 144 KiB workgroup per CU.  YT8M_X3_SLOTS (tuning aid): the CU count the K-part choice assumes -- beside a half-chip
  // persistent recurrence only 128 CUs take GEMM workgroups, and fewer K parts mean fewer slabs for the fix-up pass to sum.
  static const int SLOTS_ENV = getenv("YT8M_X3_SLOTS") ? atoi(getenv("YT8M_X3_SLOTS")) : 0;
  const int SLOTS = SLOTS_ENV >= 32 && SLOTS_ENV <= 256 ? SLOTS_ENV : 256;
  const int64_t per_part = (int64_t)TM * TN * sizeof(float);
  int64_t nfull = 0, slots = 0, fix = 0;
  for (int i = 0; i < nprob; ++i) {
    const yt8m_gemm_problem& q = probs[i];
    YT8M_REQUIRE(q.M >= 0 && q.N >= 0 && q.K >= 1 && q.ldc >= q.N, YT8M_E_BADARG, "bad GEMM problem");
    YT8M_REQUIRE(q.beta == 0.f || q.beta == 1.f, YT8M_E_BADARG, "beta must be 0 or 1");
    if (q.M == 0 || q.N == 0) continue;
    YT8M_REQUIRE(q.A && q.B && q.C, YT8M_E_BADARG, "null operand");
    YT8M_REQUIRE((((uintptr_t)q.A | (uintptr_t)q.B) & 15) == 0, YT8M_E_BADARG, "x3 images must be 16-byte aligned");
    XArgs g;
    g.A = static_cast<const float*>(q.A); g.B = static_cast<const float*>(q.B); g.C = q.C; g.bias = q.bias;
    g.ldc = q.ldc;
    g.M = (int)q.M; g.N = (int)q.N; g.KB = (int)((q.K + 15) / 16);
    // lda / ldb of an x3 problem: K blocks between the 32-row groups of the image (0: the image holds exactly this K range)
    YT8M_REQUIRE(q.lda >= 0 && q.ldb >= 0 && q.lda < (1LL << 30) && q.ldb < (1LL << 30) && (q.lda == 0 || q.lda >= g.KB) &&
                 (q.ldb == 0 || q.ldb >= g.KB), YT8M_E_BADARG, "bad image K-block stride");
    YT8M_REQUIRE((q.lda <= g.KB && q.ldb <= g.KB) || (q.K % 16) == 0, YT8M_E_SHAPE, "a K range of a larger image must be a multiple of 16");
    g.ska = q.lda ? (int)q.lda : g.KB; g.skb = q.ldb ? (int)q.ldb : g.KB;
    g.tiles_m = (int)((q.M + TM - 1) / TM); g.tiles_n = (int)((q.N + TN - 1) / TN);
    g.accumulate = q.beta != 0.f;
    g.rscale = rscale; g.cs = cs; g.cs_scale = cs_scale; g.alpha = alphas ? alphas[i] : alpha;
    g.dsa = dsa ? dsa[i] : nullptr; g.dsb = dsb ? dsb[i] : nullptr;
    g.c_bf16 = (int)((c_bf16_mask >> i) & 1u);
    YT8M_REQUIRE(!g.c_bf16 || (q.beta == 0.f && (q.ldc & 3) == 0 && ((uintptr_t)q.C & 15) == 0), YT8M_E_BADARG,
                 "a bf16 output takes beta = 0, ldc % 4 == 0 and a 16-byte aligned C");
    YT8M_REQUIRE(!(g.dsa || g.dsb || g.alpha != 1.0f) || (q.N % 4) == 0, YT8M_E_SHAPE, "a scaled product needs N % 4 == 0");
    // the last, partial round of this problem alone: S K-parts per tile; cost in K-steps = rounds x (steps per part + ramp)
    // + the fixup pass
    const int64_t T = (int64_t)g.tiles_m * g.tiles_n;
    const int full = (int)(T / SLOTS) * SLOTS, rem = (int)(T - full);
    int S = 1;
    if (workspace && rem > 0) {
      double best = 1e30;
      for (int c = 1; c <= 16; ++c) {                              // > 8 parts only pay for a handful of tiles with a very long K
        if (c > 1 && (g.KB / c < (PA == 0 ? 32 : 8) || (slots + (int64_t)rem * c) * per_part > workspace_bytes)) break;
        const int rounds = (rem * c + SLOTS - 1) / SLOTS;
        const double cost = rounds * ((double)g.KB / c + 10.0) + (c > 1 ? 4.0 + 0.065 * rem * c : 0.0);
        if (cost < best * 0.98) { best = cost; S = c; }
      }
    }
    const int k = G.nprob;
    G.p[k] = g;
    G.S[k] = S;
    G.full[k] = S > 1 ? full : (int)T;
    G.rem[k] = S > 1 ? rem : 0;
    G.full_base[k] = (int)nfull;
    G.part_base[k] = (int)slots;
    G.slot_base[k] = (int)slots;
    G.fix_base[k] = (int)fix;
    nfull += G.full[k];
    if (S > 1) { slots += (int64_t)rem * S; fix += rem; }
    ++G.nprob;
  }
  if (G.nprob == 0) return YT8M_OK;
  YT8M_REQUIRE(nfull + slots < (1LL << 30), YT8M_E_SHAPE, "too many tiles for one launch");
  for (int i = G.nprob; i <= 4; ++i) { G.full_base[i] = (int)nfull; G.part_base[i] = (int)slots; G.fix_base[i] = (int)fix; }
  for (int i = G.nprob; i < 4; ++i) { G.p[i] = G.p[0]; G.S[i] = 1; G.slot_base[i] = 0; G.full[i] = 0; G.rem[i] = 0; }
  G.ws = static_cast<float*>(workspace);
  G.cnt = fix > 0 ? g_cnt.take((int)fix) : nullptr;
  const int64_t grid = nfull + slots;
  constexpr int LDS_BYTES = PA == 0 ? 2 * B1_STAGE_F * (int)sizeof(float)
                            : PA == 2 ? NST * 4 * PLANE_F * (int)sizeof(float)
                            : PA == 4 ? NST * 3 * PLANE_F * (int)sizeof(float) : NST * (PA + 3) * PLANE_F * (int)sizeof(float);
  static DeviceOnce lds_once;                                      // per device (a process-wide flag broke cuda:1)
  if constexpr (PA == 0) YT8M_HIP_CHECK(lds_once.lds(reinterpret_cast<const void*>(gemm_b1_kernel), LDS_BYTES));
  else if constexpr (PA == 2) YT8M_HIP_CHECK(lds_once.lds(reinterpret_cast<const void*>(gemm_h2q_kernel<2>), LDS_BYTES));
  else if constexpr (PA == 4) YT8M_HIP_CHECK(lds_once.lds(reinterpret_cast<const void*>(gemm_h2q_kernel<1>), LDS_BYTES));
  else YT8M_HIP_CHECK(lds_once.lds(reinterpret_cast<const void*>(gemm_x3_kernel<(PA == 1) ? 1 : 3>), LDS_BYTES));
  double fl = 0.0;
  for (int i = 0; i < nprob; ++i) fl += 2.0 * (double)probs[i].M * (double)probs[i].N * (double)probs[i].K;
  ProfScope prof(PA == 0 ? F_GEMM : (PA == 1 ? F_GEMM_X1X3 : (PA == 2 ? F_GEMM_H2 : (PA == 4 ? F_GEMM_H1X2 : F_GEMM_X3))), as_stream(stream), fl);
  // YT8M_B1_PIPE=0: the round-3 kernel (two 64 KiB stages, one barrier per four blocks) instead of gemm_b1q_kernel
  static const bool piped_env = getenv("YT8M_B1_PIPE") == nullptr || atoi(getenv("YT8M_B1_PIPE")) != 0;
  const bool piped = g_schedule_mode == 1 || (g_schedule_mode == 0 && piped_env);
  if constexpr (PA == 0) {
    if (piped) {
      static DeviceOnce lds_once_q;
      YT8M_HIP_CHECK(lds_once_q.lds(reinterpret_cast<const void*>(gemm_b1q_kernel), BQ_SLOTS * BQ_SLOT_F * (int)sizeof(float)));
      hipLaunchKernelGGL(gemm_b1q_kernel, dim3((unsigned)grid), dim3(512), BQ_SLOTS * BQ_SLOT_F * (int)sizeof(float), as_stream(stream), G);
    } else {
      hipLaunchKernelGGL(gemm_b1_kernel, dim3((unsigned)grid), dim3(512), LDS_BYTES, as_stream(stream), G);
    }
  }
  else if constexpr (PA == 2) {
    hipLaunchKernelGGL(gemm_h2q_kernel<2>, dim3((unsigned)grid), dim3(512), LDS_BYTES, as_stream(stream), G);
  }
  else if constexpr (PA == 4) {
    hipLaunchKernelGGL(gemm_h2q_kernel<1>, dim3((unsigned)grid), dim3(512), LDS_BYTES, as_stream(stream), G);
  }
  else {
    // YT8M_X3_PIPE=0: the round-3 kernel (reads and requests issued in groups between the products)
    static const bool xpiped_env = getenv("YT8M_X3_PIPE") == nullptr || atoi(getenv("YT8M_X3_PIPE")) != 0;
    const bool xpiped = g_schedule_mode == 1 || (g_schedule_mode == 0 && xpiped_env);
    if (xpiped) {
      static DeviceOnce lds_once_xq;
      YT8M_HIP_CHECK(lds_once_xq.lds(reinterpret_cast<const void*>(gemm_x3q_kernel<(PA == 1) ? 1 : 3>), LDS_BYTES));
      hipLaunchKernelGGL(gemm_x3q_kernel<(PA == 1) ? 1 : 3>, dim3((unsigned)grid), dim3(512), LDS_BYTES, as_stream(stream), G);
    } else {
      hipLaunchKernelGGL(gemm_x3_kernel<(PA == 1) ? 1 : 3>, dim3((unsigned)grid), dim3(512), LDS_BYTES, as_stream(stream), G);
    }
  }
  if (fix > 0 && !G.cnt) hipLaunchKernelGGL(x3_fixup_kernel, dim3((unsigned)fix * 16), dim3(256), 0, as_stream(stream), G);
  return launch_status("gemm_x3_kernel");
}
}  // namespace

extern "C" int yt8m_x3_set_combine(int mode) {
  YT8M_REQUIRE(mode >= 0 && mode <= 2, YT8M_E_BADARG, "mode must be 0 (default), 1 (inside the launch) or 2 (separate pass)");
  g_combine_mode = mode;
  return YT8M_OK;
}

extern "C" int yt8m_x3_set_schedule(int mode) {
  YT8M_REQUIRE(mode >= 0 && mode <= 2, YT8M_E_BADARG, "mode must be 0 (default), 1 (interleaved kernels) or 2 (round-3 kernels)");
  g_schedule_mode = mode;
  return YT8M_OK;
}

// C[M,N] (+)= A . B^T (+ bias) from the x3 images of A ([M rows, K]) and B ([N rows, K]); yt8m_gemm_problem.A / .B are the
// images, lda / ldb are ignored, K is the logical K (the images are padded to a multiple of 16).  Up to four problems.
extern "C" int yt8m_gemm_x3_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes,
                                       yt8m_stream_t stream) {
  YT8M_REQUIRE(nprob >= 1 && nprob <= 4 && probs, YT8M_E_BADARG, "1..4 problems per launch");
  return x3_launch<3>(nprob, probs, nullptr, nullptr, 0.f, 1.0f, workspace, workspace_bytes, stream);
}

// C[M,N] (+)= A . B^T (+ bias) for bf16 operands given as ONE-plane images (yt8m_bf16_image); yt8m_gemm_problem as in
// yt8m_gemm_x3_nt_grouped (A / B = images, lda / ldb = K-block strides or 0).  Up to four problems per launch.
extern "C" int yt8m_gemm_b1_nt_grouped(int nprob, const yt8m_gemm_problem* probs, void* workspace, int64_t workspace_bytes,
                                       yt8m_stream_t stream) {
  YT8M_REQUIRE(nprob >= 1 && nprob <= 4 && probs, YT8M_E_BADARG, "1..4 problems per launch");
  return x3_launch<0>(nprob, probs, nullptr, nullptr, 0.f, 1.0f, workspace, workspace_bytes, stream);
}

// yt8m_gemm_b1_nt_grouped whose outputs flagged in `c_bf16_mask` (bit i = problem i) are bf16 matrices: C points at bf16 elements, ldc
// counts them, beta must be 0.  Accumulation stays fp32; the value is rounded to nearest even once, after bias.  The logits of the MoE
// heads in the bf16 configuration (configs[4]): half the bytes written here and read by the mixing passes.
extern "C" int yt8m_gemm_b1_nt_grouped_bf16c(int nprob, const yt8m_gemm_problem* probs, unsigned c_bf16_mask, void* workspace,
                                             int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(nprob >= 1 && nprob <= 4 && probs, YT8M_E_BADARG, "1..4 problems per launch");
  return x3_launch<0>(nprob, probs, nullptr, nullptr, 0.f, 1.0f, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, c_bf16_mask);
}

// C[M,N] (+)= alpha_i / (S_a S_b) . A . B^T (+ bias) from the h2 images of A ([M rows, K]) and B ([N rows, K]) (yt8m_h2_split):
// three f16 MFMA products per element pair, fp32 accumulation.  alpha = 1 / (host-side scales of the images); dsa / dsb (arrays of
// nprob device pointers, entries or the arrays themselves may be NULL) point at the absmax words (yt8m_h2_absmax) of operands whose
// scale S was chosen on the device.  yt8m_gemm_problem as in yt8m_gemm_x3_nt_grouped.
extern "C" int yt8m_gemm_h2_nt_grouped(int nprob, const yt8m_gemm_problem* probs, const float* alphas, const float* const* dsa,
                                       const float* const* dsb, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(nprob >= 1 && nprob <= 4 && probs, YT8M_E_BADARG, "1..4 problems per launch");
  return x3_launch<2>(nprob, probs, nullptr, nullptr, 0.f, 1.0f, workspace, workspace_bytes, stream, dsa, dsb, alphas);
}

extern "C" int yt8m_gemm_h2_nt_ex(int64_t M, int64_t N, int64_t K, const void* A2, int64_t ska, const void* B2, int64_t skb, float* C,
                                  int64_t ldc, const float* bias, float alpha, const void* dsa, const void* dsb, const float* rowscale,
                                  float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE((N % 4) == 0, YT8M_E_SHAPE, "the scaled epilogue needs N % 4 == 0");
  yt8m_gemm_problem p;
  p.M = M; p.N = N; p.K = K; p.A = A2; p.lda = ska; p.B = B2; p.ldb = skb; p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
  const float* wa = static_cast<const float*>(dsa);
  const float* wb = static_cast<const float*>(dsb);
  return x3_launch<2>(1, &p, rowscale, nullptr, 0.f, alpha, workspace, workspace_bytes, stream, dsa ? &wa : nullptr, dsb ? &wb : nullptr);
}

// C[M,N] (+)= alpha . rowscale[m] . (A1 . B^T / S_b + colsum_scale . colsum[n]) + bias[n]: A1 a ONE-plane HALF image whose elements are
// exact (q - 128: yt8m_u8_frames_image_f16 / _t_f16), B an h2 image under the device-chosen scale S_b of the absmax word dsb (NULL:
// S_b = 1): two f16 products per element pair -- the uint8 layer-0 projection and weight gradient (yt8m_gemm_x1x3_nt_ex's role at 2/3
// of its matrix time).  ska / skb / beta / rowscale / colsum as there.
extern "C" int yt8m_gemm_h1x2_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B2, int64_t skb, float* C,
                                    int64_t ldc, const float* bias, float alpha, const void* dsb, const float* rowscale, const float* colsum,
                                    float colsum_scale, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE((N % 4) == 0, YT8M_E_SHAPE, "the scaled epilogue needs N % 4 == 0");
  YT8M_REQUIRE(!colsum || (reinterpret_cast<uintptr_t>(colsum) & 15) == 0, YT8M_E_SHAPE, "colsum must be 16-byte aligned");
  yt8m_gemm_problem p;
  p.M = M; p.N = N; p.K = K; p.A = A1; p.lda = ska; p.B = B2; p.ldb = skb; p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
  const float* dw = static_cast<const float*>(dsb);
  return x3_launch<4>(1, &p, rowscale, colsum, colsum_scale, alpha, workspace, workspace_bytes, stream, nullptr, dsb ? &dw : nullptr);
}

// The uint8 input projection: C[M,N] = rowscale[m] * (A . B^T + colsum_scale * colsum[n]) + bias[n], A a ONE-plane image (elements
// exact in bf16: q - 128, written by yt8m_u8_frames_image), B the three-plane x3 image of (4/255) W^T.
extern "C" int yt8m_gemm_x1x3_nt(int64_t M, int64_t N, int64_t K, const void* A1, const void* B3, float* C, int64_t ldc,
                                 const float* bias, const float* rowscale, const float* colsum, float colsum_scale, void* workspace,
                                 int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE((rowscale == nullptr) == (colsum == nullptr), YT8M_E_BADARG, "rowscale and colsum come together");
  YT8M_REQUIRE(!rowscale || ((N % 4) == 0 && (reinterpret_cast<uintptr_t>(colsum) & 15) == 0), YT8M_E_SHAPE,
               "the affine epilogue needs N % 4 == 0 and a 16-byte aligned colsum");
  yt8m_gemm_problem p;
  p.M = M; p.N = N; p.K = K; p.A = A1; p.lda = 0; p.B = B3; p.ldb = 0; p.C = C; p.ldc = ldc; p.bias = bias; p.beta = 0.f;
  return x3_launch<1>(1, &p, rowscale, colsum, colsum_scale, 1.0f, workspace, workspace_bytes, stream);
}

// General form: C[M,N] (+)= alpha * rowscale[m] * (A1 . B3^T + colsum_scale * colsum[n]) + bias[n]; rowscale / colsum may be NULL
// independently; ska / skb = K blocks (of 16) between the 32-row groups of either image (0: the image is exactly K wide), so a
// product may read a K range of whole-sequence images (A1 / B3 then point at the first block of the range; K % 16 == 0).
// The layer-0 weight gradient of the recurrent models on raw uint8 frames is this product:
//   dW_x[d,n] = sum_m x[m,d] dz[m,n],  x = r (.) (alpha (q - 128) + beta)
//             = alpha ( (q - 128)^T . (r (.) dz) + (beta / alpha) colsum(r (.) dz) )
// with A1 = the transposed one-plane image of q - 128 (yt8m_u8_frames_image_t), B3 = the transposed x3 image of r (.) dz
// (yt8m_x3_split_ex), beta = 1 to accumulate over time chunks (W/readers.py:178-187 folded into W/train.py's gradient).
extern "C" int yt8m_gemm_x1x3_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B3, int64_t skb, float* C,
                                    int64_t ldc, const float* bias, float alpha, const float* rowscale, const float* colsum,
                                    float colsum_scale, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(!(rowscale || colsum || alpha != 1.0f) || (N % 4) == 0, YT8M_E_SHAPE, "the affine epilogue needs N % 4 == 0");
  YT8M_REQUIRE(!colsum || (reinterpret_cast<uintptr_t>(colsum) & 15) == 0, YT8M_E_SHAPE, "colsum must be 16-byte aligned");
  yt8m_gemm_problem p;
  p.M = M; p.N = N; p.K = K; p.A = A1; p.lda = ska; p.B = B3; p.ldb = skb; p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
  return x3_launch<1>(1, &p, rowscale, colsum, colsum_scale, alpha, workspace, workspace_bytes, stream);
}

// One-plane x one-plane product with the affine epilogue of yt8m_gemm_x1x3_nt_ex: C (+)= alpha * rowscale[m] * (A1 . B1^T +
// colsum_scale * colsum[n]) + bias[n].  The uint8 layer-0 projection and weight gradient of the recurrent stack in bf16-operand
// mode ((q - 128) is exact in bf16; the other operand is the bf16 rounding of (4/255) W^T or of r (.) dz).
extern "C" int yt8m_gemm_b1_nt_ex(int64_t M, int64_t N, int64_t K, const void* A1, int64_t ska, const void* B1, int64_t skb, float* C,
                                  int64_t ldc, const float* bias, float alpha, const float* rowscale, const float* colsum,
                                  float colsum_scale, float beta, void* workspace, int64_t workspace_bytes, yt8m_stream_t stream) {
  YT8M_REQUIRE(!(rowscale || colsum || alpha != 1.0f) || (N % 4) == 0, YT8M_E_SHAPE, "the affine epilogue needs N % 4 == 0");
  YT8M_REQUIRE(!colsum || (reinterpret_cast<uintptr_t>(colsum) & 15) == 0, YT8M_E_SHAPE, "colsum must be 16-byte aligned");
  yt8m_gemm_problem p;
  p.M = M; p.N = N; p.K = K; p.A = A1; p.lda = ska; p.B = B1; p.ldb = skb; p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
  return x3_launch<0>(1, &p, rowscale, colsum, colsum_scale, alpha, workspace, workspace_bytes, stream);
}
