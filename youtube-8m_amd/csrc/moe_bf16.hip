// moe_bf16.hip -- MoE mixing backward that leaves the logits' gradients directly as bf16 GEMM operands (--compute_dtype=bfloat16).
//
// In the bf16 configuration dL/dZ is consumed three times: K-contiguous by dx = dZ . W^T, row-transposed by dW = x^T . dZ, and
// column-summed for the expert-bias gradient.  The fp32 path writes dZ in place (4 B/element), a dual cast re-reads it and
// writes both bf16 layouts (4 + 2 + 2), colsum re-reads dZ_e: 15.6 B per logit element.  Here one pass reads Z (and dL/dp or the
// labels), forms dL/dZ in registers (as moe_mix_bwd_kernel / moe_mix_xent_bwd_kernel) and writes
//   dZg_b [B, V(M+1)]  dZg_t [V(M+1), B]   dZe_b [B, VM]   dZe_t [VM, B]      bf16, row pitches given by the caller
//   be_part [B/64, VM]                     fp32 column sums of dZ_e over each 64-row block (fixed order -> deterministic)
// = 4 + 2 + 2 B per element.  M = 2 (the reference's default and every BASELINE configuration); other M keep the fp32 path.
// Workgroup = 64 rows x 64 labels; a thread owns (row, 16 consecutive labels): 48 + 32 contiguous logits in, 96 + 64 contiguous
// bytes out; the transposed copies go through an LDS tile [column][64 rows] and leave as 128-byte rows.
#include "common.h"

namespace {

constexpr int M2 = 2;
constexpr int TR = 64, TL = 64;                    // rows x labels per workgroup
constexpr int GC = TL * (M2 + 1), EC = TL * M2;    // 192 gate columns, 128 expert columns per tile
constexpr int PITCH = TR + 8;                      // bf16 elements per LDS column (+8: 16-byte aligned rows, spreads banks)

__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// MODE 0: d = dp[b, l].  MODE 1: d from the labels (uint8 / float), CrossEntropyLoss fused (eps, dscale * up_dev[0]).
// IMG: the four outputs are ONE-plane operand images of the b1 kernel (csrc/gemm_x3.hip: [rows / 32][K / 16][32 rows][2 halves][8]
// bf16, half h of row r in slot h ^ ((r >> 3) & 1)) instead of row-major matrices; the *_ld arguments then carry the K-block
// counts of the images (plain: ceil(columns / 16), transposed: ceil(B / 16)).  Same 16-byte pieces, other addresses; the zero
// padding of the last K block is written too (the GEMM reads whole blocks).
__device__ __forceinline__ int64_t img_piece(int64_t row, int64_t k8, int64_t KB) {     // uint16 offset of (row, 8 values from K = k8)
  const int r = (int)(row & 31);
  return ((row >> 5) * KB + (k8 >> 4)) * 512 + r * 16 + ((int)((k8 >> 3) & 1) ^ ((r >> 3) & 1)) * 8;
}
// ZT: float logits, or unsigned short = bf16 logits (round 6: written by yt8m_gemm_b1_nt_grouped_bf16c; V % 4 == 0)
template <int MODE, typename LT, bool IMG, typename ZT = float>
__global__ __launch_bounds__(256) void moe_mix_bwd_bf16_kernel(const ZT* __restrict__ Zg, const ZT* __restrict__ Ze,
                                                               const float* __restrict__ dp, const LT* __restrict__ y,
                                                               int64_t B, int64_t V, float eps, float dscale,
                                                               const float* __restrict__ up_dev,
                                                               unsigned short* __restrict__ gb, int64_t gb_ld,
                                                               unsigned short* __restrict__ gt, int64_t gt_ld,
                                                               unsigned short* __restrict__ eb, int64_t eb_ld,
                                                               unsigned short* __restrict__ et, int64_t et_ld,
                                                               float* __restrict__ be_part) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[(GC + EC) * PITCH];   // 320 x 72 x 2 B = 45 KiB
  const int tid = threadIdx.x;
  // A thread owns (row, 4 consecutive labels) of four 16-row groups: 12 gate + 8 expert logits = 3 + 2 float4 whose lane stride
  // (48 / 32 bytes) keeps every cache line a wave instruction touches inside the next two instructions -- with 16 labels per
  // thread (192-byte lane stride) a line was revisited over 8 instructions and 64 lines per wave: the 2 x 4 waves of a CU ran
  // out of L1 and the read side alone took 273 of the kernel's 530 us (tools/mixb_bench.py).
  const int q = tid & 15, rs = tid >> 4;                              // label quad of the 64-label tile, row of a 16-row group
  const int64_t l0 = (int64_t)blockIdx.x * TL + q * 4;
  const int64_t b0 = (int64_t)blockIdx.y * TR + rs;
  if (MODE == 1 && up_dev) dscale *= up_dev[0];
  float g[4][12], e[4][8], dd[4][4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t b = b0 + it * 16;
#pragma unroll
    for (int k = 0; k < 12; ++k) g[it][k] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) e[it][k] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) dd[it][k] = 0.f;
    if (b >= B || l0 >= V) continue;
    const ZT* zg = Zg + b * V * 3 + l0 * 3;
    const ZT* ze = Ze + b * V * 2 + l0 * 2;
    const bool full = l0 + 4 <= V;
    if constexpr (sizeof(ZT) == 2) {
      if (full && (reinterpret_cast<uintptr_t>(zg) & 7) == 0 && (reinterpret_cast<uintptr_t>(ze) & 15) == 0) {
        const uint2* gp = reinterpret_cast<const uint2*>(zg);
        const uint2 g0 = gp[0], g1 = gp[1], g2 = gp[2];
        const uint4 ev = *reinterpret_cast<const uint4*>(ze);
        const unsigned gw[6] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y};
        const unsigned ew[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int k = 0; k < 6; ++k) { g[it][2 * k] = __uint_as_float(gw[k] << 16); g[it][2 * k + 1] = __uint_as_float(gw[k] & 0xffff0000u); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[it][2 * k] = __uint_as_float(ew[k] << 16); e[it][2 * k + 1] = __uint_as_float(ew[k] & 0xffff0000u); }
      } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) if (l0 + k / 3 < V) g[it][k] = bf2f((unsigned short)zg[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (l0 + k / 2 < V) e[it][k] = bf2f((unsigned short)ze[k]);
      }
    } else
    if (full && ((reinterpret_cast<uintptr_t>(zg) | reinterpret_cast<uintptr_t>(ze)) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(zg) + 4 * k);
        g[it][4 * k] = v.x; g[it][4 * k + 1] = v.y; g[it][4 * k + 2] = v.z; g[it][4 * k + 3] = v.w;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ze) + 4 * k);
        e[it][4 * k] = v.x; e[it][4 * k + 1] = v.y; e[it][4 * k + 2] = v.z; e[it][4 * k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) if (l0 + k / 3 < V) g[it][k] = (float)zg[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) if (l0 + k / 2 < V) e[it][k] = (float)ze[k];
    }
    if (MODE == 0) {
      const float* d = dp + b * V + l0;
      if (full && (reinterpret_cast<uintptr_t>(d) & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(d);
        dd[it][0] = v.x; dd[it][1] = v.y; dd[it][2] = v.z; dd[it][3] = v.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (l0 + k < V) dd[it][k] = d[k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (l0 + k < V) dd[it][k] = (float)y[b * V + l0 + k];
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t b = b0 + it * 16;
    const int row = rs + it * 16;
    unsigned short og[12], oe[8];
#pragma unroll
    for (int k = 0; k < 12; ++k) og[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) oe[k] = 0;
    if (b < B) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (l0 + j < V) {
          const float g0 = g[it][3 * j], g1 = g[it][3 * j + 1], g2 = g[it][3 * j + 2];
          const float mx = fmaxf(g0, fmaxf(g1, g2));
          const float x0 = __expf(g0 - mx), x1 = __expf(g1 - mx), x2 = __expf(g2 - mx);
          const float inv = 1.0f / (x0 + x1 + x2);
          const float s0 = x0 * inv, s1 = x1 * inv, s2 = x2 * inv;
          const float e0 = 1.0f / (1.0f + __expf(-e[it][2 * j])), e1 = 1.0f / (1.0f + __expf(-e[it][2 * j + 1]));
          const float pv = s0 * e0 + s1 * e1;
          float d;
          if (MODE == 0) {
            d = dd[it][j];
          } else {
            const float yv = dd[it][j];
            d = -(yv / (pv + eps) - (1.0f - yv) / (1.0f - pv + eps)) * dscale;
          }
          og[3 * j] = f2bf(d * s0 * (e0 - pv));
          og[3 * j + 1] = f2bf(d * s1 * (e1 - pv));
          og[3 * j + 2] = f2bf(d * s2 * (0.f - pv));
          oe[2 * j] = f2bf(d * s0 * e0 * (1.0f - e0));
          oe[2 * j + 1] = f2bf(d * s1 * e1 * (1.0f - e1));
        }
      }
    }
    // plain outputs in 16-byte pieces of 8 columns.  Gate: a thread holds 12 columns = one and a half pieces; the even lane of
    // a pair writes its first 8 and the piece made of its last 4 + the odd lane's first 4, the odd lane writes its last 8.
    unsigned int w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = og[2 * k] | ((unsigned)og[2 * k + 1] << 16);
    const unsigned int n0 = __shfl_down(w[0], 1, 64), n1 = __shfl_down(w[1], 1, 64);     // the odd neighbour's first 4 columns
    const bool odd = (q & 1) != 0;
    const int64_t cg = (int64_t)blockIdx.x * GC + (odd ? q * 12 + 4 : q * 12);              // first column of this lane's first piece
    const int64_t ce = (int64_t)blockIdx.x * EC + q * 8;
    uint4 pg0, pg1, pe0;
    if (odd) { pg0 = uint4{w[2], w[3], w[4], w[5]}; pg1 = pg0; }
    else { pg0 = uint4{w[0], w[1], w[2], w[3]}; pg1 = uint4{w[4], w[5], n0, n1}; }
    pe0.x = oe[0] | ((unsigned)oe[1] << 16); pe0.y = oe[2] | ((unsigned)oe[3] << 16);
    pe0.z = oe[4] | ((unsigned)oe[5] << 16); pe0.w = oe[6] | ((unsigned)oe[7] << 16);
    if (b < B) {
      if (IMG) {
        if (cg < gb_ld * 16) *reinterpret_cast<uint4*>(gb + img_piece(b, cg, gb_ld)) = pg0;
        if (!odd && cg + 8 < gb_ld * 16) *reinterpret_cast<uint4*>(gb + img_piece(b, cg + 8, gb_ld)) = pg1;
        if (ce < eb_ld * 16) *reinterpret_cast<uint4*>(eb + img_piece(b, ce, eb_ld)) = pe0;
      } else {
        // row-major: whole pieces where they fit into the row (columns up to the pitch are the caller's padding), else by element
        unsigned short* rg = gb + b * gb_ld;
        unsigned short* re = eb + b * eb_ld;
        const bool al = ((reinterpret_cast<uintptr_t>(rg) | reinterpret_cast<uintptr_t>(re)) & 15) == 0;
        auto put = [&](unsigned short* rowp, int64_t c, int64_t ncols, const uint4& v) {
          if (c >= ncols) return;
          if (al && c + 8 <= ncols) { *reinterpret_cast<uint4*>(rowp + c) = v; return; }
          const unsigned int u[4] = {v.x, v.y, v.z, v.w};
          for (int k = 0; k < 8 && c + k < ncols; ++k) rowp[c + k] = (unsigned short)(u[k >> 1] >> (16 * (k & 1)));
        };
        put(rg, cg, V * 3, pg0);
        if (!odd) put(rg, cg + 8, V * 3, pg1);
        put(re, ce, V * 2, pe0);
      }
    }
    // transposed image: tile[column][row]; rows beyond B / labels beyond V hold zeros
#pragma unroll
    for (int k = 0; k < 12; ++k) tile[(q * 12 + k) * PITCH + row] = og[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[(GC + q * 8 + k) * PITCH + row] = oe[k];
  }
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.y * TR;
  const bool rows_full = r0 + TR <= B;
  for (int idx = tid; idx < (GC + EC) * 8; idx += 256) {      // 16-byte pieces: 8 per column
    const int col = idx >> 3, piece = idx & 7;
    const bool gate = col < GC;
    const int64_t gcol = gate ? (int64_t)blockIdx.x * GC + col : (int64_t)blockIdx.x * EC + (col - GC);
    if (gcol >= (gate ? V * 3 : V * 2)) continue;
    const unsigned short* src = tile + col * PITCH + piece * 8;
    if (IMG) {                                                   // rows of the transposed images = logit columns, K = batch rows
      if (r0 + piece * 8 < (gate ? gt_ld : et_ld) * 16)
        *reinterpret_cast<uint4*>((gate ? gt : et) + img_piece(gcol, r0 + piece * 8, gate ? gt_ld : et_ld)) = *reinterpret_cast<const uint4*>(src);
      continue;
    }
    unsigned short* dst = (gate ? gt + gcol * gt_ld : et + gcol * et_ld) + r0 + piece * 8;
    if (rows_full && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
    } else {
      for (int k = 0; k < 8; ++k)
        if (r0 + piece * 8 + k < B) dst[k] = src[k];
    }
  }
  // expert-bias partials: column sums of the (bf16-rounded) dZ_e over this 64-row block, fixed order
  if (be_part && tid < EC) {
    const int64_t gcol = (int64_t)blockIdx.x * EC + tid;
    if (gcol < V * 2) {
      float s = 0.f;
      const unsigned short* src = tile + (GC + tid) * PITCH;
#pragma unroll 8
      for (int k = 0; k < TR; ++k) s += bf2f(src[k]);
      be_part[(int64_t)blockIdx.y * V * 2 + gcol] = s;
    }
  }
}

}  // namespace

using namespace yt8m;

extern "C" int64_t yt8m_moe_mix_bwd_bf16_partial_rows(int64_t B) { return (B + TR - 1) / TR; }

namespace {
int mix_bwd_bf16_launch(bool img, const void* Zgv, const void* Zev, bool z16, const float* dp, const void* labels, int label_dtype, int64_t B, int64_t V,
                        int M, float eps, float dscale, const float* upstream_dev, void* dZg_b, int64_t gb_ld, void* dZg_t, int64_t gt_ld,
                        void* dZe_b, int64_t eb_ld, void* dZe_t, int64_t et_ld, float* be_part, yt8m_stream_t stream) {
  YT8M_REQUIRE(M == 2, YT8M_E_BADARG, "the bf16 mixing backward is built for num_mixtures == 2");
  YT8M_REQUIRE(B >= 0 && V >= 0, YT8M_E_SHAPE, "negative dimension");
  if (B * V == 0) return YT8M_OK;
  YT8M_REQUIRE(Zgv && Zev && (dp || labels) && dZg_b && dZg_t && dZe_b && dZe_t, YT8M_E_BADARG, "null operand");
  const float* Zg = static_cast<const float*>(Zgv);
  const float* Ze = static_cast<const float*>(Zev);
  const unsigned short* Zg16 = static_cast<const unsigned short*>(Zgv);
  const unsigned short* Ze16 = static_cast<const unsigned short*>(Zev);
  YT8M_REQUIRE(!(dp && labels), YT8M_E_BADARG, "give either dp or labels");
  if (img) {
    YT8M_REQUIRE(gb_ld == (V * 3 + 15) / 16 && eb_ld == (V * 2 + 15) / 16 && gt_ld == (B + 15) / 16 && et_ld == (B + 15) / 16, YT8M_E_SHAPE,
                 "image mode: pass the K-block counts of the four images");
    YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(dZg_b) | reinterpret_cast<uintptr_t>(dZg_t) | reinterpret_cast<uintptr_t>(dZe_b) |
                   reinterpret_cast<uintptr_t>(dZe_t)) & 15) == 0, YT8M_E_BADARG, "images must be 16-byte aligned");
  } else {
    YT8M_REQUIRE(gb_ld >= V * 3 && eb_ld >= V * 2 && gt_ld >= B && et_ld >= B, YT8M_E_SHAPE, "leading dimension too small");
  }
  YT8M_REQUIRE((B + TR - 1) / TR <= 65535, YT8M_E_SHAPE, "too many rows");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const dim3 grid((unsigned)((V + TL - 1) / TL), (unsigned)((B + TR - 1) / TR));
  unsigned short *gb = static_cast<unsigned short*>(dZg_b), *gt = static_cast<unsigned short*>(dZg_t);
  unsigned short *eb = static_cast<unsigned short*>(dZe_b), *et = static_cast<unsigned short*>(dZe_t);
#define YT8M_MIX_LAUNCH(MODE, LT, IMGV, DP, LAB)                                                                                   \
  do {                                                                                                                            \
    if (z16)                                                                                                                      \
      hipLaunchKernelGGL((moe_mix_bwd_bf16_kernel<MODE, LT, IMGV, unsigned short>), grid, dim3(256), 0, s, Zg16, Ze16, DP, LAB, B, V, eps,  \
                         dscale, upstream_dev, gb, gb_ld, gt, gt_ld, eb, eb_ld, et, et_ld, be_part);                              \
    else                                                                                                                          \
      hipLaunchKernelGGL((moe_mix_bwd_bf16_kernel<MODE, LT, IMGV, float>), grid, dim3(256), 0, s, Zg, Ze, DP, LAB, B, V, eps, dscale,       \
                         upstream_dev, gb, gb_ld, gt, gt_ld, eb, eb_ld, et, et_ld, be_part);                                      \
  } while (0)
  if (dp) {
    if (img) YT8M_MIX_LAUNCH(0, uint8_t, true, dp, (const uint8_t*)nullptr);
    else YT8M_MIX_LAUNCH(0, uint8_t, false, dp, (const uint8_t*)nullptr);
  } else if (label_dtype == 0) {
    if (img) YT8M_MIX_LAUNCH(1, uint8_t, true, (const float*)nullptr, static_cast<const uint8_t*>(labels));
    else YT8M_MIX_LAUNCH(1, uint8_t, false, (const float*)nullptr, static_cast<const uint8_t*>(labels));
  } else {
    YT8M_REQUIRE(label_dtype == 1, YT8M_E_BADARG, "label_dtype must be 0 (uint8) or 1 (float32)");
    if (img) YT8M_MIX_LAUNCH(1, float, true, (const float*)nullptr, static_cast<const float*>(labels));
    else YT8M_MIX_LAUNCH(1, float, false, (const float*)nullptr, static_cast<const float*>(labels));
  }
#undef YT8M_MIX_LAUNCH
  return launch_status("moe_mix_bwd_bf16_kernel");
}
}  // namespace

extern "C" int yt8m_moe_mix_bwd_bf16(const float* Zg, const float* Ze, const float* dp, const void* labels, int label_dtype,
                                     int64_t B, int64_t V, int M, float eps, float dscale, const float* upstream_dev, void* dZg_b,
                                     int64_t gb_ld, void* dZg_t, int64_t gt_ld, void* dZe_b, int64_t eb_ld, void* dZe_t,
                                     int64_t et_ld, float* be_part, yt8m_stream_t stream) {
  return mix_bwd_bf16_launch(false, Zg, Ze, false, dp, labels, label_dtype, B, V, M, eps, dscale, upstream_dev, dZg_b, gb_ld, dZg_t, gt_ld, dZe_b,
                             eb_ld, dZe_t, et_ld, be_part, stream);
}

// The same pass with the four outputs as ONE-plane operand images of yt8m_gemm_b1_nt_grouped (plain: rows = B, K = V (M+1) / V M;
// transposed: rows = V (M+1) / V M, K = B); the *_kb arguments are the images' K-block counts (ceil(K / 16)).
extern "C" int yt8m_moe_mix_bwd_bf16_images(const float* Zg, const float* Ze, const float* dp, const void* labels, int label_dtype,
                                            int64_t B, int64_t V, int M, float eps, float dscale, const float* upstream_dev,
                                            void* dZg_img, int64_t g_kb, void* dZg_t_img, int64_t gt_kb, void* dZe_img, int64_t e_kb,
                                            void* dZe_t_img, int64_t et_kb, float* be_part, yt8m_stream_t stream) {
  return mix_bwd_bf16_launch(true, Zg, Ze, false, dp, labels, label_dtype, B, V, M, eps, dscale, upstream_dev, dZg_img, g_kb, dZg_t_img, gt_kb,
                             dZe_img, e_kb, dZe_t_img, et_kb, be_part, stream);
}

// yt8m_moe_mix_bwd_bf16_images on bf16 logits (Zg [B, 3V], Ze [B, 2V] as written by yt8m_gemm_b1_nt_grouped_bf16c): 10 instead of 20 bytes
// read per label.
extern "C" int yt8m_moe_mix_bwd_bf16_images_z16(const void* Zg, const void* Ze, const float* dp, const void* labels, int label_dtype,
                                                int64_t B, int64_t V, int M, float eps, float dscale, const float* upstream_dev,
                                                void* dZg_img, int64_t g_kb, void* dZg_t_img, int64_t gt_kb, void* dZe_img, int64_t e_kb,
                                                void* dZe_t_img, int64_t et_kb, float* be_part, yt8m_stream_t stream) {
  return mix_bwd_bf16_launch(true, Zg, Ze, true, dp, labels, label_dtype, B, V, M, eps, dscale, upstream_dev, dZg_img, g_kb, dZg_t_img, gt_kb,
                             dZe_img, e_kb, dZe_t_img, et_kb, be_part, stream);
}
