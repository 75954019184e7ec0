// optim.hip -- optimiser slice of build_graph as two multi-tensor passes over one flat fp32 arena.
//   W/train.py:435-466 (final_loss = label_loss + sum l2*0.5*|W|^2 -> compute_gradients -> clip -> apply)
//   W/utils.py:164-174 (per-tensor tf.clip_by_norm), tf.train.AdamOptimizer (TF-1 form).
// HBM-bound: sqnorm reads w,g (8 B/param); adam reads w,m,v,g and writes w,m,v (28 B/param).  Every
// chunk is <= 4096 consecutive floats of ONE tensor starting on a 16-byte boundary, so a 256-thread
// workgroup streams it as 4 float4 per lane.  Reductions are fixed-order => bitwise reproducible, which
// keeps data-parallel ranks identical after the gradient all-reduce.
#include "common.h"
#include "x3_image.h"

namespace {

// chunk size contract: <= 4096 floats (4 x float4 per lane of a 256-thread workgroup)

__global__ __launch_bounds__(256) void sqnorm_chunk_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                           const int4* __restrict__ chunks, const float* __restrict__ l2,
                                                           float gscale, float* __restrict__ partial) {
  __shared__ float red[4];
  const int4 ch = chunks[blockIdx.x];
  const float l2c = l2[ch.z];
  const float* wp = w + ch.x;
  const float* gp = g + ch.x;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = k * 1024 + threadIdx.x * 4;
    if (i + 3 < ch.y) {
      const float4 gv = *reinterpret_cast<const float4*>(gp + i);
      const float4 wv = *reinterpret_cast<const float4*>(wp + i);
      const float a = fmaf(l2c, wv.x, gv.x * gscale), b = fmaf(l2c, wv.y, gv.y * gscale);
      const float c = fmaf(l2c, wv.z, gv.z * gscale), d = fmaf(l2c, wv.w, gv.w * gscale);
      s += (a * a + b * b) + (c * c + d * d);
    } else {
      for (int j = i; j < ch.y && j < i + 4; ++j) {
        const float a = fmaf(l2c, wp[j], gp[j] * gscale);
        s += a * a;
      }
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one workgroup per tensor; thread j sums chunks j, j+256, ... of its tensor in order; fixed tree after.
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const int4* __restrict__ chunks, int nchunks,
                                                           const float* __restrict__ partial, float* __restrict__ norms,
                                                           int tensor_base, const int32_t* __restrict__ tcs, int chunk_base) {
  __shared__ double red[256];
  const int t = tensor_base + blockIdx.x;
  double s = 0.0;
  if (tcs) {  // this tensor's partials are partial[tcs[t] - chunk_base, tcs[t+1] - chunk_base)
    const int lo = tcs[t] - chunk_base, hi = tcs[t + 1] - chunk_base;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += (double)partial[i];
  } else {
    for (int i = threadIdx.x; i < nchunks; i += 256)
      if (chunks[i].z == t) s += (double)partial[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) norms[t] = (float)red[0];
}

struct AdamHyper {
  float gscale, clip, lr_t, b1, b2, eps;
};

__device__ __forceinline__ void adam1(float& w, float& m, float& v, float g, float l2c, float cs, const AdamHyper& h) {
  const float ge = fmaf(l2c, w, g * h.gscale) * cs;
  m = fmaf(h.b1, m, (1.0f - h.b1) * ge);
  v = fmaf(h.b2, v, (1.0f - h.b2) * ge * ge);
  w -= h.lr_t * m / (sqrtf(v) + h.eps);
}

__global__ __launch_bounds__(256) void adam_chunk_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ g, const int4* __restrict__ chunks,
                                                         const float* __restrict__ l2, const float* __restrict__ norms,
                                                         const AdamHyper h, const uint8_t* __restrict__ skip) {
  const int4 ch = chunks[blockIdx.x];
  if (skip && skip[ch.z]) return;                       // this tensor is updated by adam_tile_kernel (it owns operand images)
  const float l2c = l2[ch.z];
  float cs = 1.0f;
  if (h.clip > 0.f) cs = h.clip / fmaxf(sqrtf(norms[ch.z]), h.clip);
  float* wp = w + ch.x;
  float* mp = m + ch.x;
  float* vp = v + ch.x;
  const float* gp = g + ch.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = k * 1024 + threadIdx.x * 4;
    if (i + 3 < ch.y) {
      float4 wv = *reinterpret_cast<float4*>(wp + i);
      float4 mv = *reinterpret_cast<float4*>(mp + i);
      float4 vv = *reinterpret_cast<float4*>(vp + i);
      const float4 gv = *reinterpret_cast<const float4*>(gp + i);
      adam1(wv.x, mv.x, vv.x, gv.x, l2c, cs, h);
      adam1(wv.y, mv.y, vv.y, gv.y, l2c, cs, h);
      adam1(wv.z, mv.z, vv.z, gv.z, l2c, cs, h);
      adam1(wv.w, mv.w, vv.w, gv.w, l2c, cs, h);
      *reinterpret_cast<float4*>(wp + i) = wv;
      *reinterpret_cast<float4*>(mp + i) = mv;
      *reinterpret_cast<float4*>(vp + i) = vv;
    } else {
      for (int j = i; j < ch.y && j < i + 4; ++j) adam1(wp[j], mp[j], vp[j], gp[j], l2c, cs, h);
    }
  }
}

// ---- the same update as a 64 x 64 tile pass that also writes the matrix's operand images (round 5; see wimg.hip) --------------------
// One workgroup per tile of a row-major contiguous matrix [R, C] at `offset` floats into the arenas.  ADAM: update w, m, v exactly as
// adam_chunk_kernel does (same adam1, same operands: bitwise the same result); the updated tile stays in LDS and every image spec of
// the job whose row window contains the tile gets its blocks: the plain image of rows [row0, row0 + rows) as an [rows, K = C] operand
// and / or the transposed one ([C rows, K = rows]), three planes or one, times `scale`.  !ADAM: images only (first build / refresh
// after a host-side write to the weights).  Window contract (checked by the host): row0 % 64 == 0 and (rows % 64 == 0 or the window
// ends with the matrix), so a tile is wholly inside or wholly outside a window and rows beyond R are zeros in LDS.
constexpr int WIMG_H2_HEADER = 256;      // bytes in front of an h2 image: word 0 = max |w| the image was made under, word 1 = the next one
// Before a tile pass: every h2 image of the jobs takes the maximum the previous pass (or the owner's absmax pass before a first build)
// measured as its scale word and starts a new measurement.
__global__ __launch_bounds__(64) void wimg_roll_kernel(const yt8m_wimg_job* __restrict__ jobs) {
  const yt8m_wimg_job& J = jobs[blockIdx.x];
  const int si = threadIdx.x >> 1, which = threadIdx.x & 1;
  if (si >= J.nspec) return;
  const yt8m_wimg_spec& S = J.spec[si];
  if (S.planes != 2) return;
  void* img = which ? S.trans : S.plain;
  if (!img) return;
  unsigned* hdr = reinterpret_cast<unsigned*>(static_cast<char*>(img) - WIMG_H2_HEADER);
  hdr[0] = hdr[1];
  hdr[1] = 0u;
}

template <bool ADAM>
__global__ __launch_bounds__(256) void adam_tile_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ g, const yt8m_wimg_job* __restrict__ jobs, int njobs,
                                                        int64_t tile0, const float* __restrict__ l2, const float* __restrict__ norms,
                                                        const AdamHyper h) {
  __shared__ float T[64][65];
  const int64_t tile = tile0 + blockIdx.x;
  int j = 0;
  while (j + 1 < njobs && tile >= jobs[j + 1].tile_base) ++j;
  const yt8m_wimg_job& J = jobs[j];
  const int R = (int)J.R, C = (int)J.C;
  const int tiles_c = (C + 63) >> 6;
  const int lt = (int)(tile - J.tile_base);
  const int ty = lt / tiles_c, tx = lt - ty * tiles_c;
  const int r0 = ty * 64, c0 = tx * 64;
  const int t = threadIdx.x;
  float l2c = 0.f, cs = 1.0f;
  if (ADAM) {
    l2c = l2[J.tensor];
    if (h.clip > 0.f) cs = h.clip / fmaxf(sqrtf(norms[J.tensor]), h.clip);
  }
  {
    const int c4 = (t & 15) * 4, rr = t >> 4;
    const bool vec = (C & 3) == 0;                                   // arena tensors start on 256-byte boundaries
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 16 * i;
      float4 wv = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < R) {
        const int64_t at = J.offset + (int64_t)(r0 + r) * C + c0 + c4;
        if (vec && c0 + c4 + 3 < C) {
          wv = *reinterpret_cast<const float4*>(w + at);
          if (ADAM) {
            float4 mv = *reinterpret_cast<const float4*>(m + at);
            float4 vv = *reinterpret_cast<const float4*>(v + at);
            const float4 gv = *reinterpret_cast<const float4*>(g + at);
            adam1(wv.x, mv.x, vv.x, gv.x, l2c, cs, h);
            adam1(wv.y, mv.y, vv.y, gv.y, l2c, cs, h);
            adam1(wv.z, mv.z, vv.z, gv.z, l2c, cs, h);
            adam1(wv.w, mv.w, vv.w, gv.w, l2c, cs, h);
            *reinterpret_cast<float4*>(w + at) = wv;
            *reinterpret_cast<float4*>(m + at) = mv;
            *reinterpret_cast<float4*>(v + at) = vv;
          }
        } else {
          float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (c0 + c4 + k < C) {
              e[k] = w[at + k];
              if (ADAM) {
                float mk = m[at + k], vk = v[at + k];
                adam1(e[k], mk, vk, g[at + k], l2c, cs, h);
                w[at + k] = e[k]; m[at + k] = mk; v[at + k] = vk;
              }
            }
          }
          wv.x = e[0]; wv.y = e[1]; wv.z = e[2]; wv.w = e[3];
        }
      }
      T[r][c4 + 0] = wv.x; T[r][c4 + 1] = wv.y; T[r][c4 + 2] = wv.z; T[r][c4 + 3] = wv.w;
    }
  }
  __syncthreads();
  const int a = t & 63, blk = t >> 6;                               // row of the image within the tile, K block within the tile
  bool tmax_pending = true;
  float tile_max = 0.f;                                             // (per wave: max |w| over its quarter of the tile's columns)
  for (int si = 0; si < J.nspec; ++si) {
    const yt8m_wimg_spec& S = J.spec[si];
    const int w0 = (int)S.row0, wr = (int)S.rows;
    if (r0 < w0 || r0 >= w0 + wr) continue;                         // tile outside this window (uniform per workgroup)
    float sc = S.scale;
    const int NPF = S.planes * yt8m_x3::RG_F;
    // planes == 2 (round 6): two IEEE-half planes under the power of two the word IN FRONT of the image (256 bytes before it) gives --
    // max |w| as float bits, as yt8m_h2_absmax leaves it and the consumers' epilogues read it (gemm_x3.hip fold_device_scales).  It is
    // the maximum the PREVIOUS pass over these weights measured (wimg_roll_kernel moves it there): the split aims at 2^14 of a 2^16
    // range, so the weights of this step fit unless they quadrupled.  This pass measures the new maximum into the header's second word.
    const bool h2 = S.planes == 2;
    if (h2) {
      const unsigned* hdr = reinterpret_cast<const unsigned*>(static_cast<const char*>(S.plain ? S.plain : S.trans) - WIMG_H2_HEADER);
      sc *= yt8m_x3::pow2_scale_for(__uint_as_float(hdr[0]), 14);
    }
    if (h2 && tmax_pending) {                                          // once per tile and header
      tmax_pending = false;
      float mx = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) mx = fmaxf(mx, fabsf(T[a][blk * 16 + k]));
      mx = wave_max(mx);
      tile_max = mx;
    }
    if (h2 && (t & 63) == 0) {
      const unsigned bits = __float_as_uint(tile_max);
      if (S.plain) {
        unsigned* nx = reinterpret_cast<unsigned*>(static_cast<char*>(S.plain) - WIMG_H2_HEADER) + 1;
        if (bits > __hip_atomic_load(nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nx, bits);
      }
      if (S.trans) {
        unsigned* nx = reinterpret_cast<unsigned*>(static_cast<char*>(S.trans) - WIMG_H2_HEADER) + 1;
        if (bits > __hip_atomic_load(nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(nx, bits);
      }
    }
    if (S.plain) {
      const int KB = (C + 15) >> 4;
      const int row = r0 - w0 + a, kb = (c0 >> 4) + blk;
      if (row < ((wr + 31) & ~31) && kb < KB) {
        float e[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = T[a][blk * 16 + k] * sc;
        float* dst = static_cast<float*>(S.plain) + ((int64_t)(row >> 5) * KB + kb) * NPF;
        if (S.planes == 3) yt8m_x3::store_block<3>(e, dst, row);
        else if (h2) yt8m_x3::store_block_h2(e, dst, row);
        else yt8m_x3::store_block<1>(e, dst, row);
      }
    }
    if (S.trans) {
      const int KB = (wr + 15) >> 4;
      const int row = c0 + a, kb = ((r0 - w0) >> 4) + blk;
      if (row < ((C + 31) & ~31) && kb < KB) {
        float e[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = T[blk * 16 + k][a] * sc;
        float* dst = static_cast<float*>(S.trans) + ((int64_t)(row >> 5) * KB + kb) * NPF;
        if (S.planes == 3) yt8m_x3::store_block<3>(e, dst, row);
        else if (h2) yt8m_x3::store_block_h2(e, dst, row);
        else yt8m_x3::store_block<1>(e, dst, row);
      }
    }
  }
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_sqnorm_multi(const float* w, const float* g, const int32_t* chunks, int64_t nchunks, const float* l2,
                                 float gscale, float* partial, float* norms, int64_t tensor_base, int64_t ntensors,
                                 const int32_t* tensor_chunk_start, int64_t chunk_base, yt8m_stream_t stream) {
  YT8M_REQUIRE(nchunks >= 0 && ntensors >= 0 && tensor_base >= 0 && nchunks < (1LL << 31), YT8M_E_SHAPE, "bad chunk/tensor count");
  if (ntensors == 0) return YT8M_OK;
  YT8M_REQUIRE(w && g && chunks && l2 && partial && norms, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0, YT8M_E_BADARG,
               "arena must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_OPTIM, s);
  if (nchunks > 0)
    hipLaunchKernelGGL(sqnorm_chunk_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, w, g,
                       reinterpret_cast<const int4*>(chunks), l2, gscale, partial);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3((unsigned)ntensors), dim3(256), 0, s,
                     reinterpret_cast<const int4*>(chunks), (int)nchunks, partial, norms, (int)tensor_base,
                     tensor_chunk_start, (int)chunk_base);
  return launch_status("sqnorm kernels");
}

extern "C" int yt8m_adam_multi(float* w, float* m, float* v, const float* g, const int32_t* chunks, int64_t nchunks,
                               const float* l2, float gscale, const float* norms, float clip, float lr_t, float beta1,
                               float beta2, float eps, yt8m_stream_t stream) {
  return yt8m_adam_multi_ex(w, m, v, g, chunks, nchunks, l2, gscale, norms, clip, lr_t, beta1, beta2, eps, nullptr, stream);
}

// skip_tensor (device uint8[all tensors], may be NULL): chunks of a flagged tensor are left alone -- its update belongs to
// yt8m_adam_tiles, which also rewrites the tensor's operand images.
extern "C" int yt8m_adam_multi_ex(float* w, float* m, float* v, const float* g, const int32_t* chunks, int64_t nchunks,
                                  const float* l2, float gscale, const float* norms, float clip, float lr_t, float beta1,
                                  float beta2, float eps, const uint8_t* skip_tensor, yt8m_stream_t stream) {
  YT8M_REQUIRE(nchunks >= 0 && nchunks < (1LL << 31), YT8M_E_SHAPE, "bad chunk count");
  if (nchunks == 0) return YT8M_OK;
  YT8M_REQUIRE(w && m && v && g && chunks && l2, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(clip <= 0.f || norms, YT8M_E_BADARG, "clip > 0 needs norms");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_OPTIM, s);
  AdamHyper h{gscale, clip, lr_t, beta1, beta2, eps};
  hipLaunchKernelGGL(adam_chunk_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, w, m, v, g,
                     reinterpret_cast<const int4*>(chunks), l2, norms, h, skip_tensor);
  return launch_status("adam_chunk_kernel");
}

// Fills tile_base of a HOST job array (cumulative 64 x 64 tile counts) and validates it; returns the total number of tiles or a
// negative status.  The caller uploads the array and hands the device copy to yt8m_adam_tiles.
extern "C" int64_t yt8m_wimg_jobs_layout(yt8m_wimg_job* jobs, int64_t njobs) {
  YT8M_REQUIRE(jobs && njobs >= 1 && njobs <= 4096, YT8M_E_BADARG, "1..4096 jobs");
  int64_t tiles = 0;
  for (int64_t j = 0; j < njobs; ++j) {
    yt8m_wimg_job& J = jobs[j];
    YT8M_REQUIRE(J.R >= 1 && J.C >= 1 && J.R < (1LL << 31) && J.C < (1LL << 31) && J.offset >= 0 && (J.offset & 3) == 0, YT8M_E_SHAPE,
                 "bad matrix");
    YT8M_REQUIRE(J.nspec >= 0 && J.nspec <= 4 && J.tensor >= 0, YT8M_E_BADARG, "0..4 image specs per job");
    for (int i = 0; i < J.nspec; ++i) {
      const yt8m_wimg_spec& S = J.spec[i];
      YT8M_REQUIRE(S.planes == 1 || S.planes == 2 || S.planes == 3, YT8M_E_BADARG, "planes must be 1, 2 (h2: header in front of the image) or 3");
      YT8M_REQUIRE(S.plain || S.trans, YT8M_E_BADARG, "a spec needs an image");
      YT8M_REQUIRE(((reinterpret_cast<uintptr_t>(S.plain) | reinterpret_cast<uintptr_t>(S.trans)) & 15) == 0, YT8M_E_BADARG,
                   "images must be 16-byte aligned");
      YT8M_REQUIRE(S.row0 >= 0 && S.rows >= 1 && S.row0 + S.rows <= J.R && (S.row0 & 63) == 0 &&
                       ((S.rows & 63) == 0 || S.row0 + S.rows == J.R), YT8M_E_SHAPE,
                   "row window must start on a 64-row tile and end on one or with the matrix");
    }
    J.tile_base = tiles;
    tiles += ((J.R + 63) / 64) * ((J.C + 63) / 64);
  }
  YT8M_REQUIRE(tiles < (1LL << 31), YT8M_E_SHAPE, "too many tiles");
  return tiles;
}

// Adam (do_adam != 0) + operand images of the matrices jobs[0 .. njobs) (DEVICE array laid out by yt8m_wimg_jobs_layout; a sub-range
// of a larger array is fine: tile0 = tile_base of its first job, ntiles = the range's tile count).  Hyper-parameters as yt8m_adam_multi.
// do_adam == 0: images only, the arenas are not written (m, v, g, l2, norms may be NULL).
extern "C" int yt8m_adam_tiles(float* w, float* m, float* v, const float* g, const yt8m_wimg_job* jobs, int64_t njobs, int64_t tile0,
                               int64_t ntiles, const float* l2, float gscale, const float* norms, float clip, float lr_t, float beta1,
                               float beta2, float eps, int do_adam, yt8m_stream_t stream) {
  YT8M_REQUIRE(njobs >= 0 && ntiles >= 0 && ntiles < (1LL << 31) && tile0 >= 0, YT8M_E_SHAPE, "bad job / tile count");
  if (njobs == 0 || ntiles == 0) return YT8M_OK;
  YT8M_REQUIRE(w && jobs, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(!do_adam || (m && v && g && l2), YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(!do_adam || clip <= 0.f || norms, YT8M_E_BADARG, "clip > 0 needs norms");
  YT8M_REQUIRE((reinterpret_cast<uintptr_t>(w) & 15) == 0, YT8M_E_BADARG, "arena must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_OPTIM, s);
  AdamHyper h{gscale, clip, lr_t, beta1, beta2, eps};
  hipLaunchKernelGGL(wimg_roll_kernel, dim3((unsigned)njobs), dim3(64), 0, s, jobs);        // (h2 images: scale word <- measured maximum)
  if (do_adam)
    hipLaunchKernelGGL(adam_tile_kernel<true>, dim3((unsigned)ntiles), dim3(256), 0, s, w, m, v, g, jobs, (int)njobs, tile0, l2, norms, h);
  else
    hipLaunchKernelGGL(adam_tile_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, s, w, m, v, g, jobs, (int)njobs, tile0, l2, norms, h);
  return launch_status("adam_tile_kernel");
}

// clip + Adam of tensor ranges of an arena: what youtube-8m_amd/ops.py sqnorm_and_adam enqueues per range -- the two chunk passes over
// the range's slice of the chunk table and, for the image-owning matrices in it, the tile pass -- for a host that is not Python
// (and for yt8m_lstm_stack_bwd's early pass: lstm_stack.hip).
extern "C" int yt8m_optimizer_ranges(const yt8m_opt_ranges* o, yt8m_stream_t stream) {
  YT8M_REQUIRE(o && o->nranges >= 1 && o->nranges <= 8, YT8M_E_BADARG, "1..8 ranges");
  YT8M_REQUIRE(o->w && o->m && o->v && o->g && o->chunks && o->tensor_chunk_start && o->tensor_chunk_start_host && o->l2 && o->partial &&
                   o->norms, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(o->njobs >= 0 && (o->njobs == 0 || (o->jobs && o->job_tensor_host && o->job_tile_base_host && o->skip_tensor)), YT8M_E_BADARG,
               "image jobs need their tables and the skip flags");
  if (o->after_stream && o->after_stream != stream) {                // gradients some other stream is still writing
    hipEvent_t e;
    YT8M_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t rc1 = hipEventRecord(e, as_stream(o->after_stream));
    hipError_t rc2 = rc1 == hipSuccess ? hipStreamWaitEvent(as_stream(stream), e, 0) : rc1;
    (void)hipEventDestroy(e);                                        // (released once the recorded work has completed)
    YT8M_HIP_CHECK(rc2);
  }
  for (int r = 0; r < o->nranges; ++r) {
    const int lo = o->range_lo[r], hi = o->range_hi[r];
    YT8M_REQUIRE(lo >= 0 && hi >= lo, YT8M_E_BADARG, "bad tensor range");
    if (hi == lo) continue;
    const int64_t c0 = o->tensor_chunk_start_host[lo], c1 = o->tensor_chunk_start_host[hi];
    if (o->clip > 0.f) {
      const int rc = yt8m_sqnorm_multi(o->w, o->g, o->chunks + 4 * c0, c1 - c0, o->l2, o->gscale, o->partial + c0, o->norms, lo, hi - lo,
                                       o->tensor_chunk_start, c0, stream);
      if (rc != YT8M_OK) return rc;
    }
    int rc = yt8m_adam_multi_ex(o->w, o->m, o->v, o->g, o->chunks + 4 * c0, c1 - c0, o->l2, o->gscale, o->norms, o->clip, o->lr_t, o->beta1,
                                o->beta2, o->eps, o->njobs ? o->skip_tensor : nullptr, stream);
    if (rc != YT8M_OK) return rc;
    if (o->njobs) {
      int j0 = 0, j1 = 0;
      while (j0 < o->njobs && o->job_tensor_host[j0] < lo) ++j0;
      j1 = j0;
      while (j1 < o->njobs && o->job_tensor_host[j1] < hi) ++j1;
      if (j1 > j0) {
        rc = yt8m_adam_tiles(o->w, o->m, o->v, o->g, o->jobs + j0, j1 - j0, o->job_tile_base_host[j0],
                             o->job_tile_base_host[j1] - o->job_tile_base_host[j0], o->l2, o->gscale, o->norms, o->clip, o->lr_t, o->beta1,
                             o->beta2, o->eps, 1, stream);
        if (rc != YT8M_OK) return rc;
      }
    }
  }
  return YT8M_OK;
}

// ---- tf.clip_by_norm of ONE tensor (W/utils.py:164-174 clip_gradient_norms: the reference's signature, for hosts that hold their own
// gradient list; the training step clips inside the fused optimiser pass above) -----------------------------------------------------
namespace {
constexpr int CLIP_BLOCKS = 256;
__global__ __launch_bounds__(256) void clip_sqnorm_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)CLIP_BLOCKS * 256) acc += g[i] * g[i];
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void clip_scale_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n,
                                                         const float* __restrict__ partial, float max_norm) {
  __shared__ float red[4];
  const float tot = block_sum_256(partial[threadIdx.x], red);         // every workgroup sums the 256 partials in the same order
  const float f = max_norm / fmaxf(sqrtf(tot), max_norm);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = g[i] * f;
}
}  // namespace

// out = g * max_norm / max(||g||, max_norm) (out may alias g).  workspace: 256 floats.
extern "C" int yt8m_clip_by_norm_f32(const float* g, float* out, int64_t n, float max_norm, float* workspace, yt8m_stream_t stream) {
  YT8M_REQUIRE(n >= 0 && max_norm > 0.f, YT8M_E_BADARG, "n >= 0, max_norm > 0");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(g && out && workspace, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_OPTIM, s);
  hipLaunchKernelGGL(clip_sqnorm_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, s, g, n, workspace);
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, s, g, out, n, workspace, max_norm);
  return launch_status("clip_by_norm");
}
