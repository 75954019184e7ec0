// optim.hip -- optimiser slice of build_graph as two multi-tensor passes over one flat fp32 arena.
//   W/train.py:435-466 (final_loss = label_loss + sum l2*0.5*|W|^2 -> compute_gradients -> clip -> apply)
//   W/utils.py:164-174 (per-tensor tf.clip_by_norm), tf.train.AdamOptimizer (TF-1 form).
// HBM-bound: sqnorm reads w,g (8 B/param); adam reads w,m,v,g and writes w,m,v (28 B/param).  Every
// chunk is <= 4096 consecutive floats of ONE tensor starting on a 16-byte boundary, so a 256-thread
// workgroup streams it as 4 float4 per lane.  Reductions are fixed-order => bitwise reproducible, which
// keeps data-parallel ranks identical after the gradient all-reduce.
#include "common.h"

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
                                                         const AdamHyper h) {
  const int4 ch = chunks[blockIdx.x];
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
  YT8M_REQUIRE(nchunks >= 0 && nchunks < (1LL << 31), YT8M_E_SHAPE, "bad chunk count");
  if (nchunks == 0) return YT8M_OK;
  YT8M_REQUIRE(w && m && v && g && chunks && l2, YT8M_E_BADARG, "null operand");
  YT8M_REQUIRE(clip <= 0.f || norms, YT8M_E_BADARG, "clip > 0 needs norms");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_OPTIM, s);
  AdamHyper h{gscale, clip, lr_t, beta1, beta2, eps};
  hipLaunchKernelGGL(adam_chunk_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, w, m, v, g,
                     reinterpret_cast<const int4*>(chunks), l2, norms, h);
  return launch_status("adam_chunk_kernel");
}
