// random.hip -- counter-based random elementwise ops of the training path (gfx950, wave64; HBM-bound, one pass).
//   dropout:  tf.nn.dropout(x, keep_prob)  = x / keep_prob * floor(keep_prob + u),  u ~ U[0,1)
//             call sites: W/all_video_models/deep_combine_chain_model.py:57-58 and tf.contrib.rnn.DropoutWrapper(cell,
//             input_keep_prob) in W/all_frame_models/lstm_memory_model.py:36-45
//   noise:    x + N(0, stddev^2)           (W/all_frame_models/lstm_memory_model.py:62-63)
// The stream is Philox4x32-10 (Salmon et al., SC'11): element e of a logical tensor takes word (e & 3) of the block with
// counter (e >> 2, 0) under key = seed.  A mask is therefore a pure function of (seed, element index): the backward pass
// re-generates it instead of storing it, chunks of a tensor (offset = first element) draw the same numbers as one call
// over the whole tensor, and oracle/philox.py reproduces it bit for bit.  TF's own random ops use other streams; parity
// with the reference is distributional only (the reference pins none of it).
#include "common.h"
#include "philox.h"

namespace {

using yt8m_rng::U4;
using yt8m_rng::philox4x32_10;
using yt8m_rng::u01;

__device__ __forceinline__ float drop1(float x, uint32_t r, float keep) {
  return (keep + u01(r)) >= 1.0f ? __fdiv_rn(x, keep) : 0.0f;
}

// one thread per Philox block = 4 consecutive elements of the logical tensor
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float keep,
                                                      uint64_t seed, int64_t offset, int vec) {
  const int64_t g = (offset >> 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e0 = g * 4 - offset;                                    // index into x of the block's first element
  if (e0 >= n) return;
  const U4 r = philox4x32_10((uint64_t)g, seed);
  if (vec && e0 >= 0 && e0 + 4 <= n) {
    const float4 v = *reinterpret_cast<const float4*>(x + e0);
    float4 o;
    o.x = drop1(v.x, r.x, keep); o.y = drop1(v.y, r.y, keep); o.z = drop1(v.z, r.z, keep); o.w = drop1(v.w, r.w, keep);
    *reinterpret_cast<float4*>(y + e0) = o;
    return;
  }
  const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t e = e0 + k;
    if (e >= 0 && e < n) y[e] = drop1(x[e], rr[k], keep);
  }
}

// Box-Muller on word pairs (x,y) -> elements 0,1 and (z,w) -> elements 2,3 of the block
__global__ __launch_bounds__(256) void noise_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float stddev,
                                                    uint64_t seed, int64_t offset) {
  const int64_t g = (offset >> 2) + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t e0 = g * 4 - offset;
  if (e0 >= n) return;
  const U4 r = philox4x32_10((uint64_t)g, seed);
  float z[4];
  {
    const float ra = sqrtf(-2.0f * logf(1.0f - u01(r.x))), th = 6.283185307179586f * u01(r.y);
    z[0] = ra * cosf(th); z[1] = ra * sinf(th);
    const float rb = sqrtf(-2.0f * logf(1.0f - u01(r.z))), ph = 6.283185307179586f * u01(r.w);
    z[2] = rb * cosf(ph); z[3] = rb * sinf(ph);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t e = e0 + k;
    if (e >= 0 && e < n) y[e] = x[e] + stddev * z[k];
  }
}

}  // namespace

using namespace yt8m;

extern "C" int yt8m_dropout_f32(const float* x, float* y, int64_t n, float keep_prob, uint64_t seed, int64_t offset,
                                yt8m_stream_t stream) {
  YT8M_REQUIRE(n >= 0 && offset >= 0, YT8M_E_SHAPE, "negative size or offset");
  YT8M_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, YT8M_E_BADARG, "keep_prob must be in (0, 1]");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(x && y, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t groups = ((offset + n + 3) >> 2) - (offset >> 2);
  const int vec = ((offset & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) ? 1 : 0;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, x, y, n, keep_prob, seed, offset, vec);
  return launch_status("dropout_kernel");
}

extern "C" int yt8m_add_noise_f32(const float* x, float* y, int64_t n, float stddev, uint64_t seed, int64_t offset,
                                  yt8m_stream_t stream) {
  YT8M_REQUIRE(n >= 0 && offset >= 0, YT8M_E_SHAPE, "negative size or offset");
  YT8M_REQUIRE(stddev >= 0.f, YT8M_E_BADARG, "stddev must be >= 0");
  if (n == 0) return YT8M_OK;
  YT8M_REQUIRE(x && y, YT8M_E_BADARG, "null operand");
  hipStream_t s = as_stream(stream);
  ProfScope prof(F_ELEMENTWISE, s);
  const int64_t groups = ((offset + n + 3) >> 2) - (offset >> 2);
  hipLaunchKernelGGL(noise_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, x, y, n, stddev, seed, offset);
  return launch_status("noise_kernel");
}
