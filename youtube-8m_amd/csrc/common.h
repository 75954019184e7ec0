// common.h -- shared host-side helpers for libyt8m_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/yt8m_hip.h"

namespace yt8m {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b, c);
  return code;
}

#define YT8M_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      snprintf(yt8m::g_err, sizeof(yt8m::g_err), "%s failed: %s (%s:%d)", #expr,          \
               hipGetErrorString(_e), __FILE__, __LINE__);                                \
      return YT8M_E_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define YT8M_REQUIRE(cond, code, msg)                                                     \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      snprintf(yt8m::g_err, sizeof(yt8m::g_err), "%s: requirement failed: %s", __func__, msg); \
      return code;                                                                        \
    }                                                                                     \
  } while (0)

// ---- per-family hipEvent profiling (bench.py roofline leg) -----------------------------------
enum Family { F_GEMM = 0, F_MOE_FUSED = 1, F_ELEMENTWISE = 2, F_OPTIM = 3, F_LSTM = 4, F_NETVLAD = 5, F_LSTM_BWD = 6, F_GEMM_X3 = 7,
              F_GEMM_X1X3 = 8,   // x3 kernel with a ONE-plane A operand (three bf16 products per fp32 product)
              F_VLAD_ROWS = 9,   // the two streaming kernels of the fused NetVLAD pooling, timed inside the F_NETVLAD scope with their
              F_VLAD_COLS = 10,  // algorithmic HBM bytes declared (HBM-bound: bench.py reports bytes / time against 8 TB/s)
              F_NETVLAD_FWD = 11,  // the whole yt8m_netvlad_fwd_u8 call with SURVEY.md 8(d)'s bytes (uint8 frames once + parameters):
                                   // the forward pooling against the HBM roof on the bytes the ALGORITHM needs
              F_GEMM_H2 = 12,      // fp32 products as three f16 MFMA products of two-plane half images (gemm_h2q_kernel, round 5)
              F_GEMM_H1X2 = 13,    // ... with a ONE-plane exact A operand (uint8 frames minus 128): two products
              F_COUNT = 14 };

struct ProfScope {
  int fam;
  hipStream_t s;
  bool on;
  hipEvent_t e0, e1;
  double flops;           // algorithmic FLOPs of the launches inside the scope (0: not counted)
  double bytes;           // algorithmic HBM bytes of the launches inside the scope (0: not counted)
  ProfScope(int family, hipStream_t stream, double algorithmic_flops = 0.0, double algorithmic_bytes = 0.0);
  ~ProfScope();
};

bool prof_enabled();

// hipGraph replay of a launch-bound kernel chain (runtime.hip).  The key must identify every argument the chain depends on.
struct GraphKey {
  int kind;
  int pad;
  const void* p[10];
  int64_t v[8];
};
}  // namespace yt8m
#include <atomic>
#include <functional>
namespace yt8m {
int run_chain(const GraphKey& key, hipStream_t s, const std::function<int()>& launch_all);

inline hipStream_t as_stream(yt8m_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute and launch functions run on several host threads (one per
// stream of the layer pipeline): one flag per (call site, device).  Usage: static DeviceOnce once; once.lds(kernel, bytes);
struct DeviceOnce {
  std::atomic<bool> done[64];
  DeviceOnce() { for (auto& d : done) d.store(false); }
  hipError_t lds(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && done[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done[dev].store(true, std::memory_order_release);
    return e;
  }
};

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "launch of %s failed: %s", what, hipGetErrorString(e));
    return YT8M_E_HIP;
  }
  return YT8M_OK;
}

}  // namespace yt8m

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x == 256 (4 waves); result valid in all threads
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
