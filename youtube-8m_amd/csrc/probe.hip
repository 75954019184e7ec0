// probe.hip -- hardware probes used by bench.py / tools to put measured ceilings next to the roofline numbers.
//   yt8m_probe_mfma_f32 : register-only v_mfma_f32_32x32x2_f32 loop (4 independent accumulators per wave),
//                         i.e. the matrix-pipe ceiling of THIS box at its sustained clock.
//   yt8m_probe_mfma_bf16: the same loop on v_mfma_f32_32x32x16_bf16 (the ceiling of gemm_bf16.hip / gemm_x3.hip).
//   yt8m_probe_copy_f32 : float4 streaming copy (HBM ceiling).
//   yt8m_probe_placement: which XCD / CU every workgroup of a launch landed on (tools/cu_mask_probe.py: how the bits of a
//                         hipExtStreamCreateWithCUMask mask map to XCDs -- measured: bit i -> XCD i % 8).
#include "common.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, float* __restrict__ sink) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  float x = (float)(threadIdx.x & 7) * 0.125f + 0.5f, y = (float)(threadIdx.x & 3) * 0.25f - 0.3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 123.456f) sink[0] = s;  // keep the chain live
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_bf16_probe_kernel(int iters, float* __restrict__ sink, int random_operands) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  bf16x8 x, y;
#pragma unroll
  for (int r = 0; r < 8; ++r) { x[r] = (__bf16)((float)((threadIdx.x + r) & 7) * 0.125f + 0.5f); y[r] = (__bf16)((float)((threadIdx.x + r) & 3) * 0.25f - 0.3f); }
  if (random_operands) {                                             // full-entropy mantissas / signs: the data-dependent power draw of a real GEMM
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      h = h * 1664525u + 1013904223u;
      x[r] = __builtin_bit_cast(__bf16, (unsigned short)(((h >> 16) & 0x80FFu) | 0x3F00u));
      h = h * 1664525u + 1013904223u;
      y[r] = __builtin_bit_cast(__bf16, (unsigned short)(((h >> 16) & 0x80FFu) | 0x3E00u));
    }
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256) void copy_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

// where workgroups land: out[2 b] = XCC id, out[2 b + 1] = HW_ID (cu / sh / se fields); a short spin keeps early workgroups
// resident so the grid spreads over every CU the queue may use
__global__ __launch_bounds__(256) void placement_probe_kernel(int* __restrict__ out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)(xcc & 0xF); out[2 * blockIdx.x + 1] = (int)hw; }
}
}  // namespace

using namespace yt8m;

extern "C" int yt8m_probe_placement(int* out, int blocks, int spin_ticks, yt8m_stream_t stream) {
  YT8M_REQUIRE(out && blocks > 0, YT8M_E_BADARG, "bad probe arguments");
  hipLaunchKernelGGL(placement_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, spin_ticks);
  return launch_status("placement_probe_kernel");
}

// A stream whose dispatches may only use the CUs of `mask` (bit i of word i / 32 = CU i in the runtime's numbering).  seq_ops
// uses two of them to keep the GEMMs that run beside a half-chip persistent recurrence off the CUs that recurrence needs.
extern "C" int yt8m_stream_create_cu_mask(const uint32_t* mask, int words, yt8m_stream_t* stream) {
  YT8M_REQUIRE(mask && words > 0 && stream, YT8M_E_BADARG, "bad mask arguments");
  hipStream_t s = nullptr;
  YT8M_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream = (yt8m_stream_t)s;
  return YT8M_OK;
}

extern "C" int yt8m_stream_destroy(yt8m_stream_t stream) {
  YT8M_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
  return YT8M_OK;
}

// FLOPs executed = blocks * 4 waves * iters * 32 MFMAs * (2*32*32*2)
extern "C" int yt8m_probe_mfma_f32(int iters, int blocks, float* sink, yt8m_stream_t stream) {
  YT8M_REQUIRE(iters > 0 && blocks > 0 && sink, YT8M_E_BADARG, "bad probe arguments");
  hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), iters, sink);
  return launch_status("mfma_probe_kernel");
}

// FLOPs executed = blocks * 4 waves * iters * 32 MFMAs * (2*32*32*16)
extern "C" int yt8m_probe_mfma_bf16(int iters, int blocks, int random_operands, float* sink, yt8m_stream_t stream) {
  YT8M_REQUIRE(iters > 0 && blocks > 0 && sink, YT8M_E_BADARG, "bad probe arguments");
  hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), iters, sink, random_operands);
  return launch_status("mfma_bf16_probe_kernel");
}

extern "C" int yt8m_probe_copy_f32(const float* src, float* dst, int64_t n, yt8m_stream_t stream) {
  YT8M_REQUIRE(n >= 0 && n % 4 == 0 && src && dst, YT8M_E_BADARG, "bad probe arguments");
  hipLaunchKernelGGL(copy_probe_kernel, dim3(2048), dim3(256), 0, as_stream(stream), reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n / 4);
  return launch_status("copy_probe_kernel");
}
