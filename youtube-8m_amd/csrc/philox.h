// philox.h -- Philox4x32-10 block function shared by random.hip and dbof.hip (Salmon et al., SC'11; oracle/philox.py).
// Element e of a logical tensor takes word (e & 3) of the block with counter (e >> 2, 0) under key = seed.
#pragma once
#include <stdint.h>

namespace yt8m_rng {

struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint64_t ctr, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-8f; }   // 24 bits, exact

// uniform in [0, 1) of logical element e
__device__ __forceinline__ float uniform_at(uint64_t e, uint64_t key) {
  const U4 r = philox4x32_10(e >> 2, key);
  const uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
  return u01(w);
}

}  // namespace yt8m_rng
