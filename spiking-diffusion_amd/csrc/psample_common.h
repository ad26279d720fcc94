// Noise and wave-reduction helpers shared by the sampler kernels (psample.hip, step_tail.hip): the Philox4x32-10 counter
// scheme of spk_psample_step (u: stream 0, counter offset + position; q: stream 1, counter offset + position * K + class) and
// the 64-lane max / sum by lane shuffles.  Every kernel that draws noise goes through these, so that the dense loop, the
// active-set forms, the fused step tail and spk_philox_noise see the same draws.
#pragma once
#include "spk_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

// Philox4x32-10: counter (index, stream) , key = seed
__device__ __forceinline__ void philox4x32(unsigned long long seed, unsigned long long index, uint32_t stream,
                                            uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)index, (uint32_t)(index >> 32), stream, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float u01_open_left(uint32_t r) {   // (0, 1]
  return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float u01_open_right(uint32_t r) {  // [0, 1)
  return (float)(r >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

}  // namespace
