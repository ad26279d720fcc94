// spk_clock_probe: what shader clock does this device hold under a matrix-core load?
//
// Measurement aid for bench.py (MI355X_MICROARCH.md "DVFS give-back" item 6): every workgroup (one wave per SIMD)
// issues `iters` rounds of four independent block-scaled fp6 x fp4 MFMAs on pseudo-random operands (the instruction
// of the denoiser kernel: v_mfma_scale_f32_32x32x64_f8f6f4) and stamps s_memtime (shader cycles) and s_memrealtime
// (constant 100 MHz) around the loop.  clock [GHz] = d(memtime) / d(memrealtime) * 0.1.  Devices of the pool differ
// by ~10 % in the clock they sustain, which moves every MFMA-bound number of the bench line with it; the probe makes
// that visible in the line itself.  It is a separate diagnostic kernel: no product kernel executes a stamp.
#include "spk_common.h"
#include "../../include/spkdiff.h"

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void clock_probe_kernel(unsigned long long* out, int iters) {
  // operands: fp4 codes 0x0 / 0x2 (spikes) and e2m3 digit codes, different per lane
  unsigned h = 0x9E3779B9u * (threadIdx.x + 1u) + 0x85EBCA6Bu * (blockIdx.x + 1u);
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u;
    a[i] = (int)(h & 0x22222222u);
    h = h * 1664525u + 1013904223u;
    b[i] = i < 6 ? (int)(h & 0x6DB6DB6Du) : 0;
  }
  v16f f0 = {0}, f1 = {0}, f2 = {0}, f3 = {0};
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, f0, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, f1, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, f2, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, f3, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += f0[r] + f1[r] + f2[r] + f3[r];
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = c1 - c0;
    out[4 * blockIdx.x + 1] = r1 - r0;
    out[4 * blockIdx.x + 2] = (unsigned long long)(4ll * iters);
  }
  if (s == 123456.789f) out[4 * blockIdx.x + 3] = 1;      // keeps the MFMAs alive; never true in practice
}

extern "C" int spk_clock_probe(unsigned long long* out, int nblocks, int iters, hipStream_t stream) {
  if (!out || nblocks <= 0 || iters <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(nblocks), dim3(256), 0, stream, out, iters);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
