// Probe: sustained issue rate of the fp32 MFMA shapes on gfx950 -- independent accumulators vs ONE accumulator (a dependent chain),
// one / two waves per SIMD.  (s_memtime ticks at 100 MHz: wall time is what is reported.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void k(float* out, int iters) {
  float a = (float)threadIdx.x, b = 1.0f + (float)(threadIdx.x & 7);
  v16f f0 = {0}, f1 = {0}, f2 = {0}, f3 = {0};
  v4f g0 = {0}, g1 = {0}, g2 = {0}, g3 = {0};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {          // 32x32x2, four accumulators
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f0, 0, 0, 0);
      f1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f2, 0, 0, 0);
      f3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f3, 0, 0, 0);
    } else if (KIND == 1) {   // 32x32x2, one accumulator
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f0, 0, 0, 0);
      f0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, f0, 0, 0, 0);
    } else if (KIND == 2) {   // 16x16x4, four accumulators
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g1, 0, 0, 0);
      g2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g2, 0, 0, 0);
      g3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g3, 0, 0, 0);
    } else {                  // 16x16x4, one accumulator
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g0, 0, 0, 0);
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g0, 0, 0, 0);
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g0, 0, 0, 0);
      g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, g0, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += f0[r] + f1[r] + f2[r] + f3[r];
  for (int r = 0; r < 4; ++r) s += g0[r] + g1[r] + g2[r] + g3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, int nblk, int threads) {
  float* out;
  hipMalloc(&out, nblk * threads * 4);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (KIND < 2 ? 4096.0 : 2048.0) * 4.0 * iters * (threads / 64) * nblk;
  printf("%-28s blocks=%4d waves/SIMD=%d: %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, nblk, threads / 256,
         ms * 1e6 / (4.0 * iters * (threads / 256)), flop / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int threads : {256, 512}) {
    run<0>("f32_32x32x2 x4 accumulators", 256, threads);
    run<1>("f32_32x32x2 one accumulator", 256, threads);
    run<2>("f32_16x16x4 x4 accumulators", 256, threads);
    run<3>("f32_16x16x4 one accumulator", 256, threads);
  }
  return 0;
}
