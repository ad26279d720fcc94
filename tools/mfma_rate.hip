// Probe: sustained issue rate of i8 / bf16 MFMA shapes on gfx950 (one wave per SIMD, independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256, 1) void k(int* out, long long* cyc, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, 6, (int)threadIdx.x, 7};
  v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  v4i32 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  v16f f0 = {0}, f1 = {0}, f2 = {0}, f3 = {0};
  v8s sa = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, sb = {1, 1, 2, 2, 3, 3, 4, 4};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    } else if (KIND == 1) {
      d0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d2, 0, 0, 0);
      d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d3, 0, 0, 0);
    } else {
      f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, sb, f0, 0, 0, 0);
      f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, sb, f1, 0, 0, 0);
      f2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, sb, f2, 0, 0, 0);
      f3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, sb, f3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + (int)f0[r] + (int)f1[r] + (int)f2[r] + (int)f3[r];
  for (int r = 0; r < 4; ++r) s += d0[r] + d1[r] + d2[r] + d3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char* name, int nblk) {
  int* out; long long* cyc; long long h[1024];
  hipMalloc(&out, nblk * 256 * 4); hipMalloc(&cyc, nblk * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost);
  double per = (double)h[0] / (4.0 * iters);
  printf("%-22s blocks=%4d: %.1f shader cycles per MFMA (block 0), wall %.3f ms -> %.1f ns per MFMA per wave, eff clock %.2f GHz\n",
         name, nblk, per, ms, ms * 1e6 / (4.0 * iters), per / (ms * 1e6 / (4.0 * iters)));
}
int main() {
  for (int nblk : {1, 256}) {
    run<0>("i32_32x32x32_i8", nblk);
    run<1>("i32_16x16x64_i8", nblk);
    run<2>("f32_32x32x16_bf16", nblk);
  }
  return 0;
}
