#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  unsigned a = threadIdx.x, b = threadIdx.x + 100;
  v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; unsigned h[128];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 4) printf("lane %2d: a'=%3u b'=%3u\n", i, h[i], h[64 + i]);
  return 0;
}
