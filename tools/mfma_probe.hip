// Probe: operand lane layout of v_mfma_i32_32x32x32_i8 on gfx950 (hypothesis test with random int8 data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const int8_t* A, const int8_t* B, int* C, int hyp) {
  int l = threadIdx.x;
  int8_t a[16], b[16];
  for (int j = 0; j < 16; ++j) {
    int kk = hyp == 0 ? 16 * (l >> 5) + j : (j < 8 ? 8 * (l >> 5) + j : 16 + 8 * (l >> 5) + (j - 8));
    a[j] = A[(l & 31) * 32 + kk];      // A[row][k]
    b[j] = B[kk * 32 + (l & 31)];      // B[k][col]
  }
  v4i av, bv;
  __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
  v16i c = {0};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    C[row * 32 + col] = c[r];
  }
}
int main() {
  int8_t hA[1024], hB[1024]; int hC[1024], ref[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) { hA[i] = rand() % 255 - 127; hB[i] = rand() % 255 - 127; }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[i*32+kk] * hB[kk*32+j]; ref[i*32+j] = s; }
  int8_t *dA, *dB; int* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  for (int hyp = 0; hyp < 2; ++hyp) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, hyp);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
    printf("hypothesis %d: %d mismatches of 1024\n", hyp, bad);
  }
  return 0;
}
