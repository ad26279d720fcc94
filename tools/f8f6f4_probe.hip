// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 with A = fp4 (e2m1) spikes and B = fp6 (e2m3) weight digits.
//   1. operand packing / lane layout hypothesis (lane l: row|col = l & 31, k = 32 * (l >> 5) + j, little-endian packing)
//   2. exactness of the fp32 accumulation for integer digit sums (digits d in [-16, 16] encoded as d / 8)
//   3. sustained issue rate next to v_mfma_i32_32x32x32_i8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A: [32 rows][64 k] spikes 0/1; B: [64 k][32 cols] digits; C: [32][32] float; `reps` accumulating repeats
__global__ void layout_k(const uint8_t* A, const int8_t* B, float* C, int reps, int scale_b) {
  const int l = threadIdx.x, rc = l & 31, kh = l >> 5;
  unsigned aw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const int kk = 32 * kh + j;
    const unsigned a4 = A[rc * 64 + kk] ? 0x2u : 0x0u;                 // e2m1 1.0 = 0b0010
    aw[j >> 3] |= a4 << (4 * (j & 7));
    const int d = B[kk * 32 + rc];
    const unsigned mag = (unsigned)(d < 0 ? -d : d);                    // e2m3: value = mag / 8 for mag <= 16
    const unsigned code = (d < 0 ? 0x20u : 0u) | mag;
    const int bit = 6 * j;
    bw[bit >> 5] |= code << (bit & 31);
    if ((bit & 31) > 26) bw[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
  }
  v8i av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (int)aw[i]; bv[i] = (int)bw[i]; }
  v16f c = {0};
  for (int r = 0; r < reps; ++r)
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4 /*A fp4*/, 2 /*B fp6 e2m3*/, 0, 0x7f7f7f7f, 0, scale_b);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    C[row * 32 + col] = c[r];
  }
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void rate_k(float* out, long long* cyc, int iters) {
  v4i a4 = {(int)threadIdx.x, 1, 2, 3}, b4 = {5, 6, (int)threadIdx.x, 7};
  v8i a8 = {0x22022002, 0x20202222, 0x02022020, 0x22222222, 0, 0, 0, 0};
  v8i b8 = {0x11111111, 0x01010101, 0x10101010, 0x12345678, 0x01020304, 0x04030201, 0, 0};
  v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  v16f f0 = {0}, f1 = {0}, f2 = {0}, f3 = {0};
  v4f g0 = {0}, g1 = {0}, g2 = {0}, g3 = {0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, c3, 0, 0, 0);
    } else if (KIND == 1) {   // A fp4, B fp6
      f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f2, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f3, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else if (KIND == 2) {   // A fp6, B fp6
      f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f2, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f3, 2, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else if (KIND == 3) {   // A fp8, B fp8
      f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else if (KIND == 4) {   // A fp8 (spikes as e4m3), B fp6
      f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f2, 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f3, 0, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {                  // 16x16x128, A fp4, B fp6
      g0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, g0, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      g1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, g1, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      g2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, g2, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      g3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, g3, 4, 2, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += (float)(c0[r] + c1[r] + c2[r] + c3[r]) + f0[r] + f1[r] + f2[r] + f3[r];
  for (int r = 0; r < 4; ++r) s += g0[r] + g1[r] + g2[r] + g3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, int nblk) {
  float* out; long long* cyc; long long h[1024];
  hipMalloc(&out, nblk * 256 * 4); hipMalloc(&cyc, nblk * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_k<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_k<KIND>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, nblk * 8, hipMemcpyDeviceToHost);
  printf("%-28s blocks=%4d: %.1f memtime ticks per MFMA, wall %.3f ms -> %.2f ns per MFMA per wave\n", name, nblk,
         (double)h[0] / (4.0 * iters), ms, ms * 1e6 / (4.0 * iters));
  hipFree(out); hipFree(cyc);
}

int main() {
  uint8_t hA[32 * 64]; int8_t hB[64 * 32]; float hC[1024]; double ref[1024];
  srand(7);
  for (int dens = 0; dens < 2; ++dens) {
    for (int i = 0; i < 32 * 64; ++i) hA[i] = dens ? 1 : (rand() % 100 < 30);
    for (int i = 0; i < 64 * 32; ++i) hB[i] = dens ? (i & 1 ? 16 : 15) : (int8_t)(rand() % 33 - 16);
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0;
        for (int kk = 0; kk < 64; ++kk) s += hA[i * 64 + kk] * (double)hB[kk * 32 + j];
        ref[i * 32 + j] = s;
      }
    uint8_t* dA; int8_t* dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int reps : {1, 72, 1000}) {
      for (int sb : {0x7f7f7f7f, (int)0x82828282}) {
        hipLaunchKernelGGL(layout_k, dim3(1), dim3(64), 0, 0, dA, dB, dC, reps, sb);
        hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        const double mul = (sb == 0x7f7f7f7f ? 0.125 : 1.0) * reps;
        int bad = 0; double worst = 0;
        for (int i = 0; i < 1024; ++i) { double e = hC[i] - ref[i] * mul; if (e != 0) { ++bad; if (e < 0) e = -e; if (e > worst) worst = e; } }
        printf("dens=%d reps=%4d scale_b=%08x: %d mismatches of 1024 (worst abs err %g, |ref| max %g)\n", dens, reps, (unsigned)sb, bad,
               worst, ref[0] * mul);
      }
    }
  }
  for (int nblk : {1, 256}) {
    run<0>("i32_32x32x32_i8", nblk);
    run<1>("scale_32x32x64 A=fp4 B=fp6", nblk);
    run<2>("scale_32x32x64 A=fp6 B=fp6", nblk);
    run<3>("scale_32x32x64 A=fp8 B=fp8", nblk);
    run<4>("scale_32x32x64 A=fp8 B=fp6", nblk);
    run<5>("scale_16x16x128 A=fp4 B=fp6", nblk);
  }
  return 0;
}
