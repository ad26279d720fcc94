// Probe: sustained rate of the fp6 kernel's K loop in isolation (no DMA, no epilogue) under register-file variants.
//   -DNVACC=n : accumulators 3*i+j >= 21-n live in VGPRs (inline asm "v"), the rest in AGPRs
//   -DNOLDS   : operands stay in registers (no ds_read)      -DNONOP : no s_nop before the MFMA
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
#ifndef NVACC
#define NVACC 5
#endif
#ifndef NT
#define NT 7
#endif
#ifdef NONOP
#define PRE ""
#else
#define PRE "s_nop 1\n\t"
#endif
#define MFMA(CLS, acc, a4, b6, sa, sb) asm volatile(PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2" : "+" CLS(acc) : "v"(a4), "v"(b6), "v"(sa), "v"(sb))
__global__ __launch_bounds__(256, 1) void loopk(const uint8_t* in, float* out, int nchunks, int PW, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 150000 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(in)[i];
  int a_off[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) a_off[i] = (((wave + 4 * i) * 2 + (lane >> 2 & 1)) % 60 + 9) * 512 + (lane & 31) * 16;
  v16f acc[NT][3];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
  int dummy = 0;
  typedef int v4ii __attribute__((ext_vector_type(4)));
  v4ii rsrc; { const unsigned long long b = (unsigned long long)in; rsrc[0] = (int)(unsigned)b; rsrc[1] = (int)(unsigned)(b >> 32); rsrc[2] = 0x7fffffff; rsrc[3] = 0x00020000; }
  asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) void*)(lds + 120000 + wave * 1024)))) : "m0");
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    const uint8_t* A = lds + (c & 1) * 37376;
    const uint8_t* Wb = lds + 80000 + (c & 1) * 1536;
    auto lda = [&](int s) -> v4i {
      const int tap = s / NT, i = s % NT;
      const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * 512;
#ifdef NOLDS
      v4i r = {a_off[i], toff, s, lane};
      return r;
#else
      return *reinterpret_cast<const v4i*>(A + a_off[i] + toff);
#endif
    };
    auto ldb = [&](int tap, int j) -> v6i {
#ifdef NOLDS
      v6i r = {tap, j, lane, 1, 2, 3};
      return r;
#else
      const uint8_t* p = Wb + (tap * 3 + j) * 1536;
      const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
      const v2i y = *reinterpret_cast<const v2i*>(p + 1024 + lane * 8);
      const v6i r = {x[0], x[1], x[2], x[3], y[0], y[1]};
      return r;
#endif
    };
    v6i bc[3], bn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { bc[j] = ldb(0, j); bn[j] = bc[j]; }
    constexpr int PF = 4;
    v4i af[PF];
#pragma unroll
    for (int s = 0; s < PF; ++s) af[s] = lda(s);
#pragma unroll
    for (int s = 0; s < 9 * NT; ++s) {
      const int tap = s / NT, i = s % NT;
      const v4i a4 = af[s % PF];
      if (s + PF < 9 * NT) af[s % PF] = lda(s + PF);
      if (i == 0 && tap + 1 < 9) {
#pragma unroll
        for (int j = 0; j < 3; ++j) bn[j] = ldb(tap + 1, j);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (3 * i + j < 3 * NT - NVACC) MFMA("a", acc[i][j], a4, bc[j], sa, sb); else MFMA("v", acc[i][j], a4, bc[j], sa, sb);
      }
      if (i == NT - 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) bc[j] = bn[j];
      }
#ifdef DMAV
      if (s % 3 == 0 && s / 3 < 18) {
        const unsigned ldsb = (unsigned)(size_t)((__attribute__((address_space(3))) void*)(lds + 120000 + wave * 1024));
        const unsigned voff = lane * 16 + (s / 3) * 1024 + wave * 20480;
#if DMAV == 1
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(in), "s"(__builtin_amdgcn_readfirstlane(ldsb + (s & 1) * 4096)) : "memory", "m0");
#elif DMAV == 2
        asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(in) : "memory");
#elif DMAV == 3
        v4i tmp;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(tmp) : "v"(voff), "s"(in) : "memory");
        dummy += tmp[0];
#elif DMAV == 4
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(in), "s"(__builtin_amdgcn_readfirstlane(ldsb + (s & 1) * 4096)) : "memory", "m0");
#elif DMAV == 5
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane(ldsb + (s & 1) * 4096)) : "memory", "m0");
#endif
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef DMAV
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float s = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (3 * i + j < 3 * NT - NVACC) asm volatile("" : "+a"(acc[i][j]));
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r] * (float)(r + j);
    }
    out[(blockIdx.x * NT + i) * 256 + threadIdx.x] = s + dummy;
  }
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  uint8_t* in; float* out; long long* cyc; long long h[256];
  hipMalloc(&in, 1 << 20); hipMemset(in, 0x22, 1 << 20); hipMalloc(&out, 256 * NT * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int nchunks = 2000;
  hipFuncSetAttribute((const void*)loopk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(loopk, dim3(256), dim3(256), 158 * 1024, 0, in, out, nchunks, 8, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)nchunks * 9 * NT * 3;
    printf("NT=%d NVACC=%d: %.3f ms, %.2f ns per MFMA per wave, %.3f us per chunk (%d MFMAs), err=%d\n", NT, NVACC, ms,
           ms * 1e6 / n, ms * 1e3 / nchunks, 9 * NT * 3, (int)hipGetLastError());
  }
  return 0;
}
