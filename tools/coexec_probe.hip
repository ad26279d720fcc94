// How much vector work rides for free next to the matrix cores on one SIMD of gfx950?
//
// Every kernel of this repo that is not at its MFMA rate carries vector instructions next to its MFMAs (LIF scan, certification,
// operand splits).  PMC shows SQ_VALU_MFMA_COEXEC_CYCLES at 2-6 % of the MFMA-busy cycles in all of them; this probe measures the
// thing itself: shader cycles (s_memtime) per loop iteration of
//   M     four independent MFMAs                                         (the matrix pipe's own rate)
//   V     4 x NV independent v_fma_f32                                   (the vector pipe's own rate)
//   MV    four MFMAs, each followed by NV v_fma_f32 in the SAME wave      (same-wave overlap)
//   M|V   two waves per SIMD: one issues only the MFMAs, the other only the vector work (cross-wave overlap)
// for v_mfma_f32_32x32x16_bf16 and v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 x fp4).  All instructions are volatile inline asm: the
// issue order is the source order.   build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/coexec tools/coexec_probe.hip && /tmp/coexec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void mfma(v16f& c, const v8i& a, const v8i& b) {
  if (KIND == 0) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(*reinterpret_cast<const v4i*>(&a)), "v"(*reinterpret_cast<const v4i*>(&b)));
  } else {
    // fp4 (A, 4 registers) x fp6 (B, 6 registers), scales 1.0: the denoiser kernel's instruction and modifiers
    v4i a4 = {b[0], b[1], b[2], b[3]};
    v6i b6 = {a[0], a[1], a[2], a[3], a[4], a[5]};
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:2"
                 : "+v"(c) : "v"(a4), "v"(b6), "v"(KIND == 2 ? (int)0x87828782u : 0x7f7f7f7f));
  }
}

template <int NV>
__device__ __forceinline__ void valu(float (&x)[8], float m, float d) {
#pragma unroll
  for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(m), "v"(d));
}

// MODE 0: M, 1: V, 2: MV (same wave), 3: M|V (waves 0..3 MFMA, waves 4..7 vector; launch 512 threads)
template <int KIND, int MODE, int NV>
__global__ __launch_bounds__(1024, 1) void probe(unsigned long long* out, int iters) {
  unsigned h = 0x9E3779B9u * (threadIdx.x + 1u) + 0x85EBCA6Bu * (blockIdx.x + 1u);
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = KIND ? (int)(h & 0x6DB6DB6Du) : (int)(h & 0x3F803F80u);
    h = h * 1664525u + 1013904223u; b[i] = KIND ? (int)(h & 0x22222222u) : (int)(h & 0x3F803F80u);
  }
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 1.f + 1e-3f * j;
  const float m = 0.999f, d = 1e-3f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && threadIdx.x < 256);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && threadIdx.x >= 256);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 4) {           // accumulation chains: NV accumulators in rotation (1 = every MFMA waits for the one before it)
    for (int i = 0; i < iters; ++i) {
      if (NV == 1) { mfma<KIND>(c0, a, b); mfma<KIND>(c0, a, b); mfma<KIND>(c0, a, b); mfma<KIND>(c0, a, b); }
      if (NV == 2) { mfma<KIND>(c0, a, b); mfma<KIND>(c1, a, b); mfma<KIND>(c0, a, b); mfma<KIND>(c1, a, b); }
      if (NV == 3) { mfma<KIND>(c0, a, b); mfma<KIND>(c1, a, b); mfma<KIND>(c2, a, b); mfma<KIND>(c0, a, b); }
    }
  } else if (MODE == 3) {
    if (do_m) for (int i = 0; i < iters; ++i) { mfma<KIND>(c0, a, b); mfma<KIND>(c1, a, b); mfma<KIND>(c2, a, b); mfma<KIND>(c3, a, b); }
    else for (int i = 0; i < iters; ++i) { valu<NV>(x, m, d); valu<NV>(x, m, d); valu<NV>(x, m, d); valu<NV>(x, m, d); }
  } else {
    for (int i = 0; i < iters; ++i) {
      if (do_m) mfma<KIND>(c0, a, b);
      if (do_v) valu<NV>(x, m, d);
      if (do_m) mfma<KIND>(c1, a, b);
      if (do_v) valu<NV>(x, m, d);
      if (do_m) mfma<KIND>(c2, a, b);
      if (do_v) valu<NV>(x, m, d);
      if (do_m) mfma<KIND>(c3, a, b);
      if (do_v) valu<NV>(x, m, d);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  for (int j = 0; j < 8; ++j) s += x[j];
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
    out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
  }
  if (s == 123456.789f) out[0] = 1;
}

// Vector instruction kinds of the LIF scans: cycles per wave instruction at 2 / 4 waves per SIMD (8 independent chains per wave)
//   0 v_fma_f32   1 v_pk_fma_f32   2 v_max_f32 with |abs|   3 v_cndmask_b32 (vcc)   4 v_cmp_le_f32 (-> vcc)
//   5 the spike-under-mask sequence of vae_fp6 (s_mov sv, exec; v_cmpx_le; v_mov; v_add; s_mov exec, sv: 3 vector + 2 scalar)
//   6 v_sub_f32 + v_fma_f32 dependent pair   7 v_mov_b32
template <int K>
__global__ __launch_bounds__(1024, 1) void valu_probe(unsigned long long* out, int iters) {
  float x[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f y[8];
  for (int j = 0; j < 8; ++j) { x[j] = 1.f + 1e-3f * (j + threadIdx.x % 7); y[j] = (v2f){x[j], x[j] + 0.5f}; }
  const float m = 0.999f, d = 1e-3f;
  const v2f m2 = {m, m}, d2 = {d, d};
  float acc = 0.f, tmp = 0.f;
  const unsigned long long mask = 0x5555555555555555ull + blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float& r = x[j & 7];
      if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(d));
      if (K == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[j & 7]) : "v"(m2), "v"(d2));
      if (K == 2) asm volatile("v_max_f32 %0, %0, |%1|" : "+v"(r) : "v"(d));
      if (K == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(d));
      if (K == 4) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(r), "v"(d) : "vcc");
      if (K == 5) { unsigned long long sv; asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_le_f32_e32 1.0, %[v]\n\tv_mov_b32_e32 %[v], 0\n\tv_add_f32_e32 %[a], %[c], %[a]\n\ts_mov_b64 exec, %[sv]" : [v] "+v"(r), [a] "+v"(acc), [sv] "=&s"(sv) : [c] "s"(d) : "vcc"); }
      if (K == 6) { float t; asm volatile("v_sub_f32 %1, %2, %0\n\tv_fma_f32 %0, %1, 0.5, %0" : "+v"(r), "=&v"(t) : "v"(d)); }
      if (K == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(d));
      if (K == 8) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r) : "v"(d), "s"(mask));
      if (K == 9) { unsigned long long c; asm volatile("v_cmp_le_f32_e64 %1, 1.0, %0\n\tv_cndmask_b32_e64 %0, %0, 0, %1" : "+v"(r), "=&s"(c)); }
      if (K == 10) { unsigned long long c; asm volatile("v_cmp_le_f32_e64 %1, 1.0, %0\n\tv_cndmask_b32_e64 %0, %0, 0, %1\n\tv_cndmask_b32_e64 %3, 0, %4, %1\n\tv_add_f32 %2, %2, %3" : "+v"(r), "=&s"(c), "+v"(acc), "=&v"(tmp) : "v"(d)); }
      if (K == 11) asm volatile("v_cmp_le_f32 vcc, 1.0, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(d) : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc;
  for (int j = 0; j < 8; ++j) s += x[j] + y[j][0] + y[j][1];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (s == 123456.789f) out[0] = 1;
}

template <int K>
static void run_valu(const char* label, int ninstr, unsigned long long* dev, int iters) {
  printf("VALU %-44s", label);
  for (int wps : {1, 2, 3, 4}) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((valu_probe<K>), dim3(256), dim3(256 * wps), 0, 0, dev, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4 * wps; ++w) cyc.push_back((double)h[b * 16 + w]);
    std::sort(cyc.begin(), cyc.end());
    printf("  %dw/SIMD %5.2f", wps, cyc[cyc.size() / 2] / ((double)iters * 16 * ninstr) / wps);
  }
  printf("   cycles of the SIMD per vector instruction\n");
}

// LDS read bandwidth: every wave reads `iters` x 8 conflict-free vectors of WIDTH bytes per lane from a 64 KB window
template <int WIDTH>
__global__ __launch_bounds__(1024, 1) void lds_probe(unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<int*>(lds)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * WIDTH + (threadIdx.x >> 6) * 64 * WIDTH;
  v4i s4 = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned addr = (base + j * 4096) & 0xFFFFu;
      if (WIDTH == 16) { v4i v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); s4[0] ^= 0; (void)v; }
      else if (WIDTH == 8) { long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); (void)v; }
      else { int v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); (void)v; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (s4[0] == 12345) out[0] = 1;
}

template <int WIDTH>
static void run_lds(int waves, unsigned long long* dev, int iters) {
  const int nb = 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((lds_probe<WIDTH>), dim3(nb), dim3(64 * waves), 65536, 0, dev, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nb * 16);
  hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc;
  for (int b = 0; b < nb; ++b) for (int w = 0; w < waves; ++w) cyc.push_back((double)h[b * 16 + w]);
  std::sort(cyc.begin(), cyc.end());
  const double c = cyc[cyc.size() / 2];
  printf("LDS  ds_read_b%-3d %2d waves/CU: %7.1f bytes per cycle and CU  (%.1f cycles per wave instruction)\n", WIDTH * 8, waves,
         (double)waves * iters * 8 * 64 * WIDTH / c, c / (iters * 8.0));
}

template <int KIND, int MODE, int NV>
static void run(const char* label, int waves_per_simd, unsigned long long* dev, int iters) {
  const int nb = 256, threads = MODE == 3 ? 512 : 256 * waves_per_simd;
  hipMemset(dev, 0, nb * 16 * 2 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<KIND, MODE, NV>), dim3(nb), dim3(threads), 0, 0, dev, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(nb * 16 * 2);
  hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, ghz;
  const int nw = threads / 64;
  for (int bidx = 0; bidx < nb; ++bidx)
    for (int w = 0; w < nw; ++w) {
      cyc.push_back((double)h[2 * (bidx * 16 + w)] / iters);
      ghz.push_back((double)h[2 * (bidx * 16 + w)] / (double)h[2 * (bidx * 16 + w) + 1] * 0.1);
    }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  printf("%-58s %8.1f cycles per iteration (4 MFMA%s)   clock %.2f GHz\n", label, cyc[cyc.size() / 2], MODE == 0 ? "" : " / 4 x NV fma", ghz[ghz.size() / 2]);
}

template <int KIND>
static void all(unsigned long long* dev, int iters) {
  printf("== %s\n", KIND == 2 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 x fp4), scales 2^8 / 2^3" : KIND ? "v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 x fp4)" : "v_mfma_f32_32x32x16_bf16");
  run<KIND, 0, 0>("M    1 wave/SIMD", 1, dev, iters);
  run<KIND, 0, 0>("M    2 waves/SIMD (each 4 MFMA per iteration)", 2, dev, iters);
  run<KIND, 4, 1>("M    1 wave/SIMD, ONE accumulator (dependent chain)", 1, dev, iters);
  run<KIND, 4, 2>("M    1 wave/SIMD, two accumulators in rotation", 1, dev, iters);
  run<KIND, 4, 1>("M    2 waves/SIMD, ONE accumulator each", 2, dev, iters);
  run<KIND, 4, 2>("M    2 waves/SIMD, two accumulators each", 2, dev, iters);
  run<KIND, 1, 4>("V    1 wave/SIMD, NV = 4", 1, dev, iters);
  run<KIND, 1, 8>("V    1 wave/SIMD, NV = 8", 1, dev, iters);
  run<KIND, 1, 8>("V    2 waves/SIMD, NV = 8", 2, dev, iters);
  run<KIND, 1, 8>("V    3 waves/SIMD, NV = 8", 3, dev, iters);
  run<KIND, 1, 8>("V    4 waves/SIMD, NV = 8", 4, dev, iters);
  run<KIND, 2, 8>("MV   4 waves/SIMD, NV = 8", 4, dev, iters);
  run<KIND, 2, 2>("MV   1 wave/SIMD, NV = 2", 1, dev, iters);
  run<KIND, 2, 4>("MV   1 wave/SIMD, NV = 4", 1, dev, iters);
  run<KIND, 2, 6>("MV   1 wave/SIMD, NV = 6", 1, dev, iters);
  run<KIND, 2, 8>("MV   1 wave/SIMD, NV = 8", 1, dev, iters);
  run<KIND, 2, 4>("MV   2 waves/SIMD, NV = 4", 2, dev, iters);
  run<KIND, 2, 8>("MV   2 waves/SIMD, NV = 8", 2, dev, iters);
  run<KIND, 3, 4>("M|V  2 waves/SIMD (one MFMA, one vector), NV = 4", 2, dev, iters);
  run<KIND, 3, 8>("M|V  2 waves/SIMD (one MFMA, one vector), NV = 8", 2, dev, iters);
  run<KIND, 3, 16>("M|V  2 waves/SIMD (one MFMA, one vector), NV = 16", 2, dev, iters);
}

int main() {
  unsigned long long* dev;
  hipMalloc(&dev, 256 * 16 * 2 * 8);
  run_valu<0>("v_fma_f32", 1, dev, 2000);
  run_valu<1>("v_pk_fma_f32", 1, dev, 2000);
  run_valu<2>("v_max_f32 x, x, |y|", 1, dev, 2000);
  run_valu<3>("v_cndmask_b32", 1, dev, 2000);
  run_valu<4>("v_cmp_le_f32 -> vcc", 1, dev, 2000);
  run_valu<5>("spike under mask (3 vector + 2 scalar)", 3, dev, 2000);
  run_valu<6>("v_sub_f32 + dependent v_fma_f32", 2, dev, 2000);
  run_valu<7>("v_mov_b32", 1, dev, 2000);
  run_valu<8>("v_cndmask_b32_e64 (mask in an SGPR pair)", 1, dev, 2000);
  run_valu<9>("v_cmp_e64 -> SGPR pair + v_cndmask_e64", 2, dev, 2000);
  run_valu<10>("compare + two selects + add (4 vector)", 4, dev, 2000);
  run_valu<11>("v_cmp -> vcc + v_cndmask vcc", 2, dev, 2000);
  for (int w : {4, 8, 16}) { run_lds<16>(w, dev, 2000); run_lds<8>(w, dev, 2000); run_lds<4>(w, dev, 2000); }
  all<0>(dev, 2000);
  all<1>(dev, 2000);
  all<2>(dev, 2000);
  return 0;
}
