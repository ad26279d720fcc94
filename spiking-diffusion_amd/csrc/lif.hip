// Elementwise / scan kernels of the time-stepped SNN path at the reference's tensor interface
// ([T, N] fp32 views of [T,B,C,H,W]):
//   spk_lif_fwd        multi-step LIF scan          SJ/activation_based/neuron.py:799-811, :930-1011
//   spk_bn_prepare     eval-BN affine terms         aten batch_norm_kernel.cpp (pinned by fixture F7)
//   spk_bn_eval_fwd    y = fma(x, a[c], b[c])       SJ/activation_based/layer.py:458-465
//   spk_memout_fwd     sum_t x[t] * coef[t]         R/snn_model/snn_layers.py:36-41
//   spk_spikes_to_ptc / spk_ptc_to_spikes           layout converters fp32 [T,B,C,H,W] <-> u8 [B,H,W,T,C]
//
// All kernels are HBM-bound streaming kernels: 16 B per lane accesses, one pass over the data,
// membrane potential kept in VGPRs across the T loop.
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ LIF
// One thread owns VEC consecutive neurons and walks T with stride N.  Loads of a chunk of TU steps are
// issued back to back (TU x 16 B in flight per lane) before the dependent scan consumes them.
#ifndef SPK_LIF_TU
#define SPK_LIF_TU 8            // time steps whose loads are in flight together (x 16 B per lane)
#endif
#ifndef SPK_LIF_BLOCK
#define SPK_LIF_BLOCK 256
#endif
#ifndef SPK_LIF_GRID_PER_CU
#define SPK_LIF_GRID_PER_CU 32  // workgroups per CU at most (grid-stride beyond)
#endif
#ifndef SPK_LIF_NT
#define SPK_LIF_NT 3            // bit 0: non-temporal loads, bit 1: non-temporal stores
#endif
template <int VEC, int OUT, bool DIV>
__global__ __launch_bounds__(SPK_LIF_BLOCK) void lif_fwd_kernel(const float* __restrict__ x, float* __restrict__ v_io,
                                                      void* __restrict__ out, int T, long long N, float tau,
                                                      float inv_tau, float v_th, float v_reset) {
  constexpr int TU = SPK_LIF_TU;
  const long long ngroups = (N + VEC - 1) / VEC;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups;
       g += (long long)gridDim.x * blockDim.x) {
    const long long n0 = g * VEC;
    float v[VEC];
    if constexpr (VEC >= 4) {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        float4 t4 = *reinterpret_cast<const float4*>(v_io + n0 + 4 * q);
        v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
      }
    } else {
      v[0] = v_io[n0];
    }
    for (int t0 = 0; t0 < T; t0 += TU) {
      float xv[TU][VEC];
#pragma unroll
      for (int i = 0; i < TU; ++i) {
        if (t0 + i < T) {
          const float* p = x + (long long)(t0 + i) * N + n0;
          if constexpr (VEC >= 4) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
              f32x4 t4 = (SPK_LIF_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + q)
                                          : reinterpret_cast<const f32x4*>(p)[q];
              xv[i][4 * q] = t4.x; xv[i][4 * q + 1] = t4.y; xv[i][4 * q + 2] = t4.z; xv[i][4 * q + 3] = t4.w;
            }
          } else {
            xv[i][0] = __builtin_nontemporal_load(p);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TU; ++i) {
        if (t0 + i < T) {
          bool s[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) s[j] = spk_lif_step<DIV>(v[j], xv[i][j], tau, inv_tau, v_th, v_reset);
          const long long o = (long long)(t0 + i) * N + n0;
          if constexpr (OUT == SPK_SPIKE_F32) {
            float* po = reinterpret_cast<float*>(out) + o;
            if constexpr (VEC >= 4) {
#pragma unroll
              for (int q = 0; q < VEC / 4; ++q) {
                f32x4 r = {s[4 * q] ? 1.f : 0.f, s[4 * q + 1] ? 1.f : 0.f, s[4 * q + 2] ? 1.f : 0.f, s[4 * q + 3] ? 1.f : 0.f};
                if (SPK_LIF_NT & 2) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(po) + q);
                else reinterpret_cast<f32x4*>(po)[q] = r;
              }
            } else {
              po[0] = s[0] ? 1.f : 0.f;
            }
          } else {
            uint8_t* po = reinterpret_cast<uint8_t*>(out) + o;
            if constexpr (VEC >= 4) {
#pragma unroll
              for (int q = 0; q < VEC / 4; ++q) {
                uint32_t r = (uint32_t)s[4 * q] | ((uint32_t)s[4 * q + 1] << 8) | ((uint32_t)s[4 * q + 2] << 16) | ((uint32_t)s[4 * q + 3] << 24);
                reinterpret_cast<uint32_t*>(po)[q] = r;
              }
            } else {
              po[0] = (uint8_t)s[0];
            }
          }
        }
      }
    }
    if constexpr (VEC >= 4) {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q)
        *reinterpret_cast<float4*>(v_io + n0 + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
      v_io[n0] = v[0];
    }
  }
}

// Bit-packed spikes: lane l of a wave owns neuron 64*w + l, the wave ballot of "fired" IS the output word
// out[t][w] (bit l <-> neuron 64*w + l).  One u64 store per wave per time step.
template <bool DIV>
__global__ __launch_bounds__(256) void lif_fwd_bits_kernel(const float* __restrict__ x, float* __restrict__ v_io,
                                                           unsigned long long* __restrict__ out, int T, long long N,
                                                           float tau, float inv_tau, float v_th, float v_reset) {
  const long long words = (N + 63) / 64;
  const int lane = threadIdx.x & 63;
  for (long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < words;
       w += ((long long)gridDim.x * blockDim.x) >> 6) {
    const long long n = w * 64 + lane;
    const bool live = n < N;
    float v = live ? v_io[n] : 0.f;
    for (int t = 0; t < T; ++t) {
      float xv = live ? __builtin_nontemporal_load(x + (long long)t * N + n) : 0.f;
      bool s = spk_lif_step<DIV>(v, xv, tau, inv_tau, v_th, v_reset) && live;
      unsigned long long m = __ballot(s);
      if (lane == 0) out[(long long)t * words + w] = m;
    }
    if (live) v_io[n] = v;
  }
}


// The other eval forms of the reference neuron (SJ/activation_based/neuron.py:827-900: soft reset, decay_input = False, and
// the ..._with_v_seq variants that also return the membrane potential after every step).  Same one-pass structure as
// lif_fwd_kernel; one neuron per thread (these forms are off the benchmark path), the reference's operations one for one:
//   charge  decay_input:      v = v + (x - (v - v_reset)) / tau      (hard)      v = v + (x - v) / tau          (soft)
//           no decay_input:   v = v - (v - v_reset) / tau + x        (hard)      v = v * (1 - 1 / tau) + x      (soft)
//   reset   hard: v = v_reset * s + (1 - s) * v                      soft: v = v - s * v_th
// (1 - 1 / tau) is evaluated in double and rounded to fp32, as the scalar of the reference's tensor-scalar product is.
template <bool SOFT, bool DECAY, bool DIV>
__global__ __launch_bounds__(256) void lif_fwd_ex_kernel(const float* __restrict__ x, float* __restrict__ v_io,
                                                         float* __restrict__ out, float* __restrict__ v_seq, int T,
                                                         long long N, float tau, float inv_tau, float keep, float v_th,
                                                         float v_reset) {
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    float v = v_io[n];
    for (int t = 0; t < T; ++t) {
      const float xv = __builtin_nontemporal_load(x + (long long)t * N + n);
      float h;
      if (DECAY) {
        const float d = SOFT ? xv - v : xv - (v - v_reset);
        h = v + (DIV ? d / tau : d * inv_tau);
      } else if (SOFT) {
        h = v * keep + xv;
      } else {
        const float d = v - v_reset;
        h = (v - (DIV ? d / tau : d * inv_tau)) + xv;
      }
      const bool sp = h >= v_th;
      const float sf = sp ? 1.0f : 0.0f;
      v = SOFT ? h - sf * v_th : (sp ? (v_reset + 0.0f * h) : (v_reset * 0.0f + h));
      out[(long long)t * N + n] = sf;
      if (v_seq) v_seq[(long long)t * N + n] = v;
    }
    v_io[n] = v;
  }
}

// ------------------------------------------------------------------------------------------ BN
__global__ void bn_prepare_kernel(const float* gamma, const float* beta, const float* mean, const float* var,
                                  float eps, float* a, float* b, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float inv = 1.0f / sqrtf(var[c] + eps);      // correctly rounded fp32 div/sqrt (hipcc default)
  float g = gamma ? gamma[c] : 1.0f;
  float al = inv * g;
  a[c] = al;
  b[c] = fmaf(-mean[c], al, beta ? beta[c] : 0.0f);
}

// x: [M, C, HW] contiguous (M = T*B); 4 elements per lane when HW % 4 == 0.
template <int VEC>
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                      const float* __restrict__ b, float* __restrict__ y,
                                                      long long total, int C, int HW) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < total;
       i += (long long)gridDim.x * blockDim.x * VEC) {
    int c = (int)((i / HW) % C);
    float al = a[c], be = b[c];
    if constexpr (VEC == 4) {
      float4 t = *reinterpret_cast<const float4*>(x + i);
      t.x = fmaf(t.x, al, be); t.y = fmaf(t.y, al, be); t.z = fmaf(t.z, al, be); t.w = fmaf(t.w, al, be);
      *reinterpret_cast<float4*>(y + i) = t;
    } else {
      y[i] = fmaf(x[i], al, be);
    }
  }
}

// ------------------------------------------------------------------------------------------ memout
// out[n] = sum_t x[t][n] * coef[t], accumulated in t order (fp32, separate multiply and add as the
// reference's  torch.sum(x * coef, dim=0)).
template <int VEC>
__global__ __launch_bounds__(256) void memout_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                     float* __restrict__ out, int T, long long N) {
  __shared__ float sc[64];
  if (threadIdx.x < T && threadIdx.x < 64) sc[threadIdx.x] = coef[threadIdx.x];
  __syncthreads();
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < N;
       i += (long long)gridDim.x * blockDim.x * VEC) {
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int t = 0; t < T; ++t) {
      const float c = sc[t];
      if constexpr (VEC == 4) {
        float4 v = *reinterpret_cast<const float4*>(x + (long long)t * N + i);
        acc[0] = acc[0] + v.x * c; acc[1] = acc[1] + v.y * c; acc[2] = acc[2] + v.z * c; acc[3] = acc[3] + v.w * c;
      } else {
        acc[0] = acc[0] + x[(long long)t * N + i] * c;
      }
    }
    if constexpr (VEC == 4) {
      *reinterpret_cast<float4*>(out + i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      out[i] = acc[0];
    }
  }
}

// ------------------------------------------------------------------------------------------ layout
// fp32 spikes [T,B,C,H,W]  ->  u8 [B,H,W,T,C]   (one thread per (b, hw, t, c); reads strided, writes coalesced)
__global__ __launch_bounds__(256) void spikes_to_ptc_kernel(const float* __restrict__ s, uint8_t* __restrict__ o,
                                                            int T, int B, int C, int HW, int chunk) {
  // output memory order: [B][C/chunk][HW][T][chunk]  (chunk == C: plain PTC [B][HW][T][C])
  long long total = (long long)T * B * C * HW;
  const int nch = C / chunk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int cc = (int)(i % chunk);
    long long r = i / chunk;
    int t = (int)(r % T); r /= T;
    int hw = (int)(r % HW); r /= HW;
    int c = (int)(r % nch) * chunk + cc;
    int b = (int)(r / nch);
    float f = s[(((long long)t * B + b) * C + c) * HW + hw];
    o[i] = f != 0.0f ? 1 : 0;
  }
}

// u8 [B,H,W,T,C] -> fp32 [T,B,C,H,W]   (one thread per output element)
__global__ __launch_bounds__(256) void ptc_to_spikes_kernel(const uint8_t* __restrict__ s, float* __restrict__ o,
                                                            int T, int B, int C, int HW, int chunk) {
  long long total = (long long)T * B * C * HW;
  const int nch = C / chunk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int hw = (int)(i % HW);
    long long r = i / HW;
    int c = (int)(r % C); r /= C;
    int b = (int)(r % B);
    int t = (int)(r / B);
    o[i] = (float)s[((((long long)b * nch + c / chunk) * HW + hw) * T + t) * chunk + c % chunk];
  }
}

inline int grid_for(long long work_items) {
  long long g = (work_items + 255) / 256;
  const long long cap = 256 * 8 * 4;    // 256 CUs x 8 blocks, x4 for tail balance; grid-stride the rest
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline int lif_grid(long long work_items) {
  long long g = (work_items + SPK_LIF_BLOCK - 1) / SPK_LIF_BLOCK;
  const long long cap = 256ll * SPK_LIF_GRID_PER_CU;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int spk_lif_fwd(const float* x_seq, float* v_inout, void* spike_out, int T, long long N, float tau,
                           float v_threshold, float v_reset, int spike_dtype, hipStream_t stream) {
  if (!x_seq || !v_inout || !spike_out || T <= 0 || N <= 0 || !(tau > 0.f)) return SPK_ERR_ARG;
  int ex;
  float m = frexpf(tau, &ex);
  const bool pow2 = (m == 0.5f);
  const float inv_tau = 1.0f / tau;
  if (spike_dtype == SPK_SPIKE_BITS) {
    long long words = (N + 63) / 64;
    int grid = grid_for(words * 64);
    if (pow2)
      hipLaunchKernelGGL(lif_fwd_bits_kernel<false>, dim3(grid), dim3(256), 0, stream, x_seq, v_inout,
                         (unsigned long long*)spike_out, T, N, tau, inv_tau, v_threshold, v_reset);
    else
      hipLaunchKernelGGL(lif_fwd_bits_kernel<true>, dim3(grid), dim3(256), 0, stream, x_seq, v_inout,
                         (unsigned long long*)spike_out, T, N, tau, inv_tau, v_threshold, v_reset);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  if (spike_dtype != SPK_SPIKE_F32 && spike_dtype != SPK_SPIKE_U8) return SPK_ERR_UNSUPPORTED;
  const bool f32 = spike_dtype == SPK_SPIKE_F32;
  const uintptr_t al = (uintptr_t)x_seq | (uintptr_t)v_inout | (uintptr_t)spike_out;
  const bool vec = (N % 4 == 0) && (al % 16 == 0);
#ifndef SPK_LIF_VEC8
#define SPK_LIF_VEC8 0          // 1: eight neurons per thread (two 16-byte requests per lane and step) where N % 8 == 0.  Round 6, same box:
                                // 3.7-3.8 TB/s against 6.4 (a wave's request then covers every other 16 bytes of 2 KB): not instantiated
#endif
  [[maybe_unused]] const bool vec8 = SPK_LIF_VEC8 && vec && (N % 8 == 0) && (al % 32 == 0);
#define SPK_LIF_LAUNCH(VEC, OUT, DIV)                                                                              \
  hipLaunchKernelGGL((lif_fwd_kernel<VEC, OUT, DIV>), dim3(lif_grid((N + VEC - 1) / VEC)), dim3(SPK_LIF_BLOCK), 0, stream, \
                     x_seq, v_inout, spike_out, T, N, tau, inv_tau, v_threshold, v_reset)
#if SPK_LIF_VEC8
  if (vec8) {
    if (f32) { if (pow2) SPK_LIF_LAUNCH(8, SPK_SPIKE_F32, false); else SPK_LIF_LAUNCH(8, SPK_SPIKE_F32, true); }
    else     { if (pow2) SPK_LIF_LAUNCH(8, SPK_SPIKE_U8, false);  else SPK_LIF_LAUNCH(8, SPK_SPIKE_U8, true); }
  } else
#endif
  if (vec) {
    if (f32) { if (pow2) SPK_LIF_LAUNCH(4, SPK_SPIKE_F32, false); else SPK_LIF_LAUNCH(4, SPK_SPIKE_F32, true); }
    else     { if (pow2) SPK_LIF_LAUNCH(4, SPK_SPIKE_U8, false);  else SPK_LIF_LAUNCH(4, SPK_SPIKE_U8, true); }
  } else {
    if (f32) { if (pow2) SPK_LIF_LAUNCH(1, SPK_SPIKE_F32, false); else SPK_LIF_LAUNCH(1, SPK_SPIKE_F32, true); }
    else     { if (pow2) SPK_LIF_LAUNCH(1, SPK_SPIKE_U8, false);  else SPK_LIF_LAUNCH(1, SPK_SPIKE_U8, true); }
  }
#undef SPK_LIF_LAUNCH
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_lif_fwd_ex(const float* x_seq, float* v_inout, float* spike_out_f32, float* v_seq_out_or_null, int T,
                              long long N, float tau, float v_threshold, float v_reset, int soft_reset, int decay_input,
                              hipStream_t stream) {
  if (!x_seq || !v_inout || !spike_out_f32 || T <= 0 || N <= 0 || !(tau > 0.f)) return SPK_ERR_ARG;
  int ex;
  const bool pow2 = frexpf(tau, &ex) == 0.5f;
  const float inv_tau = 1.0f / tau, keep = (float)(1.0 - 1.0 / (double)tau);
  const int grid = grid_for(N);
#define SPK_LIFX(SOFT, DECAY, DIV)                                                                                   \
  hipLaunchKernelGGL((lif_fwd_ex_kernel<SOFT, DECAY, DIV>), dim3(grid), dim3(256), 0, stream, x_seq, v_inout,        \
                     spike_out_f32, v_seq_out_or_null, T, N, tau, inv_tau, keep, v_threshold, v_reset)
  if (soft_reset) {
    if (decay_input) { if (pow2) SPK_LIFX(true, true, false); else SPK_LIFX(true, true, true); }
    else SPK_LIFX(true, false, false);
  } else {
    if (decay_input) { if (pow2) SPK_LIFX(false, true, false); else SPK_LIFX(false, true, true); }
    else { if (pow2) SPK_LIFX(false, false, false); else SPK_LIFX(false, false, true); }
  }
#undef SPK_LIFX
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_bn_prepare(const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, float eps, float* a_out, float* b_out, int C,
                              hipStream_t stream) {
  if (!running_mean || !running_var || !a_out || !b_out || C <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((C + 63) / 64), dim3(64), 0, stream, gamma, beta, running_mean,
                     running_var, eps, a_out, b_out, C);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_bn_eval_fwd(const float* x, const float* a, const float* b, float* y, long long M, int C, int HW,
                               hipStream_t stream) {
  if (!x || !a || !b || !y || M <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  long long total = M * C * HW;
  const bool vec = (HW % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y) % 16) == 0);
  if (vec)
    hipLaunchKernelGGL(bn_eval_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, stream, x, a, b, y, total, C, HW);
  else
    hipLaunchKernelGGL(bn_eval_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, x, a, b, y, total, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_memout_fwd(const float* x_seq, const float* coef, float* out, int T, long long N,
                              hipStream_t stream) {
  if (!x_seq || !coef || !out || T <= 0 || T > 64 || N <= 0) return SPK_ERR_ARG;
  const bool vec = (N % 4 == 0) && ((((uintptr_t)x_seq | (uintptr_t)out) % 16) == 0);
  if (vec)
    hipLaunchKernelGGL(memout_kernel<4>, dim3(grid_for(N / 4)), dim3(256), 0, stream, x_seq, coef, out, T, N);
  else
    hipLaunchKernelGGL(memout_kernel<1>, dim3(grid_for(N)), dim3(256), 0, stream, x_seq, coef, out, T, N);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_spikes_to_ptc(const float* spikes_tbchw, uint8_t* out_bhwtc, int T, int B, int C, int HW,
                                 int chunk, hipStream_t stream) {
  if (!spikes_tbchw || !out_bhwtc || T <= 0 || B <= 0 || C <= 0 || HW <= 0 || chunk <= 0 || C % chunk)
    return SPK_ERR_ARG;
  hipLaunchKernelGGL(spikes_to_ptc_kernel, dim3(grid_for((long long)T * B * C * HW)), dim3(256), 0, stream,
                     spikes_tbchw, out_bhwtc, T, B, C, HW, chunk);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_ptc_to_spikes(const uint8_t* in_bhwtc, float* spikes_tbchw, int T, int B, int C, int HW,
                                 int chunk, hipStream_t stream) {
  if (!in_bhwtc || !spikes_tbchw || T <= 0 || B <= 0 || C <= 0 || HW <= 0 || chunk <= 0 || C % chunk)
    return SPK_ERR_ARG;
  hipLaunchKernelGGL(ptc_to_spikes_kernel, dim3(grid_for((long long)T * B * C * HW)), dim3(256), 0, stream,
                     in_bhwtc, spikes_tbchw, T, B, C, HW, chunk);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
