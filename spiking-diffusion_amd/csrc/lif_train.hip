// Surrogate-gradient LIF for the training path: multi-step forward that keeps the membrane
// potential before reset (h) for the backward, and the BPTT backward with the ATan surrogate.
//
// Reference semantics: LIFNode training forward, torch backend -- neuronal_charge / neuronal_fire / neuronal_reset
// (SJ/activation_based/neuron.py:739-749 charge with decay_input, :133-135 hard reset, surrogate.atan
// SJ/activation_based/surrogate.py:664-678); native contract mirrored: the CuPy pair LIFNodeFPTTKernel /
// LIFNodeBPTTKernel (SJ/activation_based/auto_cuda/neuron_kernel.py:102-225,479-540, ATan derivative
// auto_cuda/cfunction.py:252-258).  Hard reset, decay_input = True (the configuration of snn_model).
//
//   forward, t = 0..T-1:   h_t = v + (x_t - (v - v_reset)) / tau;  s_t = h_t - v_th >= 0;  v = (1 - s_t) h_t + s_t v_reset
//   backward, t = T-1..0:  g_s = alpha/2 / (1 + (pi/2 alpha (h_t - v_th))^2)
//                          dv/dh = (1 - s_t) + (v_reset - h_t) g_s        (second term dropped when detach_reset)
//                          dL/dh_t = G dv/dh + dL/ds_t g_s;   dL/dx_t = dL/dh_t / tau;   G <- dL/dh_t (1 - 1/tau)
//                          with G = dL/dv_T (gradient of the final state) before the loop and dL/dv_init = G after it.
// Both kernels stream: one thread owns 4 consecutive neurons, T steps in registers; HBM-bound (fwd 4 B in + 8 B out
// per neuron-step, bwd 8 B in + 4 B out).
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

inline int grid_for(long long work_items) {
  long long g = (work_items + 255) / 256;
  const long long cap = 256 * 8 * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

template <int VEC>
__global__ __launch_bounds__(256) void lif_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ v_init,
                                                            float* __restrict__ h_seq, float* __restrict__ s_seq,
                                                            float* __restrict__ v_out, int T, long long N, float tau,
                                                            float v_th, float v_reset) {
  const long long ngroups = (N + VEC - 1) / VEC;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
    const long long n0 = g * VEC;
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = v_init[n0 + k];
    for (int t = 0; t < T; ++t) {
      float xv[VEC], hv[VEC], sv[VEC];
      const float* p = x + (long long)t * N + n0;
      if constexpr (VEC == 4) {
        const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w;
      } else {
        xv[0] = *p;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float h = v[k] + (xv[k] - (v[k] - v_reset)) / tau;
        const float s = (h - v_th >= 0.0f) ? 1.0f : 0.0f;
        v[k] = (1.0f - s) * h + s * v_reset;
        hv[k] = h; sv[k] = s;
      }
      float* ph = h_seq + (long long)t * N + n0;
      float* ps = s_seq + (long long)t * N + n0;
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(ph) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        *reinterpret_cast<float4*>(ps) = make_float4(sv[0], sv[1], sv[2], sv[3]);
      } else {
        *ph = hv[0]; *ps = sv[0];
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) v_out[n0 + k] = v[k];
  }
}

template <int VEC, bool DETACH>
__global__ __launch_bounds__(256) void lif_train_bwd_kernel(const float* __restrict__ grad_s, const float* __restrict__ grad_v_last,
                                                            const float* __restrict__ h_seq, float* __restrict__ grad_x,
                                                            float* __restrict__ grad_v_init, int T, long long N, float tau,
                                                            float v_th, float v_reset, float alpha) {
  const long long ngroups = (N + VEC - 1) / VEC;
  const float inv_tau = 1.0f / tau, carry = 1.0f - inv_tau;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
    const long long n0 = g * VEC;
    float G[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) G[k] = grad_v_last ? grad_v_last[n0 + k] : 0.0f;
    for (int t = T - 1; t >= 0; --t) {
      float gs[VEC], hv[VEC], gx[VEC];
      const float* pg = grad_s + (long long)t * N + n0;
      const float* ph = h_seq + (long long)t * N + n0;
      if constexpr (VEC == 4) {
        const f32x4 a4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pg));
        const f32x4 b4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ph));
        gs[0] = a4.x; gs[1] = a4.y; gs[2] = a4.z; gs[3] = a4.w;
        hv[0] = b4.x; hv[1] = b4.y; hv[2] = b4.z; hv[3] = b4.w;
      } else {
        gs[0] = *pg; hv[0] = *ph;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float over = hv[k] - v_th;
        const float s = over >= 0.0f ? 1.0f : 0.0f;
        const float ax = 1.57079632679489661923f * alpha * over;
        const float g_s = alpha / 2.0f / (1.0f + ax * ax);
        float dv_dh = 1.0f - s;
        if (!DETACH) dv_dh = (v_reset - hv[k]) * g_s + dv_dh;
        const float gh = G[k] * dv_dh + gs[k] * g_s;
        gx[k] = gh * inv_tau;
        G[k] = gh * carry;
      }
      float* px = grad_x + (long long)t * N + n0;
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(px) = make_float4(gx[0], gx[1], gx[2], gx[3]);
      else *px = gx[0];
    }
    if (grad_v_init) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) grad_v_init[n0 + k] = G[k];
    }
  }
}

}  // namespace

extern "C" int spk_lif_train_fwd(const float* x_seq, const float* v_init, float* h_seq, float* spike_seq, float* v_out,
                                 int T, long long N, float tau, float v_threshold, float v_reset, hipStream_t stream) {
  if (!x_seq || !v_init || !h_seq || !spike_seq || !v_out || T <= 0 || N <= 0 || !(tau > 0.f)) return SPK_ERR_ARG;
  const uintptr_t al = (uintptr_t)x_seq | (uintptr_t)v_init | (uintptr_t)h_seq | (uintptr_t)spike_seq | (uintptr_t)v_out;
  if ((N % 4 == 0) && (al % 16 == 0))
    hipLaunchKernelGGL(lif_train_fwd_kernel<4>, dim3(grid_for(N / 4)), dim3(256), 0, stream, x_seq, v_init, h_seq, spike_seq,
                       v_out, T, N, tau, v_threshold, v_reset);
  else
    hipLaunchKernelGGL(lif_train_fwd_kernel<1>, dim3(grid_for(N)), dim3(256), 0, stream, x_seq, v_init, h_seq, spike_seq,
                       v_out, T, N, tau, v_threshold, v_reset);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_lif_train_bwd(const float* grad_spike_seq, const float* grad_v_last, const float* h_seq,
                                 float* grad_x_seq, float* grad_v_init, int T, long long N, float tau, float v_threshold,
                                 float v_reset, float alpha, int detach_reset, hipStream_t stream) {
  if (!grad_spike_seq || !h_seq || !grad_x_seq || T <= 0 || N <= 0 || !(tau > 0.f) || !(alpha > 0.f)) return SPK_ERR_ARG;
  const uintptr_t al = (uintptr_t)grad_spike_seq | (uintptr_t)grad_v_last | (uintptr_t)h_seq | (uintptr_t)grad_x_seq |
                       (uintptr_t)grad_v_init;
  const bool vec = (N % 4 == 0) && (al % 16 == 0);
#define SPK_BWD(VEC, DET)                                                                                               \
  hipLaunchKernelGGL((lif_train_bwd_kernel<VEC, DET>), dim3(grid_for((N + VEC - 1) / VEC)), dim3(256), 0, stream,         \
                     grad_spike_seq, grad_v_last, h_seq, grad_x_seq, grad_v_init, T, N, tau, v_threshold, v_reset, alpha)
  if (vec) { if (detach_reset) SPK_BWD(4, true); else SPK_BWD(4, false); }
  else     { if (detach_reset) SPK_BWD(1, true); else SPK_BWD(1, false); }
#undef SPK_BWD
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// ---- PSP: the post-synaptic-potential filter of the VQ-VAE training losses (R/snn_model/snn_layers.py:6-26) ---------
//   forward   syn_t = syn_{t-1} + (x_t - syn_{t-1}) / tau_s,  syn_{-1} = 0            (one output per step)
//   backward  (adjoint, t = T-1..0)   G <- G + dL/dsyn_t;  dL/dx_t = G / tau_s;  G <- G (1 - 1/tau_s)
// Element-wise over the N neurons of a dense [T][N] tensor (any memory order inside N), T steps in a register.
namespace {

template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void psp_kernel(const float* __restrict__ in, float* __restrict__ out, int T, long long N,
                                                  float tau) {
  const long long ngroups = (N + VEC - 1) / VEC;
  const float inv_tau = 1.0f / tau, carry = 1.0f - inv_tau;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (long long)gridDim.x * blockDim.x) {
    const long long n0 = g * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    for (int i = 0; i < T; ++i) {
      const int t = BWD ? T - 1 - i : i;
      float xv[VEC], ov[VEC];
      const float* p = in + (long long)t * N + n0;
      if constexpr (VEC == 4) {
        const f32x4 t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        xv[0] = t4.x; xv[1] = t4.y; xv[2] = t4.z; xv[3] = t4.w;
      } else {
        xv[0] = *p;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        if (BWD) {
          const float G = acc[k] + xv[k];
          ov[k] = G * inv_tau;
          acc[k] = G * carry;
        } else {
          acc[k] = acc[k] + (xv[k] - acc[k]) / tau;
          ov[k] = acc[k];
        }
      }
      float* q = out + (long long)t * N + n0;
      if constexpr (VEC == 4) *reinterpret_cast<float4*>(q) = make_float4(ov[0], ov[1], ov[2], ov[3]);
      else *q = ov[0];
    }
  }
}

}  // namespace

extern "C" int spk_psp(const float* in, float* out, int T, long long N, float tau_s, int backward, hipStream_t stream) {
  if (!in || !out || T <= 0 || N <= 0 || !(tau_s > 0.f)) return SPK_ERR_ARG;
  const bool vec = (N % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) % 16) == 0);
  if (vec && backward) hipLaunchKernelGGL((psp_kernel<4, true>), dim3(grid_for(N / 4)), dim3(256), 0, stream, in, out, T, N, tau_s);
  else if (vec) hipLaunchKernelGGL((psp_kernel<4, false>), dim3(grid_for(N / 4)), dim3(256), 0, stream, in, out, T, N, tau_s);
  else if (backward) hipLaunchKernelGGL((psp_kernel<1, true>), dim3(grid_for(N)), dim3(256), 0, stream, in, out, T, N, tau_s);
  else hipLaunchKernelGGL((psp_kernel<1, false>), dim3(grid_for(N)), dim3(256), 0, stream, in, out, T, N, tau_s);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
