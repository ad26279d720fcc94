// Shared helpers for the libspkdiff HIP kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPK_OK 0
#define SPK_ERR_ARG (-1)      // bad argument (null pointer, non-positive size, unsupported combination)
#define SPK_ERR_UNSUPPORTED (-2)

#define SPK_MAX_T 16          // time steps kept in registers by the fused kernels

// spike storage dtypes (spk_lif_fwd)
#define SPK_SPIKE_F32 0
#define SPK_SPIKE_U8 1
#define SPK_SPIKE_BITS 2      // one bit per neuron-step, 64 neurons per u64 word (wave ballot)

#define SPK_LAUNCH_CHECK()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return (int)e__;                  \
  } while (0)

static inline int spk_blocks(long long n, int threads) { return (int)((n + threads - 1) / threads); }

// One LIF step, exactly the arithmetic of
// SJ/activation_based/neuron.py:799-811 with v_reset as a parameter:
//   v = v + (x - (v - v_reset)) / tau ; s = v >= v_th ; v = v_reset * s + (1 - s) * v
// DIV=false multiplies by inv_tau instead of dividing: bit-identical when tau is a power of two
// (the default tau = 2); DIV=true performs the correctly rounded fp32 division for any other tau.
template <bool DIV>
__device__ __forceinline__ bool spk_lif_step(float& v, float x, float tau, float inv_tau, float v_th, float v_reset) {
  float d = x - (v - v_reset);
  float h = v + (DIV ? d / tau : d * inv_tau);
  bool s = h >= v_th;
  // v_reset*1 + 0*h  |  v_reset*0 + 1*h : the reference's arithmetic maps -0.0 to +0.0 via the add
  v = s ? (v_reset + 0.0f * h) : (v_reset * 0.0f + h);
  return s;
}

// Default neuron of the models (tau 2, v_th 1, v_reset 0: R/snn_model/vae_model.py:37,112,...), lean form for the
// fused epilogues:  h = v + (x - v)/2 ; s = h >= 1 ; v = s ? 0 : h.  Same spikes and the same v as the reference's
// arithmetic except that a zero membrane potential may keep its sign (-0.0 instead of +0.0), which no later
// operation can observe (x - (-0) == x - (+0), and torch.equal(-0., +0.) is True).
__device__ __forceinline__ bool spk_lif_step_default(float& v, float x) {
  const float h = v + (x - v) * 0.5f;
  const bool s = h >= 1.0f;
  v = s ? 0.0f : h;
  return s;
}
