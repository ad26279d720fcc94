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

// Launch-shape choices that earlier rounds measured against each other.  In the shipped library they are COMPILE-TIME CONSTANTS
// (spk_opt folds to the default: no process-wide state, include/spkdiff.h's contract); only a -DSPK_V2_VARIANTS=1 build
// (`make variants`, include/spkdiff_variants.h) keeps them settable through spk_set_option so that the A/B measurements can be repeated.
#ifndef SPK_V2_VARIANTS
#define SPK_V2_VARIANTS 0
#endif
enum { SPK_OPT_CONV6_SHARED = 0, SPK_OPT_CONV6_SHARED_DYN, SPK_OPT_MFMA_DEBUG, SPK_OPT_FP6_XCD_WALK, SPK_OPT_FP6_WAVES,
       SPK_OPT_V2_WAVES, SPK_OPT_V2_LAG, SPK_OPT_V2_DUO, SPK_OPT_V2_DEFER, SPK_OPT_V2_LPS, SPK_OPT_COUNT };
#define SPK_OPT_DEFAULT_VALUES {1, 1, 0, 1, 4, 8, 0, 0, 0, 1}
#if SPK_V2_VARIANTS
int spk_opt(int id);
#else
constexpr int SPK_OPT_DEFAULTS[SPK_OPT_COUNT] = SPK_OPT_DEFAULT_VALUES;
constexpr int spk_opt(int id) { return SPK_OPT_DEFAULTS[id]; }
#endif

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
// operation can observe (x - (-0) == x - (+0), and torch.equal(-0., +0.) is True), and that an INFINITE h resets to 0 here
// and to NaN there ((1 - s) * h): pre-activations are finite.
__device__ __forceinline__ bool spk_lif_step_default(float& v, float x) {
  const float h = v + (x - v) * 0.5f;
  const bool s = h >= 1.0f;
  v = s ? 0.0f : h;
  return s;
}

// The default neuron under a CONSTANT input x from v = 0 (the layers whose input is the same frame at every step: encoder
// conv1, the spike generator, the denoiser's conv1).  After a spike the state is v = 0 again, so the train is periodic with
// the step p(x) of the first spike, and p is a step function of x: p(x) <= k  <=>  x >= theta_k.  The sixteen thresholds
// below are those of the fp32 recurrence above (h = v + (x - v) * 0.5f, three roundings), found by running it on EVERY float
// in [0.5, 4) (tests/test_cabi_and_host.py repeats that, 25 M values; x <= 1 and NaN never fire, x >= 2 fires every step):
// theta_k is the smallest float whose first spike comes at step k or earlier (~ 1 / (1 - 2^-k)).  Sixteen LIF steps become
// one table look-up: with t = x - 1 (exact in [1, 2]) and k = -ilogb(t), theta_{k+1} - 1 <= 2^-k <= t, so p is k or k + 1.
#define SPK_LIF_CONST_TH_BITS {0x40000000u, 0x3faaaaabu, 0x3f924925u, 0x3f888889u, 0x3f842108u, 0x3f820821u, 0x3f810204u, \
                               0x3f808081u, 0x3f804020u, 0x3f802008u, 0x3f801002u, 0x3f800801u, 0x3f800400u, 0x3f800200u, \
                               0x3f800100u, 0x3f800080u}
// spike bits (bit t = step t) of a train with period p = 1 .. 16; p = 17: no spike within sixteen steps; entry 0 unused
#define SPK_LIF_CONST_PATTERNS {0u, 0xffffu, 0xaaaau, 0x4924u, 0x8888u, 0x4210u, 0x0820u, 0x2040u, 0x8080u, 0x0100u, 0x0200u, \
                                0x0400u, 0x0800u, 0x1000u, 0x2000u, 0x4000u, 0x8000u, 0u}
// s_th[k - 1] = theta_k (16 floats), s_pat[p] (18 words): LDS copies of the two tables
__device__ __forceinline__ unsigned spk_lif_const_input_bits16(float x, const float* s_th, const unsigned* s_pat) {
  const float t = x - 1.0f;
  int k = 1 - __builtin_amdgcn_frexp_expf(t);              // t = m * 2^e, m in [0.5, 1): ilogb(t) = e - 1
  k = k < 1 ? 1 : (k > 16 ? 16 : k);
  int p = x >= s_th[k - 1] ? k : k + 1;
  p = x >= 2.0f ? 1 : p;
  p = x > 1.0f ? p : 17;                                    // (also NaN)
  return s_pat[p];
}
