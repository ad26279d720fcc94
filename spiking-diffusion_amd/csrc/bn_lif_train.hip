// Training-mode BatchNorm2d + surrogate-gradient LIF as ONE native operator (the diffusion training
// step).  The reference runs, per denoiser block in train() mode, layer.BatchNorm2d 'm' mode with batch statistics
// (SJ/activation_based/layer.py:458-465 -> F.batch_norm(training=True)) followed by the LIFNode training forward
// (SJ/activation_based/neuron.py:739-749 charge, :133-135 hard reset, surrogate.atan SJ/activation_based/surrogate.py:664-678);
// its native contract for the neuron half is the CuPy pair LIFNodeFPTTKernel / LIFNodeBPTTKernel
// (SJ/activation_based/auto_cuda/neuron_kernel.py:102-225,479-540).  Here both halves are fused:
//
//   forward   stats:    per channel sum / sum of squares over (T, B, HW) in fp64, deterministic two-stage reduction
//             finalize: mean, biased var, invstd = 1/sqrt(var + eps); running statistics (momentum, unbiased var)
//             apply:    z_t = fma(y_t, a, b), a = gamma invstd, b = beta - mean a;  LIF scan over T in registers -> spikes
//   backward  B1:       recompute z_t and the membrane potentials h_t from y (nothing but y is kept from the forward),
//                       BPTT with the ATan surrogate -> g_t = dL/dz_t (stored in grad_y), per-channel sum g, sum g zhat
//             finalize: grad_beta = sum g, grad_gamma = sum g zhat
//             B2:       grad_y = a (g - grad_beta / M - zhat grad_gamma / M),  zhat = (y - mean) invstd,  M = T B HW
//
// HBM bytes per neuron-step: forward 4 (stats) + 4 (apply read) + 4 (spikes) = 12; backward 8 + 4 (B1) + 8 + 4 (B2) = 24;
// the unfused module sequence moves ~3x that (BN output, h_seq, their gradients).
//
// Layout: channels-last, [T][B][HW][C] fp32 for y, spikes and gradients, [B][HW][C] for the membrane state -- the
// layout the library's NHWC convolution kernels read and write without transposes, and the one in which a per-channel
// reduction is coalesced.  Two kernel families with the same partial-sum workspace [slice][C][2]:
//   vector forms (*_v_kernel<VEC>, C % VEC == 0 and 256 % (C / VEC) == 0 -- every layer of the models): a thread owns
//     VEC = 2 or 4 consecutive channels of a row and keeps all T steps in flight (8-16 B loads, 256 consecutive words per
//     workgroup step);
//   scalar forms (any C): a lane owns one channel, a wave reads 64 consecutive channels of one (b, hw) row.
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

constexpr int TPB = 256;          // 4 waves: 64 channels x 4 row phases
constexpr int NW = TPB / 64;

struct Geo {
  int T, R, C;                    // time steps, rows (B * HW) per step, channels
  int S;                          // row slices (gridDim.y); a wave walks rows  (blockIdx.y * NW + wave) + k * NW * S
  long long gs_ts, gs_pitch;      // layout of grad_spike_seq in floats: step stride (0 = the same gradient at every step) and row pitch
};

// per-(slice, channel) partial of two doubles: waves of a block are reduced in LDS in a fixed order
__device__ __forceinline__ void store_partials(double a, double b, double* __restrict__ ws, const Geo& g, int c) {
  __shared__ double sh[NW][64][2];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  sh[w][l][0] = a; sh[w][l][1] = b;
  __syncthreads();
  if (w == 0 && c < g.C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { s0 += sh[i][l][0]; s1 += sh[i][l][1]; }
    double* o = ws + ((long long)blockIdx.y * g.C + c) * 2;
    o[0] = s0; o[1] = s1;
  }
}

__global__ __launch_bounds__(TPB) void bn_stats_kernel(const float* __restrict__ y, double* __restrict__ ws, Geo g) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long ts = (long long)g.R * g.C;
  double s = 0.0, q = 0.0;
  if (c < g.C) {
    for (int r = blockIdx.y * NW + (threadIdx.x >> 6); r < g.R; r += NW * g.S) {
      const float* p = y + (long long)r * g.C + c;
      float yv[SPK_MAX_T];
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t) yv[t] = t < g.T ? p[t * ts] : 0.0f;
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t) {
        const double v = (double)yv[t];
        s += v; q += v * v;
      }
    }
  }
  store_partials(s, q, ws, g, c);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// one wave per channel: lanes stride over the S slice partials (fixed order -> deterministic), shuffle tree, lane 0 writes
__device__ __forceinline__ void reduce_slices(const double* __restrict__ ws, int S, int C, int c, double& a, double& b) {
  a = 0.0; b = 0.0;
  for (int i = threadIdx.x & 63; i < S; i += 64) { a += ws[((long long)i * C + c) * 2]; b += ws[((long long)i * C + c) * 2 + 1]; }
  a = wave_sum(a); b = wave_sum(b);
}

__global__ __launch_bounds__(TPB) void bn_finalize_kernel(const double* __restrict__ ws, int S, int C, double M, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd) {
  const int c = blockIdx.x * NW + (threadIdx.x >> 6);
  if (c >= C) return;
  double s, q;
  reduce_slices(ws, S, C, c, s, q);
  if ((threadIdx.x & 63) != 0) return;
  const double mean = s / M;
  double var = q / M - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (float)((double)momentum * mean + (1.0 - (double)momentum) * (double)running_mean[c]);
  if (running_var) {
    const double unbiased = M > 1.0 ? var * M / (M - 1.0) : var;
    running_var[c] = (float)((double)momentum * unbiased + (1.0 - (double)momentum) * (double)running_var[c]);
  }
}

__global__ __launch_bounds__(TPB) void bn_lif_apply_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, const float* __restrict__ v_init,
                                                           float* __restrict__ spikes, float* __restrict__ v_out, Geo g,
                                                           float tau, float v_th, float v_reset) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= g.C) return;
  const long long ts = (long long)g.R * g.C;
  const float a = (gamma ? gamma[c] : 1.0f) * save_invstd[c];
  const float b = (beta ? beta[c] : 0.0f) - save_mean[c] * a;
  for (int r = blockIdx.y * NW + (threadIdx.x >> 6); r < g.R; r += NW * g.S) {
    const long long n = (long long)r * g.C + c;
    float v = v_init ? v_init[n] : v_reset;
    float yv[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) yv[t] = t < g.T ? y[n + t * ts] : 0.0f;
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
        const float z = fmaf(yv[t], a, b);
        const float h = v + (z - (v - v_reset)) / tau;
        const float s = (h - v_th >= 0.0f) ? 1.0f : 0.0f;
        v = (1.0f - s) * h + s * v_reset;
        spikes[n + t * ts] = s;
      }
    }
    if (v_out) v_out[n] = v;
  }
}

template <bool DETACH>
__global__ __launch_bounds__(TPB) void bn_lif_bwd1_kernel(const float* __restrict__ grad_s, const float* __restrict__ grad_v_last,
                                                          const float* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, const float* __restrict__ v_init,
                                                          float* __restrict__ grad_y, float* __restrict__ grad_v_init,
                                                          double* __restrict__ ws, Geo g, float tau, float v_th, float v_reset,
                                                          float alpha) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const long long ts = (long long)g.R * g.C;
  double s1 = 0.0, s2 = 0.0;
  if (c < g.C) {
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float a = (gamma ? gamma[c] : 1.0f) * invstd;
    const float b = (beta ? beta[c] : 0.0f) - mean * a;
    const float inv_tau = 1.0f / tau, carry = 1.0f - inv_tau;
    for (int r = blockIdx.y * NW + (threadIdx.x >> 6); r < g.R; r += NW * g.S) {
      const long long n = (long long)r * g.C + c;
      float yv[SPK_MAX_T], hv[SPK_MAX_T], gsv[SPK_MAX_T];
      float v = v_init ? v_init[n] : v_reset;
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t) {
        if (t < g.T) {
          yv[t] = y[n + t * ts];
          gsv[t] = grad_s[(long long)r * g.gs_pitch + c + t * g.gs_ts];
        }
      }
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t) {
        if (t < g.T) {
          const float z = fmaf(yv[t], a, b);
          const float h = v + (z - (v - v_reset)) / tau;
          const float s = (h - v_th >= 0.0f) ? 1.0f : 0.0f;
          v = (1.0f - s) * h + s * v_reset;
          hv[t] = h;
        }
      }
      float G = grad_v_last ? grad_v_last[n] : 0.0f;
#pragma unroll
      for (int t = SPK_MAX_T - 1; t >= 0; --t) {
        if (t < g.T) {
          const float over = hv[t] - v_th;
          const float s = over >= 0.0f ? 1.0f : 0.0f;
          const float ax = 1.57079632679489661923f * alpha * over;
          const float g_s = alpha / 2.0f / (1.0f + ax * ax);
          float dv_dh = 1.0f - s;
          if (!DETACH) dv_dh = (v_reset - hv[t]) * g_s + dv_dh;
          const float gh = G * dv_dh + gsv[t] * g_s;
          const float gz = gh * inv_tau;
          G = gh * carry;
          grad_y[n + t * ts] = gz;
          s1 += (double)gz;
          s2 += (double)gz * (double)((yv[t] - mean) * invstd);
        }
      }
      if (grad_v_init) grad_v_init[n] = G;
    }
  }
  store_partials(s1, s2, ws, g, c);
}

__global__ __launch_bounds__(TPB) void bn_bwd_finalize_kernel(const double* __restrict__ ws, int S, int C,
                                                              float* __restrict__ grad_gamma, float* __restrict__ grad_beta) {
  const int c = blockIdx.x * NW + (threadIdx.x >> 6);
  if (c >= C) return;
  double s1, s2;
  reduce_slices(ws, S, C, c, s1, s2);
  if ((threadIdx.x & 63) != 0) return;
  grad_beta[c] = (float)s1;
  grad_gamma[c] = (float)s2;
}

__global__ __launch_bounds__(TPB) void bn_bwd2_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                      const float* __restrict__ grad_gamma, const float* __restrict__ grad_beta,
                                                      float* __restrict__ grad_y, Geo g, float inv_M) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  if (c >= g.C) return;
  const long long ts = (long long)g.R * g.C;
  const float mean = save_mean[c], invstd = save_invstd[c];
  const float a = (gamma ? gamma[c] : 1.0f) * invstd;
  const float c1 = grad_beta[c] * inv_M, c2 = grad_gamma[c] * inv_M;
  for (int r = blockIdx.y * NW + (threadIdx.x >> 6); r < g.R; r += NW * g.S) {
    const long long n = (long long)r * g.C + c;
    float yv[SPK_MAX_T], gv[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      yv[t] = t < g.T ? y[n + t * ts] : 0.0f;
      gv[t] = t < g.T ? grad_y[n + t * ts] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
        const float zh = (yv[t] - mean) * invstd;
        grad_y[n + t * ts] = a * (gv[t] - c1 - zh * c2);
      }
    }
  }
}

// ---- vector forms: a thread owns VEC consecutive channels (q = tid % Q, Q = C / VEC, 256 % Q == 0) and walks rows
// j = tid / Q, j + 256 / Q, ...: consecutive threads read consecutive 4*VEC-byte words, a workgroup reads 256 * VEC
// contiguous floats per step and keeps T of them in flight per thread.  Partials land in the same ws layout [slice][C][2].
template <int VEC> struct Vec;
template <> struct Vec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef float type __attribute__((ext_vector_type(4))); };

struct GeoV {
  int T, R, C, S, Q, RPB;         // Q = C / VEC threads per row, RPB = 256 / Q rows per workgroup step
  long long gs_ts, gs_pitch;      // layout of grad_spike_seq in floats (see Geo)
};

template <int VEC>
__device__ __forceinline__ void store_partials_v(const double (&a)[VEC], const double (&b)[VEC], double* __restrict__ ws,
                                                 const GeoV& g) {
  __shared__ double sh[TPB][2 * VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { sh[threadIdx.x][2 * i] = a[i]; sh[threadIdx.x][2 * i + 1] = b[i]; }
  __syncthreads();
  if ((int)threadIdx.x < g.Q) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double s0 = 0.0, s1 = 0.0;
      for (int j = 0; j < g.RPB; ++j) { s0 += sh[threadIdx.x + j * g.Q][2 * i]; s1 += sh[threadIdx.x + j * g.Q][2 * i + 1]; }
      double* o = ws + ((long long)blockIdx.x * g.C + threadIdx.x * VEC + i) * 2;
      o[0] = s0; o[1] = s1;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(TPB) void bn_stats_v_kernel(const float* __restrict__ y, double* __restrict__ ws, GeoV g) {
  typedef typename Vec<VEC>::type vf;
  const int q = threadIdx.x % g.Q, j = threadIdx.x / g.Q;
  const long long ts = (long long)g.R * g.Q;
  const vf* yv = reinterpret_cast<const vf*>(y);
  double s[VEC], sq[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { s[i] = 0.0; sq[i] = 0.0; }
  for (int r = blockIdx.x * g.RPB + j; r < g.R; r += g.RPB * g.S) {
    const long long n = (long long)r * g.Q + q;
    vf v[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) if (t < g.T) v[t] = yv[n + t * ts];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const double d = (double)v[t][i]; s[i] += d; sq[i] += d * d; }
      }
    }
  }
  store_partials_v<VEC>(s, sq, ws, g);
}

template <int VEC>
__global__ __launch_bounds__(TPB) void bn_lif_apply_v_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                             const float* __restrict__ save_invstd, const float* __restrict__ v_init,
                                                             float* __restrict__ spikes, float* __restrict__ v_out, GeoV g,
                                                             float tau, float v_th, float v_reset, uint8_t* __restrict__ c4,
                                                             int HW) {
  typedef typename Vec<VEC>::type vf;
  const int q = threadIdx.x % g.Q, j = threadIdx.x / g.Q;
  const long long ts = (long long)g.R * g.Q;
  const vf* yv = reinterpret_cast<const vf*>(y);
  vf* sv = reinterpret_cast<vf*>(spikes);
  float a[VEC], b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = q * VEC + i;
    a[i] = (gamma ? gamma[c] : 1.0f) * save_invstd[c];
    b[i] = (beta ? beta[c] : 0.0f) - save_mean[c] * a[i];
  }
  for (int r = blockIdx.x * g.RPB + j; r < g.R; r += g.RPB * g.S) {
    const long long n = (long long)r * g.Q + q;
    vf in[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) if (t < g.T) in[t] = yv[n + t * ts];
    vf v;
    if (v_init) v = reinterpret_cast<const vf*>(v_init)[n];
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = v_reset;
    }
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
        vf o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float z = fmaf(in[t][i], a[i], b[i]);
          const float h = v[i] + (z - (v[i] - v_reset)) / tau;
          const float sp = (h - v_th >= 0.0f) ? 1.0f : 0.0f;
          v[i] = (1.0f - sp) * h + sp * v_reset;
          o[i] = sp;
        }
        sv[n + t * ts] = o;
        if constexpr (VEC == 4) {
          // the same spikes as "C4" records [B][C / 64][HW][T][32 bytes = 64 channels x e2m1] -- what the next layer's exact
          // MFMA forward reads (spk_spikes_nhwc_to_fp4 made them from the fp32 tensor: a launch per layer and iteration)
          if (c4) {
            const int bb = r / HW, hw = r - bb * HW, c = q * 4;
            const unsigned w = (o[0] != 0.f ? 0x2u : 0u) | (o[1] != 0.f ? 0x20u : 0u) | (o[2] != 0.f ? 0x200u : 0u) |
                               (o[3] != 0.f ? 0x2000u : 0u);
            *reinterpret_cast<uint16_t*>(c4 + ((((long long)bb * (g.C >> 6) + (c >> 6)) * HW + hw) * g.T + t) * 32 +
                                         ((c & 63) >> 1)) = (uint16_t)w;
          }
        }
      }
    }
    if (v_out) reinterpret_cast<vf*>(v_out)[n] = v;
  }
}

template <int VEC, bool DETACH>
__global__ __launch_bounds__(TPB) void bn_lif_bwd1_v_kernel(const float* __restrict__ grad_s, const float* __restrict__ grad_v_last,
                                                            const float* __restrict__ y, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd, const float* __restrict__ v_init,
                                                            float* __restrict__ grad_y, float* __restrict__ grad_v_init,
                                                            double* __restrict__ ws, GeoV g, float tau, float v_th, float v_reset,
                                                            float alpha) {
  typedef typename Vec<VEC>::type vf;
  const int q = threadIdx.x % g.Q, j = threadIdx.x / g.Q;
  const long long ts = (long long)g.R * g.Q;
  const vf* yv = reinterpret_cast<const vf*>(y);
  const vf* gsv = reinterpret_cast<const vf*>(grad_s);
  vf* gyv = reinterpret_cast<vf*>(grad_y);
  const float inv_tau = 1.0f / tau, carry = 1.0f - inv_tau;
  float mean[VEC], invstd[VEC], a[VEC], b[VEC];
  double s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = q * VEC + i;
    mean[i] = save_mean[c]; invstd[i] = save_invstd[c];
    a[i] = (gamma ? gamma[c] : 1.0f) * invstd[i];
    b[i] = (beta ? beta[c] : 0.0f) - mean[i] * a[i];
    s1[i] = 0.0; s2[i] = 0.0;
  }
  for (int r = blockIdx.x * g.RPB + j; r < g.R; r += g.RPB * g.S) {
    const long long n = (long long)r * g.Q + q;
    vf yy[SPK_MAX_T], hh[SPK_MAX_T], gs[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) { yy[t] = yv[n + t * ts]; gs[t] = gsv[((long long)r * g.gs_pitch + t * g.gs_ts) / VEC + q]; }
    }
    vf v;
    if (v_init) v = reinterpret_cast<const vf*>(v_init)[n];
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = v_reset;
    }
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float z = fmaf(yy[t][i], a[i], b[i]);
          const float h = v[i] + (z - (v[i] - v_reset)) / tau;
          const float sp = (h - v_th >= 0.0f) ? 1.0f : 0.0f;
          v[i] = (1.0f - sp) * h + sp * v_reset;
          hh[t][i] = h;
        }
      }
    }
    vf G;
    if (grad_v_last) G = reinterpret_cast<const vf*>(grad_v_last)[n];
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) G[i] = 0.0f;
    }
#pragma unroll
    for (int t = SPK_MAX_T - 1; t >= 0; --t) {
      if (t < g.T) {
        vf o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float over = hh[t][i] - v_th;
          const float sp = over >= 0.0f ? 1.0f : 0.0f;
          const float ax = 1.57079632679489661923f * alpha * over;
          const float g_s = alpha / 2.0f / (1.0f + ax * ax);
          float dv_dh = 1.0f - sp;
          if (!DETACH) dv_dh = (v_reset - hh[t][i]) * g_s + dv_dh;
          const float gh = G[i] * dv_dh + gs[t][i] * g_s;
          const float gz = gh * inv_tau;
          G[i] = gh * carry;
          o[i] = gz;
          s1[i] += (double)gz;
          s2[i] += (double)gz * (double)((yy[t][i] - mean[i]) * invstd[i]);
        }
        gyv[n + t * ts] = o;
      }
    }
    if (grad_v_init) reinterpret_cast<vf*>(grad_v_init)[n] = G;
  }
  store_partials_v<VEC>(s1, s2, ws, g);
}

template <int VEC>
__global__ __launch_bounds__(TPB) void bn_bwd2_v_kernel(const float* __restrict__ y, const float* __restrict__ gamma,
                                                        const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                        const float* __restrict__ grad_gamma, const float* __restrict__ grad_beta,
                                                        float* __restrict__ grad_y, GeoV g, float inv_M) {
  typedef typename Vec<VEC>::type vf;
  const int q = threadIdx.x % g.Q, j = threadIdx.x / g.Q;
  const long long ts = (long long)g.R * g.Q;
  const vf* yv = reinterpret_cast<const vf*>(y);
  vf* gyv = reinterpret_cast<vf*>(grad_y);
  float mean[VEC], invstd[VEC], a[VEC], c1[VEC], c2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int c = q * VEC + i;
    mean[i] = save_mean[c]; invstd[i] = save_invstd[c];
    a[i] = (gamma ? gamma[c] : 1.0f) * invstd[i];
    c1[i] = grad_beta[c] * inv_M; c2[i] = grad_gamma[c] * inv_M;
  }
  for (int r = blockIdx.x * g.RPB + j; r < g.R; r += g.RPB * g.S) {
    const long long n = (long long)r * g.Q + q;
    vf yy[SPK_MAX_T], gg[SPK_MAX_T];
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) { yy[t] = yv[n + t * ts]; gg[t] = gyv[n + t * ts]; }
    }
#pragma unroll
    for (int t = 0; t < SPK_MAX_T; ++t) {
      if (t < g.T) {
        vf o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = a[i] * (gg[t][i] - c1[i] - ((yy[t][i] - mean[i]) * invstd[i]) * c2[i]);
        gyv[n + t * ts] = o;
      }
    }
  }
}

// vector width usable for C channels: Q = C / VEC threads per row must divide the workgroup
inline int vec_for(int C, int want) {
  for (int v = want; v >= 2; v >>= 1)
    if (C % v == 0 && C / v <= TPB && TPB % (C / v) == 0) return v;
  return 1;
}

inline GeoV geo_v(int T, int R, int C, int vec, int S, long long gs_ts = -1, long long gs_pitch = -1) {
  const int Q = C / vec, RPB = TPB / Q;
  return GeoV{T, R, C, S, Q, RPB, gs_ts < 0 ? (long long)R * C : gs_ts, gs_pitch < 0 ? (long long)C : gs_pitch};
}

// row slices of the scalar form: enough waves to fill 256 CUs several times over, never more than one row per wave
inline int slices(int R, int C) {
  const int groups = (C + 63) / 64;
  int s = 8192 / (groups * NW);
  const int cap = (R + NW - 1) / NW;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

// workgroups of the vector forms: 4 per CU (8-16 KB in flight per wave), never more than one row step per workgroup
inline int slices_v(int R, int C, int vec) {
  const int rpb = TPB / (C / vec);
  int s = 1024;
  const int cap = (R + rpb - 1) / rpb;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

inline int max_slices(int R, int C) {
  int m = slices(R, C);
  for (int v = 2; v <= 4; v <<= 1) {
    if (vec_for(C, v) == v) { const int s = slices_v(R, C, v); if (s > m) m = s; }
  }
  return m;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" long long spk_bn_lif_train_ws_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0 || (long long)B * HW > (1LL << 30)) return SPK_ERR_ARG;
  return (long long)C * max_slices(B * HW, C) * 2 * (long long)sizeof(double);
}

static int bn_lif_train_fwd_impl(const float* y, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, const float* v_init, float* spike_seq,
                                 float* v_out, float* save_mean, float* save_invstd, void* ws, long long ws_bytes, int T,
                                 int B, int C, int HW, float tau, float v_threshold, float v_reset, uint8_t* c4,
                                 hipStream_t stream) {
  if (!y || !spike_seq || !save_mean || !save_invstd || !ws || T <= 0 || T > SPK_MAX_T || B <= 0 || C <= 0 || HW <= 0 ||
      !(tau > 0.f) || (long long)B * HW > (1LL << 30))
    return SPK_ERR_ARG;
  if (ws_bytes < spk_bn_lif_train_ws_bytes(B, C, HW)) return SPK_ERR_ARG;
  const int R = B * HW;
  const double M = (double)T * R;
  const bool al = aligned16(y) && aligned16(spike_seq) && (!v_init || aligned16(v_init)) && (!v_out || aligned16(v_out));
  const int vec = al ? vec_for(C, 4) : 1;
  if (c4 && (vec != 4 || (C % 64))) return SPK_ERR_UNSUPPORTED;          // (C4 records hold 64 channels; four per thread)
  if (vec >= 2) {
    const int S = slices_v(R, C, vec);
    const GeoV g = geo_v(T, R, C, vec, S);
    if (vec == 4) hipLaunchKernelGGL(bn_stats_v_kernel<4>, dim3(S), dim3(TPB), 0, stream, y, (double*)ws, g);
    else hipLaunchKernelGGL(bn_stats_v_kernel<2>, dim3(S), dim3(TPB), 0, stream, y, (double*)ws, g);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + NW - 1) / NW), dim3(TPB), 0, stream, (const double*)ws, S, C, M, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd);
    if (vec == 4)
      hipLaunchKernelGGL(bn_lif_apply_v_kernel<4>, dim3(S), dim3(TPB), 0, stream, y, gamma, beta, save_mean, save_invstd,
                         v_init, spike_seq, v_out, g, tau, v_threshold, v_reset, c4, HW);
    else
      hipLaunchKernelGGL(bn_lif_apply_v_kernel<2>, dim3(S), dim3(TPB), 0, stream, y, gamma, beta, save_mean, save_invstd,
                         v_init, spike_seq, v_out, g, tau, v_threshold, v_reset, (uint8_t*)nullptr, HW);
  } else {
    const int S = slices(R, C);
    const Geo g{T, R, C, S, (long long)R * C, (long long)C};
    const dim3 grid((C + 63) / 64, S);
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(TPB), 0, stream, y, (double*)ws, g);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + NW - 1) / NW), dim3(TPB), 0, stream, (const double*)ws, S, C, M, eps,
                       momentum, running_mean, running_var, save_mean, save_invstd);
    hipLaunchKernelGGL(bn_lif_apply_kernel, grid, dim3(TPB), 0, stream, y, gamma, beta, save_mean, save_invstd, v_init,
                       spike_seq, v_out, g, tau, v_threshold, v_reset);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_bn_lif_train_fwd(const float* y, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, const float* v_init, float* spike_seq,
                                    float* v_out, float* save_mean, float* save_invstd, void* ws, long long ws_bytes, int T,
                                    int B, int C, int HW, float tau, float v_threshold, float v_reset, hipStream_t stream) {
  return bn_lif_train_fwd_impl(y, gamma, beta, running_mean, running_var, momentum, eps, v_init, spike_seq, v_out, save_mean,
                               save_invstd, ws, ws_bytes, T, B, C, HW, tau, v_threshold, v_reset, nullptr, stream);
}

// The same forward, also leaving the spikes as C4 records [B][C / 64][HW][T][32] (spikes_c4_out, or null) for the next layer's
// exact MFMA forward: the apply launch writes them next to the fp32 spikes (a conversion launch per layer and iteration
// before).  spikes_c4_out needs C % 64 == 0 and 16-byte aligned tensors (SPK_ERR_UNSUPPORTED otherwise: nothing launched).
// (Measured and dropped in round 4: finishing the per-channel reduction in the last workgroup of the statistics launch instead of
//  a finalize launch -- one workgroup reading 784 x 512 partial pairs takes 200-350 us where the 128-workgroup finalize launch
//  takes 6.)
extern "C" int spk_bn_lif_train_fwd_c4(const float* y, const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, float momentum, float eps, const float* v_init, float* spike_seq,
                                       float* v_out, float* save_mean, float* save_invstd, uint8_t* spikes_c4_out, void* ws,
                                       long long ws_bytes, int T, int B, int C, int HW, float tau, float v_threshold,
                                       float v_reset, hipStream_t stream) {
  return bn_lif_train_fwd_impl(y, gamma, beta, running_mean, running_var, momentum, eps, v_init, spike_seq, v_out, save_mean,
                               save_invstd, ws, ws_bytes, T, B, C, HW, tau, v_threshold, v_reset, spikes_c4_out, stream);
}

static int bn_lif_train_bwd_impl(const float* grad_spike_seq, long long gs_ts, long long gs_pitch, const float* grad_v_last,
                                 const float* y, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, const float* v_init,
                                    float* grad_y, float* grad_gamma, float* grad_beta, float* grad_v_init, void* ws,
                                    long long ws_bytes, int T, int B, int C, int HW, float tau, float v_threshold,
                                    float v_reset, float alpha, int detach_reset, hipStream_t stream) {
  if (!grad_spike_seq || !y || !save_mean || !save_invstd || !grad_y || !grad_gamma || !grad_beta || !ws || T <= 0 ||
      T > SPK_MAX_T || B <= 0 || C <= 0 || HW <= 0 || !(tau > 0.f) || !(alpha > 0.f) || (long long)B * HW > (1LL << 30))
    return SPK_ERR_ARG;
  if (ws_bytes < spk_bn_lif_train_ws_bytes(B, C, HW)) return SPK_ERR_ARG;
  const int R = B * HW;
  const float inv_M = (float)(1.0 / ((double)T * R));
  if (gs_ts < 0 || gs_pitch < C) return SPK_ERR_ARG;
  const bool al = aligned16(y) && aligned16(grad_spike_seq) && aligned16(grad_y) && (!v_init || aligned16(v_init)) &&
                  (!grad_v_last || aligned16(grad_v_last)) && (!grad_v_init || aligned16(grad_v_init)) && (gs_ts % 4) == 0 &&
                  (gs_pitch % 4) == 0;
  // the BPTT pass keeps y, h and grad_s of T steps per channel in registers: 2 channels per thread; the rest 4
  const int vec1 = al ? vec_for(C, 2) : 1, vec2 = al ? vec_for(C, 4) : 1;
#define SPK_BWD1_ARGS grad_spike_seq, grad_v_last, y, gamma, beta, save_mean, save_invstd, v_init, grad_y, grad_v_init, (double*)ws
  int S;
  if (vec1 == 2) {
    S = slices_v(R, C, 2);
    const GeoV g = geo_v(T, R, C, 2, S, gs_ts, gs_pitch);
    if (detach_reset)
      hipLaunchKernelGGL((bn_lif_bwd1_v_kernel<2, true>), dim3(S), dim3(TPB), 0, stream, SPK_BWD1_ARGS, g, tau, v_threshold,
                         v_reset, alpha);
    else
      hipLaunchKernelGGL((bn_lif_bwd1_v_kernel<2, false>), dim3(S), dim3(TPB), 0, stream, SPK_BWD1_ARGS, g, tau, v_threshold,
                         v_reset, alpha);
  } else {
    S = slices(R, C);
    const Geo g{T, R, C, S, gs_ts, gs_pitch};
    const dim3 grid((C + 63) / 64, S);
    if (detach_reset)
      hipLaunchKernelGGL(bn_lif_bwd1_kernel<true>, grid, dim3(TPB), 0, stream, SPK_BWD1_ARGS, g, tau, v_threshold, v_reset, alpha);
    else
      hipLaunchKernelGGL(bn_lif_bwd1_kernel<false>, grid, dim3(TPB), 0, stream, SPK_BWD1_ARGS, g, tau, v_threshold, v_reset, alpha);
  }
#undef SPK_BWD1_ARGS
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + NW - 1) / NW), dim3(TPB), 0, stream, (const double*)ws, S, C, grad_gamma,
                     grad_beta);
  if (vec2 >= 2) {
    const int S2 = slices_v(R, C, vec2);
    const GeoV g = geo_v(T, R, C, vec2, S2);
    if (vec2 == 4)
      hipLaunchKernelGGL(bn_bwd2_v_kernel<4>, dim3(S2), dim3(TPB), 0, stream, y, gamma, save_mean, save_invstd, grad_gamma,
                         grad_beta, grad_y, g, inv_M);
    else
      hipLaunchKernelGGL(bn_bwd2_v_kernel<2>, dim3(S2), dim3(TPB), 0, stream, y, gamma, save_mean, save_invstd, grad_gamma,
                         grad_beta, grad_y, g, inv_M);
  } else {
    const int S2 = slices(R, C);
    const Geo g{T, R, C, S2, (long long)R * C, (long long)C};
    hipLaunchKernelGGL(bn_bwd2_kernel, dim3((C + 63) / 64, S2), dim3(TPB), 0, stream, y, gamma, save_mean, save_invstd,
                       grad_gamma, grad_beta, grad_y, g, inv_M);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_bn_lif_train_bwd(const float* grad_spike_seq, const float* grad_v_last, const float* y, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, const float* v_init,
                                    float* grad_y, float* grad_gamma, float* grad_beta, float* grad_v_init, void* ws,
                                    long long ws_bytes, int T, int B, int C, int HW, float tau, float v_threshold,
                                    float v_reset, float alpha, int detach_reset, hipStream_t stream) {
  return bn_lif_train_bwd_impl(grad_spike_seq, (long long)B * HW * C, (long long)C, grad_v_last, y, gamma, beta, save_mean,
                               save_invstd, v_init, grad_y, grad_gamma, grad_beta, grad_v_init, ws, ws_bytes, T, B, C, HW, tau,
                               v_threshold, v_reset, alpha, detach_reset, stream);
}

// The same backward with the layout of grad_spike_seq spelled out (floats): step stride -- 0 when every step receives the
// SAME gradient, as the spikes in front of the denoiser's last layer do (its time mean hands g / T to every step:
// R/snn_model/vq_diffusion.py:205-206) -- and row pitch >= C (a channel slice of a wider channels-last tensor: the x5 / x1 halves
// of cat(x5, x1), :205).  The gradient is then read where autograd left it instead of being expanded and copied first.
extern "C" int spk_bn_lif_train_bwd_strided(const float* grad_spike_seq, long long grad_step_stride, long long grad_row_pitch,
                                            const float* grad_v_last, const float* y, const float* gamma, const float* beta,
                                            const float* save_mean, const float* save_invstd, const float* v_init, float* grad_y,
                                            float* grad_gamma, float* grad_beta, float* grad_v_init, void* ws, long long ws_bytes,
                                            int T, int B, int C, int HW, float tau, float v_threshold, float v_reset, float alpha,
                                            int detach_reset, hipStream_t stream) {
  return bn_lif_train_bwd_impl(grad_spike_seq, grad_step_stride, grad_row_pitch, grad_v_last, y, gamma, beta, save_mean,
                               save_invstd, v_init, grad_y, grad_gamma, grad_beta, grad_v_init, ws, ws_bytes, T, B, C, HW, tau,
                               v_threshold, v_reset, alpha, detach_reset, stream);
}
