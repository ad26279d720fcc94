// VectorQuantizer kernels.
//   spk_vq_readout_argmin  membrane/rate read-out + L2 argmin + codebook gather
//                          R/snn_model/vae_model.py:40-52 (x_memout), :87-95 (get_code_indices), :97-99 (quantize)
//   spk_embedding_fwd      nn.Embedding lookup ("quantize"), optionally written as [B,D,h,w]
//
// Read-out arithmetic follows the reference in fp32:  x = (1-alpha)*sum_t z[t]*coef[t] + (alpha*sum_t z[t])/T.
// The distance  |x|^2 + |e_k|^2 - 2 x.e_k  is evaluated in fp64 (exact products, order-independent), and the
// argmin takes the FIRST minimal index like torch.argmin.  Code indices are int64 like the reference's.
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

constexpr int VQ_MAX_D = 64;

// block = 256 threads = 4 waves; one wave per latent position; the codebook and its squared norms are staged in LDS once
// per block; a position's read-out vector lives in registers (16 lane broadcasts), so the position loop has no barrier.
// Distances in fp64 in the reference's order of operations: x2 + e2 - 2 * dot, sums over d ascending.
// DC > 0: D == DC at compile time (the read-out vector then sits in DC scalar-broadcast registers and the loops unroll)
template <int DC>
__global__ __launch_bounds__(256) void vq_kernel(const uint8_t* __restrict__ z, const float* __restrict__ x_in,
                                                 const float* __restrict__ coef,
                                                 const float* __restrict__ alpha_p, const float* __restrict__ cb,
                                                 long long* __restrict__ idx_out, float* __restrict__ zq_out,
                                                 float* __restrict__ xm_out, int T, int B, int D_rt, int HW, int K) {
  const int D = DC > 0 ? DC : D_rt;
  extern __shared__ float lds[];
  float* s_cb = lds;                                               // [K][D+1]
  double* s_e2 = reinterpret_cast<double*>(s_cb + ((K * (D + 1) + 1) & ~1));   // [K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) s_cb[(i / D) * (D + 1) + (i % D)] = cb[i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double e2 = 0.0;
    for (int d = 0; d < D; ++d) { const double e = s_cb[k * (D + 1) + d]; e2 += e * e; }
    s_e2[k] = e2;
  }
  __syncthreads();
  const float alpha = alpha_p ? alpha_p[0] : 0.f;
  const float one_m_alpha = 1.0f - alpha;
  const long long npos = (long long)B * HW;
  for (long long p = (long long)blockIdx.x * 4 + wave; p < npos; p += (long long)gridDim.x * 4) {
    float xl = 0.f;                                                // lane d < D holds x[d]
    if (lane < D && x_in) {
      xl = x_in[p * D + lane];
    } else if (lane < D) {
      const uint8_t* zp = z + p * T * D + lane;
      float m = 0.f, cnt = 0.f;
      if (DC > 0 && T == 16) {
        // all sixteen spike bytes and coefficients are requested before the first is used (a rolled loop waits for each
        // pair in turn: sixteen dependent round trips per position); the sums keep the reference's order over t
        float sv[16], cf[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { sv[t] = (float)zp[t * D]; cf[t] = coef[t]; }
#pragma unroll
        for (int t = 0; t < 16; ++t) { m = m + sv[t] * cf[t]; cnt = cnt + sv[t]; }
      } else {
        for (int t = 0; t < T; ++t) {
          float s = (float)zp[t * D];
          m = m + s * coef[t];
          cnt = cnt + s;
        }
      }
      xl = one_m_alpha * m + (alpha * cnt) / (float)T;
      if (xm_out) xm_out[p * D + lane] = xl;
    }
    double best = 1.0e300;
    int besti = 0x7fffffff;
    double x2 = 0.0;
    if constexpr (DC > 0) {
      double xr[DC];
#pragma unroll
      for (int d = 0; d < DC; ++d) { xr[d] = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl), d)); x2 += xr[d] * xr[d]; }
      for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const int kc = k < K ? k : K - 1;
        double dot = 0.0;
#pragma unroll
        for (int d = 0; d < DC; ++d) dot += xr[d] * (double)s_cb[kc * (DC + 1) + d];
        const double dist = x2 + s_e2[kc] - 2.0 * dot;
        if (k < K && dist < best) { best = dist; besti = k; }      // k increasing per lane: first minimum kept
      }
    } else {
      for (int d = 0; d < D; ++d) { const double xv = (double)__shfl(xl, d); x2 += xv * xv; }
      for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const int kc = k < K ? k : K - 1;
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += (double)__shfl(xl, d) * (double)s_cb[kc * (D + 1) + d];
        const double dist = x2 + s_e2[kc] - 2.0 * dot;
        if (k < K && dist < best) { best = dist; besti = k; }      // k increasing per lane: first minimum kept
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      double ob = __shfl_xor(best, off);
      int oi = __shfl_xor(besti, off);
      if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) idx_out[p] = (long long)besti;
    if (zq_out && lane < D) {
      const int b = (int)(p / HW), hw = (int)(p % HW);
      zq_out[((long long)b * D + lane) * HW + hw] = s_cb[besti * (D + 1) + lane];
    }
  }
}

// D == 16, T == 16, spike input, fp64 codebook in LDS: the launch is a chain of dependent round trips per position (sixteen
// byte loads, then the distances), so a wave reads out FOUR positions at once (lane = position within the group * 16 + d) and
// then scans the codebook for each of them from registers.  Same arithmetic and order of operations as vq_kernel.
__global__ __launch_bounds__(256) void vq16_kernel(const uint8_t* __restrict__ z, const float* __restrict__ coef,
                                                   const float* __restrict__ alpha_p, const float* __restrict__ cb,
                                                   long long* __restrict__ idx_out, float* __restrict__ zq_out,
                                                   float* __restrict__ xm_out, int B, int HW, int K) {
  constexpr int D = 16, T = 16;
  extern __shared__ float lds[];
  double* s_cbd = reinterpret_cast<double*>(lds);                  // [K][D+1]
  double* s_e2 = s_cbd + K * (D + 1);                              // [K]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) s_cbd[(i / D) * (D + 1) + (i % D)] = (double)cb[i];
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double e2 = 0.0;
    for (int d = 0; d < D; ++d) { const double e = s_cbd[k * (D + 1) + d]; e2 += e * e; }
    s_e2[k] = e2;
  }
  __syncthreads();
  const float alpha = alpha_p[0], one_m_alpha = 1.0f - alpha;
  float cf[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) cf[t] = coef[t];
  const long long npos = (long long)B * HW;
  const int sub = lane >> 4, dl = lane & 15;
  for (long long p0 = ((long long)blockIdx.x * 4 + wave) * 4; p0 < npos; p0 += (long long)gridDim.x * 16) {
    const long long pm = p0 + sub;
    float xl = 0.f;
    if (pm < npos) {
      const uint8_t* zp = z + pm * T * D + dl;
      float sv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) sv[t] = (float)zp[t * D];
      float m = 0.f, cnt = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) { m = m + sv[t] * cf[t]; cnt = cnt + sv[t]; }
      xl = one_m_alpha * m + (alpha * cnt) / (float)T;
      if (xm_out) xm_out[pm * D + dl] = xl;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long p = p0 + j;
      if (p >= npos) break;                                        // (uniform over the wave)
      double xr[D], x2 = 0.0;
#pragma unroll
      // (products of two fp32 values are exact in fp64: fma(a, b, s) == s + a * b bit for bit, in half the instructions)
      for (int d = 0; d < D; ++d) { xr[d] = (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xl), 16 * j + d)); x2 = fma(xr[d], xr[d], x2); }
      double best = 1.0e300;
      int besti = 0x7fffffff;
      for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        const int kc = k < K ? k : K - 1;
        double dot = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) dot = fma(xr[d], s_cbd[kc * (D + 1) + d], dot);
        const double dist = x2 + s_e2[kc] - 2.0 * dot;
        if (k < K && dist < best) { best = dist; besti = k; }      // k increasing per lane: first minimum kept
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
      }
      if (lane == 0) idx_out[p] = (long long)besti;
      if (zq_out && lane < D) {
        const int b = (int)(p / HW), hw = (int)(p % HW);
        zq_out[((long long)b * D + lane) * HW + hw] = (float)s_cbd[besti * (D + 1) + lane];
      }
    }
  }
}

__global__ __launch_bounds__(256) void embedding_kernel(const long long* __restrict__ tok, const float* __restrict__ cb,
                                                        float* __restrict__ out, long long N, int D, int K, int HW,
                                                        int nchw) {
  const long long total = N * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / D;
    const int d = (int)(i % D);
    const long long k = tok[n];
    const float v = (k >= 0 && k < K) ? cb[k * D + d] : __builtin_nanf("");
    if (nchw) {
      const long long b = n / HW;
      const int hw = (int)(n % HW);
      out[(b * D + d) * HW + hw] = v;
    } else {
      out[i] = v;
    }
  }
}

}  // namespace

extern "C" int spk_vq_readout_argmin(const uint8_t* z_ptc, const float* coef, const float* alpha,
                                     const float* codebook, long long* idx_out, float* zq_out_bdhw, float* xm_out,
                                     int T, int B, int D, int HW, int K, hipStream_t stream) {
  if (!z_ptc || !coef || !alpha || !codebook || !idx_out || T <= 0 || B <= 0 || D <= 0 || D > VQ_MAX_D || HW <= 0 ||
      K <= 0)
    return SPK_ERR_ARG;
  size_t lds = (size_t)(((K * (D + 1) + 1) & ~1)) * sizeof(float) + (size_t)K * sizeof(double);
  if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
  long long npos = (long long)B * HW;
  int grid = (int)((npos + 3) / 4);
  if (grid > 2048) grid = 2048;
  const size_t lds64 = ((size_t)K * (D + 1) + (size_t)K) * sizeof(double);
  if (D == 16 && T == 16 && lds64 <= 64 * 1024) {
    int g16 = (int)((npos + 15) / 16);
#ifndef SPK_VQ16_GRID
#define SPK_VQ16_GRID 4096
#endif
    if (g16 > SPK_VQ16_GRID) g16 = SPK_VQ16_GRID;            // (one group of four positions per wave: the loads of all of them overlap)
    hipLaunchKernelGGL(vq16_kernel, dim3(g16), dim3(256), lds64, stream, z_ptc, coef, alpha, codebook, idx_out, zq_out_bdhw,
                       xm_out, B, HW, K);
  } else if (D == 16)
    hipLaunchKernelGGL(vq_kernel<16>, dim3(grid), dim3(256), lds, stream, z_ptc, (const float*)nullptr, coef, alpha,
                       codebook, idx_out, zq_out_bdhw, xm_out, T, B, D, HW, K);
  else
    hipLaunchKernelGGL(vq_kernel<0>, dim3(grid), dim3(256), lds, stream, z_ptc, (const float*)nullptr, coef, alpha,
                       codebook, idx_out, zq_out_bdhw, xm_out, T, B, D, HW, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_vq_argmin(const float* flat_x, const float* codebook, long long* idx_out, long long N, int D, int K,
                             hipStream_t stream) {
  if (!flat_x || !codebook || !idx_out || N <= 0 || N > 0x7fffffff || D <= 0 || D > VQ_MAX_D || K <= 0)
    return SPK_ERR_ARG;
  size_t lds = (size_t)(((K * (D + 1) + 1) & ~1)) * sizeof(float) + (size_t)K * sizeof(double);
  if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
  int grid = (int)((N + 3) / 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(vq_kernel<0>, dim3(grid), dim3(256), lds, stream, (const uint8_t*)nullptr, flat_x,
                     (const float*)nullptr, (const float*)nullptr, codebook, idx_out, (float*)nullptr, (float*)nullptr,
                     1, (int)N, D, 1, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_embedding_fwd(const long long* tokens, const float* codebook, float* out, long long N, int D, int K,
                                 int HW, int nchw, hipStream_t stream) {
  if (!tokens || !codebook || !out || N <= 0 || D <= 0 || K <= 0 || (nchw && (HW <= 0 || N % HW))) return SPK_ERR_ARG;
  long long g = (N * D + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(embedding_kernel, dim3((int)g), dim3(256), 0, stream, tokens, codebook, out, N, D, K, HW, nchw);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
