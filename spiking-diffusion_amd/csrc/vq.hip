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

// block = 256 threads = 4 waves; one wave per latent position; codebook staged in LDS once per block.
__global__ __launch_bounds__(256) void vq_kernel(const uint8_t* __restrict__ z, const float* __restrict__ x_in,
                                                 const float* __restrict__ coef,
                                                 const float* __restrict__ alpha_p, const float* __restrict__ cb,
                                                 long long* __restrict__ idx_out, float* __restrict__ zq_out,
                                                 float* __restrict__ xm_out, int T, int B, int D, int HW, int K) {
  extern __shared__ float lds[];
  float* s_cb = lds;                         // [K][D+1]
  float* s_x = s_cb + K * (D + 1);           // [4][VQ_MAX_D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < K * D; i += blockDim.x) s_cb[(i / D) * (D + 1) + (i % D)] = cb[i];
  __syncthreads();
  const float alpha = alpha_p ? alpha_p[0] : 0.f;
  const float one_m_alpha = 1.0f - alpha;
  const long long npos = (long long)B * HW;
  for (long long p0 = (long long)blockIdx.x * 4; p0 < npos; p0 += (long long)gridDim.x * 4) {
    const long long p = p0 + wave;
    const bool live = p < npos;
    if (live && lane < D && x_in) {
      s_x[wave * VQ_MAX_D + lane] = x_in[p * D + lane];
    } else if (live && lane < D) {
      const uint8_t* zp = z + p * T * D + lane;
      float m = 0.f, cnt = 0.f;
      for (int t = 0; t < T; ++t) {
        float s = (float)zp[t * D];
        m = m + s * coef[t];
        cnt = cnt + s;
      }
      float x = one_m_alpha * m + (alpha * cnt) / (float)T;
      s_x[wave * VQ_MAX_D + lane] = x;
      if (xm_out) xm_out[p * D + lane] = x;
    }
    __syncthreads();
    double best = 1.0e300;
    int besti = 0x7fffffff;
    if (live) {
      double x2 = 0.0;
      for (int d = 0; d < D; ++d) { double xv = s_x[wave * VQ_MAX_D + d]; x2 += xv * xv; }
      for (int k = lane; k < K; k += 64) {
        double e2 = 0.0, dot = 0.0;
        for (int d = 0; d < D; ++d) {
          double e = s_cb[k * (D + 1) + d];
          e2 += e * e;
          dot += (double)s_x[wave * VQ_MAX_D + d] * e;
        }
        double dist = x2 + e2 - 2.0 * dot;
        if (dist < best) { best = dist; besti = k; }       // k increasing per lane: first minimum kept
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      double ob = __shfl_xor(best, off);
      int oi = __shfl_xor(besti, off);
      if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (live) {
      if (lane == 0) idx_out[p] = (long long)besti;
      if (zq_out && lane < D) {
        const int b = (int)(p / HW), hw = (int)(p % HW);
        zq_out[((long long)b * D + lane) * HW + hw] = s_cb[besti * (D + 1) + lane];
      }
    }
    __syncthreads();
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
  size_t lds = (size_t)(K * (D + 1) + 4 * VQ_MAX_D) * sizeof(float);
  if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
  long long npos = (long long)B * HW;
  int grid = (int)((npos + 3) / 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(vq_kernel, dim3(grid), dim3(256), lds, stream, z_ptc, (const float*)nullptr, coef, alpha,
                     codebook, idx_out, zq_out_bdhw, xm_out, T, B, D, HW, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_vq_argmin(const float* flat_x, const float* codebook, long long* idx_out, long long N, int D, int K,
                             hipStream_t stream) {
  if (!flat_x || !codebook || !idx_out || N <= 0 || N > 0x7fffffff || D <= 0 || D > VQ_MAX_D || K <= 0)
    return SPK_ERR_ARG;
  size_t lds = (size_t)(K * (D + 1) + 4 * VQ_MAX_D) * sizeof(float);
  if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
  int grid = (int)((N + 3) / 4);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(vq_kernel, dim3(grid), dim3(256), lds, stream, (const uint8_t*)nullptr, flat_x,
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
