// Training branch of VectorQuantizer.forward (R/snn_model/vae_model.py:61-85) as four launches instead of
// the ~40 element-wise / reduce / embedding-backward launches autograd needs for the same algebra:
//   x_m   = (1 - alpha) * sum_t x[t] * coef[t] + alpha * sum_t x[t] / T          (spk_vq_train_readout, NHWC rows [B*h*w, D])
//   idx   = argmin_k ||x_m - e_k||^2                                                (spk_vq_argmin, unchanged)
//   q     = E[idx];  loss_1 = mse(q, sg(x_m)) + beta * mse(x_m, sg(q));  out = x_m + sg(q - x_m)      (spk_vq_train_quant)
// and backward, one launch (spk_vq_train_bwd):
//   g_xm  = g_out + g_loss * beta * 2 (x_m - q) / (N D)          (straight-through estimator + commitment term)
//   g_x[t] = g_xm * (1 - alpha) * coef[t] + g_xm * alpha / T
//   g_alpha = sum g_xm * (sum_t x[t] / T - sum_t x[t] * coef[t])
//   g_E[k] = g_loss * 2 / (N D) * sum_{rows with idx = k} (q - x_m)     (one workgroup per code, fixed order: deterministic,
//            where the framework's embedding backward takes 110 us of atomics at the reference's batch of 32)
// fp32 element-wise arithmetic in the reference's operation order (-ffp-contract=off); the two mean-square reductions and the
// alpha gradient are summed in fp64 in a fixed order (partials per workgroup, combined by the last one to finish).
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

constexpr int VT_MAX_D = 64, VT_MAX_BLOCKS = 256;

__global__ __launch_bounds__(256) void vq_train_readout_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                               const float* __restrict__ alpha, float* __restrict__ xm,
                                                               float* __restrict__ dxa, int T, int B, int D, int HW) {
  const long long n_all = (long long)B * D * HW;
  const float al = alpha[0];
  const float a1 = 1.0f - al;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const long long r = i / HW;
    const int d = (int)(r % D), b = (int)(r / D);
    float mo = 0.f, sm = 0.f;
    for (int t = 0; t < T; ++t) {                        // (the accumulation order of spk_memout_fwd / torch.sum over dim 0)
      const float v = x[(long long)t * n_all + i];
      mo = mo + v * coef[t];
      sm = sm + v;
    }
    const float smT = sm / (float)T;
    const long long o = ((long long)b * HW + p) * D + d;
    xm[o] = a1 * mo + (al * sm) / (float)T;
    dxa[o] = smT - mo;
  }
}

// partial sums of a workgroup -> ws_part[block]; the last workgroup (ticket) adds them in block order
__device__ __forceinline__ bool vt_block_sum(double v, double* red, double* ws_part, unsigned* ticket, double& total) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  __shared__ bool last;
  if (tid == 0) {
    ws_part[blockIdx.x] = red[0];
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return false;
  if (tid == 0) {
    __threadfence();
    double s = 0.0;
    for (unsigned k = 0; k < gridDim.x; ++k) s += ((volatile double*)ws_part)[k];
    total = s;
    *ticket = 0u;
  }
  return tid == 0;
}

__global__ __launch_bounds__(256) void vq_train_quant_kernel(const float* __restrict__ xm, const long long* __restrict__ idx,
                                                             const float* __restrict__ E, float* __restrict__ out_bdhw,
                                                             float* __restrict__ loss, float beta, double* ws_part,
                                                             unsigned* ticket, long long N, int D, int HW) {
  __shared__ double red[256];
  double part = 0.0;
  const long long total = N * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / D;
    const int d = (int)(i - row * D);
    const float q = E[idx[row] * D + d], v = xm[i];
    const float df = q - v;
    part += (double)(df * df);
    const long long b = row / HW;
    const int p = (int)(row - b * HW);
    out_bdhw[(b * D + d) * HW + p] = v + df;               // x_m + (q - x_m): the straight-through value
  }
  double tot = 0.0;
  if (vt_block_sum(part, red, ws_part, ticket, tot)) {
    const float m = (float)(tot / (double)total);          // both mean-square terms have this value
    loss[0] = m + beta * m;
  }
}

// blocks [0, nb_x): element-wise part (g_x, partial g_alpha); blocks [nb_x, nb_x + K): one codebook row each
__global__ __launch_bounds__(256) void vq_train_bwd_kernel(const float* __restrict__ gout_bdhw, const float* __restrict__ gloss,
                                                           const float* __restrict__ xm, const long long* __restrict__ idx,
                                                           const float* __restrict__ E, const float* __restrict__ dxa,
                                                           const float* __restrict__ coef, const float* __restrict__ alpha,
                                                           float beta, float* __restrict__ gx, float* __restrict__ galpha,
                                                           float* __restrict__ gE, double* ws_part, unsigned* ticket, int nb_x,
                                                           int T, long long N, int D, int HW, int K) {
  __shared__ double red[256];
  const float gl = gloss ? gloss[0] : 0.f;
  const long long total = N * D;
  const float inv = 2.0f / (float)total;
  if ((int)blockIdx.x >= nb_x) {
    // g_E[k][d]: thread (row lane, d) walks the rows in order; the row lanes are combined in a fixed order
    const int k = (int)blockIdx.x - nb_x, tid = threadIdx.x;
    const int RL = 256 / D > 0 ? 256 / D : 1;              // row lanes (D <= 64)
    const int d = tid % D, rl = tid / D;
    double acc = 0.0;
    if (rl < RL)
      for (long long r = rl; r < N; r += RL)
        if (idx[r] == k) acc += (double)(E[(long long)k * D + d] - xm[r * D + d]);
    red[tid] = rl < RL ? acc : 0.0;
    __syncthreads();
    if (tid < D) {
      double s = 0.0;
      for (int j = 0; j < RL; ++j) s += red[j * D + tid];
      gE[(long long)k * D + tid] = gl * inv * (float)s;
    }
    return;
  }
  const float al = alpha[0];
  const float a1 = 1.0f - al;
  double part = 0.0;
  const long long n_all = total;                           // elements of one time step of x
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)nb_x * blockDim.x) {
    // i runs over the NCHW elements (b, d, p) so that the T stores per element are coalesced
    const int p = (int)(i % HW);
    const long long r = i / HW;
    const int d = (int)(r % D);
    const long long b = r / D;
    const long long row = b * HW + p, o = row * D + d;
    const float v = xm[o], q = E[idx[row] * D + d];
    const float g = gout_bdhw[i] + gl * beta * inv * (v - q);
    part += (double)g * (double)dxa[o];
    const float g1 = g * a1, g2 = (g * al) / (float)T;
    for (int t = 0; t < T; ++t) gx[(long long)t * n_all + i] = g1 * coef[t] + g2;
  }
  // (only the element-wise blocks take part in the ticket: gridDim.x is not their count)
  const int tid = threadIdx.x;
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    ws_part[blockIdx.x] = red[0];
    __threadfence();
    if (atomicAdd(ticket, 1u) == (unsigned)nb_x - 1) {
      __threadfence();
      double s = 0.0;
      for (int k = 0; k < nb_x; ++k) s += ((volatile double*)ws_part)[k];
      galpha[0] = (float)s;
      *ticket = 0u;
    }
  }
}

// PSP losses of the same branch (R/snn_model/vae_model.py:79-84; filter: R/snn_model/snn_layers.py:12-26):
//   loss_2 = mean((psp(q) - sg(psp(x)))^2) + beta * mean((sg(psp(q)) - psp(x))^2),   syn_t = syn_{t-1} + (in_t - syn_{t-1}) / tau_s.
// Forward: both filters run in registers, nothing but the sum of squares leaves the launch.  Backward: the filters are run
// again (T differences kept in registers), then both adjoint filters backwards over t: g_q[t] and g_x[t] in one launch.
template <int TMAX>
__global__ __launch_bounds__(256) void psp_loss_fwd_kernel(const float* __restrict__ q, const float* __restrict__ x,
                                                           float* __restrict__ loss, float beta, float tau, double* ws_part,
                                                           unsigned* ticket, int T, long long N) {
  __shared__ double red[256];
  double part = 0.0;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    float sq = 0.f, sx = 0.f;
    for (int t = 0; t < T; ++t) {
      sq = sq + (q[(long long)t * N + n] - sq) / tau;
      sx = sx + (x[(long long)t * N + n] - sx) / tau;
      const float d = sq - sx;
      part += (double)(d * d);
    }
  }
  double tot = 0.0;
  if (vt_block_sum(part, red, ws_part, ticket, tot)) {
    const float m = (float)(tot / ((double)T * (double)N));
    loss[0] = m + beta * m;
  }
}

template <int TMAX>
__global__ __launch_bounds__(256) void psp_loss_bwd_kernel(const float* __restrict__ q, const float* __restrict__ x,
                                                           const float* __restrict__ gloss, float* __restrict__ gq,
                                                           float* __restrict__ gx, float beta, float tau, int T, long long N) {
  const float inv_tau = 1.0f / tau, carry = 1.0f - inv_tau;
  const float c = gloss[0] * (2.0f / ((float)T * (float)N));
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    float d[TMAX];
    float sq = 0.f, sx = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
        sq = sq + (q[(long long)t * N + n] - sq) / tau;
        sx = sx + (x[(long long)t * N + n] - sx) / tau;
        d[t] = sq - sx;
      }
    }
    float aq = 0.f, ax = 0.f;                               // the adjoint filters (spk_psp, backward)
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
      const int t = TMAX - 1 - i;
      if (t < T) {
        const float Gq = aq + c * d[t], Gx = ax - (beta * c) * d[t];
        gq[(long long)t * N + n] = Gq * inv_tau;
        gx[(long long)t * N + n] = Gx * inv_tau;
        aq = Gq * carry;
        ax = Gx * carry;
      }
    }
  }
}

// Reconstruction loss of SNN_VQVAE.forward in training (R/snn_model/vae_model.py:189-196): x_recon = tanh(sum_t y[t] * coef[t]),
// loss = mean((x_recon - image)^2); backward g_y[t] = g_loss * 2 (x_recon - image) / N * (1 - x_recon^2) * coef[t].
__global__ __launch_bounds__(256) void recon_loss_fwd_kernel(const float* __restrict__ y, const float* __restrict__ coef,
                                                             const float* __restrict__ img, float* __restrict__ xr,
                                                             float* __restrict__ loss, double* ws_part, unsigned* ticket, int T,
                                                             long long N) {
  __shared__ double red[256];
  double part = 0.0;
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    float m = 0.f;
    for (int t = 0; t < T; ++t) m = m + y[(long long)t * N + n] * coef[t];
    const float r = tanhf(m);
    xr[n] = r;
    const float d = r - img[n];
    part += (double)(d * d);
  }
  double tot = 0.0;
  if (vt_block_sum(part, red, ws_part, ticket, tot)) loss[0] = (float)(tot / (double)N);
}

__global__ __launch_bounds__(256) void recon_loss_bwd_kernel(const float* __restrict__ xr, const float* __restrict__ img,
                                                             const float* __restrict__ coef, const float* __restrict__ gloss,
                                                             float* __restrict__ gy, int T, long long N) {
  const float c = gloss[0] * (2.0f / (float)N);
  for (long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (long long)gridDim.x * blockDim.x) {
    const float r = xr[n];
    const float g = (c * (r - img[n])) * (1.0f - r * r);
    for (int t = 0; t < T; ++t) gy[(long long)t * N + n] = g * coef[t];
  }
}

static int vt_blocks(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > VT_MAX_BLOCKS ? VT_MAX_BLOCKS : g));
}

}  // namespace

extern "C" long long spk_vq_train_ws_bytes(void) { return (long long)VT_MAX_BLOCKS * sizeof(double) + 64; }

extern "C" int spk_vq_train_readout(const float* x_seq, const float* coef, const float* alpha, float* xm_out, float* dxa_out,
                                    int T, int B, int D, int HW, hipStream_t stream) {
  if (!x_seq || !coef || !alpha || !xm_out || !dxa_out || T <= 0 || B <= 0 || D <= 0 || HW <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(vq_train_readout_kernel, dim3(vt_blocks((long long)B * D * HW)), dim3(256), 0, stream, x_seq, coef, alpha,
                     xm_out, dxa_out, T, B, D, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_vq_train_quant(const float* xm, const long long* idx, const float* codebook, float* out_bdhw, float* loss_out,
                                  float beta, void* ws, long long N, int D, int HW, hipStream_t stream) {
  if (!xm || !idx || !codebook || !out_bdhw || !loss_out || !ws || N <= 0 || D <= 0 || HW <= 0 || N % HW) return SPK_ERR_ARG;
  double* part = reinterpret_cast<double*>(ws);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + VT_MAX_BLOCKS);
  hipLaunchKernelGGL(vq_train_quant_kernel, dim3(vt_blocks(N * D)), dim3(256), 0, stream, xm, idx, codebook, out_bdhw, loss_out,
                     beta, part, ticket, N, D, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_vq_train_bwd(const float* gout_bdhw, const float* gloss_or_null, const float* xm, const long long* idx,
                                const float* codebook, const float* dxa, const float* coef, const float* alpha, float beta,
                                float* gx_seq, float* galpha_out, float* gcodebook_out, void* ws, int T, long long N, int D, int HW,
                                int K, hipStream_t stream) {
  if (!gout_bdhw || !xm || !idx || !codebook || !dxa || !coef || !alpha || !gx_seq || !galpha_out || !gcodebook_out || !ws ||
      T <= 0 || N <= 0 || D <= 0 || HW <= 0 || K <= 0 || N % HW)
    return SPK_ERR_ARG;
  if (D > VT_MAX_D) return SPK_ERR_UNSUPPORTED;
  double* part = reinterpret_cast<double*>(ws);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + VT_MAX_BLOCKS);
  const int nb_x = vt_blocks(N * D);
  hipLaunchKernelGGL(vq_train_bwd_kernel, dim3(nb_x + K), dim3(256), 0, stream, gout_bdhw, gloss_or_null, xm, idx, codebook, dxa,
                     coef, alpha, beta, gx_seq, galpha_out, gcodebook_out, part, ticket, nb_x, T, N, D, HW, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_psp_loss_fwd(const float* q_seq, const float* x_seq, float* loss_out, float beta, float tau_s, void* ws, int T,
                                long long N, hipStream_t stream) {
  if (!q_seq || !x_seq || !loss_out || !ws || T <= 0 || N <= 0 || !(tau_s > 0.f)) return SPK_ERR_ARG;
  double* part = reinterpret_cast<double*>(ws);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + VT_MAX_BLOCKS);
  hipLaunchKernelGGL((psp_loss_fwd_kernel<16>), dim3(vt_blocks(N)), dim3(256), 0, stream, q_seq, x_seq, loss_out, beta, tau_s, part,
                     ticket, T, N);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_psp_loss_bwd(const float* q_seq, const float* x_seq, const float* gloss, float* gq_seq, float* gx_seq, float beta,
                                float tau_s, int T, long long N, hipStream_t stream) {
  if (!q_seq || !x_seq || !gloss || !gq_seq || !gx_seq || T <= 0 || N <= 0 || !(tau_s > 0.f)) return SPK_ERR_ARG;
  if (T > 16) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((psp_loss_bwd_kernel<16>), dim3(vt_blocks(N)), dim3(256), 0, stream, q_seq, x_seq, gloss, gq_seq, gx_seq, beta,
                     tau_s, T, N);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_recon_loss_fwd(const float* y_seq, const float* coef, const float* image, float* xr_out, float* loss_out, void* ws,
                                  int T, long long N, hipStream_t stream) {
  if (!y_seq || !coef || !image || !xr_out || !loss_out || !ws || T <= 0 || N <= 0) return SPK_ERR_ARG;
  double* part = reinterpret_cast<double*>(ws);
  unsigned* ticket = reinterpret_cast<unsigned*>(part + VT_MAX_BLOCKS);
  hipLaunchKernelGGL(recon_loss_fwd_kernel, dim3(vt_blocks(N)), dim3(256), 0, stream, y_seq, coef, image, xr_out, loss_out, part,
                     ticket, T, N);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_recon_loss_bwd(const float* xr, const float* image, const float* coef, const float* gloss, float* gy_seq, int T,
                                  long long N, hipStream_t stream) {
  if (!xr || !image || !coef || !gloss || !gy_seq || T <= 0 || N <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(recon_loss_bwd_kernel, dim3(vt_blocks(N)), dim3(256), 0, stream, xr, image, coef, gloss, gy_seq, T, N);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
