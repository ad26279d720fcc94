// Masked cross-entropy of the absorbing-diffusion training loss, forward and gradient in one pass.
// Reference: the loss tail of AbsorbingDiffusion._train_loss, R/snn_model/vq_diffusion.py:85-88 --
//   F.cross_entropy(logits.reshape(b, K, hw), x_0_ignore.reshape(b, hw).long(), ignore_index=-1, reduction='none')
// with x_0_ignore = x_0 where the token was masked by q_sample and -1 elsewhere (:70-73).
//
//   ce[b, p]       = -( (l[tgt] - m) - log sum_k exp(l_k - m) ),  m = max_k l_k          (0 where tgt < 0)
//   dlogits[b,k,p] = coef[b] * (softmax_k - [k == tgt])                                   (0 where tgt < 0)
// coef[b] is the per-sample factor the caller derives from the loss weighting (reweighted ELBO: (1 - t/T) / (ln 2 hw B)).
// One workgroup per sample: the [K][hw] logit tile (25 KB at K=128, 7x7) is staged in LDS with coalesced loads; one
// thread per position does the log-sum-exp over K from LDS (stride hw: conflict-free for odd hw, 2-way at 8x8).
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

constexpr int CE_TPB = 256;
constexpr int CE_MAX_TILE = 16384;      // floats of LDS for the logit tile (64 KB)

__global__ __launch_bounds__(CE_TPB) void masked_ce_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                           const float* __restrict__ coef, float* __restrict__ ce_out,
                                                           float* __restrict__ dlogits, int K, int HW) {
  extern __shared__ float tile[];                  // [K*HW] logits, then [HW] log-sum-exp offsets, then [HW] targets
  float* lse = tile + K * HW;
  int* tgt = reinterpret_cast<int*>(lse + HW);
  const int b = blockIdx.x, n = K * HW;
  const float* src = logits + (long long)b * n;
  for (int i = threadIdx.x; i < n; i += CE_TPB) tile[i] = src[i];
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += CE_TPB) {
    float m = tile[p];
    for (int k = 1; k < K; ++k) m = fmaxf(m, tile[k * HW + p]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(tile[k * HW + p] - m);
    const float off = m + logf(s);               // l - off = log-softmax
    const float tf = target[(long long)b * HW + p];
    const int tg = (tf >= 0.0f && tf < (float)K) ? (int)tf : -1;
    lse[p] = off; tgt[p] = tg;
    ce_out[(long long)b * HW + p] = tg >= 0 ? -((tile[tg * HW + p] - m) - logf(s)) : 0.0f;
  }
  __syncthreads();
  if (dlogits) {
    const float cb = coef[b];
    float* dst = dlogits + (long long)b * n;
    for (int i = threadIdx.x; i < n; i += CE_TPB) {
      const int k = i / HW, p = i - k * HW;
      const int tg = tgt[p];
      dst[i] = tg < 0 ? 0.0f : cb * (expf(tile[i] - lse[p]) - (k == tg ? 1.0f : 0.0f));
    }
  }
}

// Same arithmetic without the LDS tile, for [K][hw] tiles beyond 64 KB (large codebooks): one thread per position,
// logits re-read from global memory (threads of a wave read adjacent positions of one class: coalesced).
__global__ __launch_bounds__(CE_TPB) void masked_ce_global_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                                  const float* __restrict__ coef, float* __restrict__ ce_out,
                                                                  float* __restrict__ dlogits, int K, int HW) {
  const int b = blockIdx.x;
  const float* src = logits + (long long)b * K * HW;
  for (int p = threadIdx.x; p < HW; p += CE_TPB) {
    float m = src[p];
    for (int k = 1; k < K; ++k) m = fmaxf(m, src[k * HW + p]);
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += expf(src[k * HW + p] - m);
    const float off = m + logf(s);
    const float tf = target[(long long)b * HW + p];
    const int tg = (tf >= 0.0f && tf < (float)K) ? (int)tf : -1;
    ce_out[(long long)b * HW + p] = tg >= 0 ? -((src[tg * HW + p] - m) - logf(s)) : 0.0f;
    if (dlogits) {
      const float cb = coef[b];
      float* dst = dlogits + (long long)b * K * HW;
      for (int k = 0; k < K; ++k)
        dst[k * HW + p] = tg < 0 ? 0.0f : cb * (expf(src[k * HW + p] - off) - (k == tg ? 1.0f : 0.0f));
    }
  }
}

}  // namespace

extern "C" int spk_masked_ce(const float* logits, const float* target, const float* coef, float* ce_out, float* dlogits, int B,
                             int K, int HW, hipStream_t stream) {
  if (!logits || !target || !ce_out || (dlogits && !coef) || B <= 0 || K <= 0 || HW <= 0) return SPK_ERR_ARG;
  if ((long long)K * HW + 2LL * HW > CE_MAX_TILE) {
    hipLaunchKernelGGL(masked_ce_global_kernel, dim3(B), dim3(CE_TPB), 0, stream, logits, target, coef, ce_out, dlogits, K, HW);
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  const size_t lds = ((size_t)K * HW + 2 * (size_t)HW) * sizeof(float);
  hipLaunchKernelGGL(masked_ce_kernel, dim3(B), dim3(CE_TPB), lds, stream, logits, target, coef, ce_out, dlogits, K, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
