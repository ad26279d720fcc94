/* CPU ORACLE (plain C restatement) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Scalar restatements of the index / threshold arithmetic of the hot path, used by tests/ to cross-check the
 * torch-based oracle (oracle/snn_ref.py) and the golden fixtures with an implementation that shares no code with
 * either PyTorch or the HIP kernels.
 *
 *   lif_ref        eval multi-step LIF, hard reset, decay_input  SJ/activation_based/neuron.py:799-811
 *   bn_fma_ref     eval BatchNorm as PyTorch-CPU evaluates it     (fixture F7; SJ/activation_based/layer.py:458-465)
 *   vq_argmin_ref  first-index argmin of |x|^2+|e|^2-2x.e         R/snn_model/vae_model.py:87-95
 *   psample_ref    mask update + exponential-race argmax          R/snn_model/vq_diffusion.py:113-124,134-140
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC -> oracle/_build/liboracle.so)
 */
#include <math.h>
#include <stdint.h>

/* x_seq [T][N], v [N] in/out, spikes [T][N] (0/1 as uint8) */
void lif_ref(const float* x_seq, float* v, uint8_t* spikes, int T, long long N, float tau, float v_th, float v_reset) {
  for (long long n = 0; n < N; ++n) {
    float vv = v[n];
    for (int t = 0; t < T; ++t) {
      float h = vv + (x_seq[(long long)t * N + n] - (vv - v_reset)) / tau;
      float s = h >= v_th ? 1.0f : 0.0f;
      vv = v_reset * s + (1.0f - s) * h;
      spikes[(long long)t * N + n] = (uint8_t)s;
    }
    v[n] = vv;
  }
}

/* y[m][c][hw] = fma(x, a_c, b_c), a = (1/sqrt(var+eps))*gamma, b = fma(-mean, a, beta) */
void bn_fma_ref(const float* x, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                float* y, long long M, int C, int HW) {
  for (int c = 0; c < C; ++c) {
    float inv = 1.0f / sqrtf(var[c] + eps);
    float a = inv * gamma[c];
    float b = fmaf(-mean[c], a, beta[c]);
    for (long long m = 0; m < M; ++m)
      for (int i = 0; i < HW; ++i) {
        long long k = (m * C + c) * HW + i;
        y[k] = fmaf(x[k], a, b);
      }
  }
}

/* flat_x [N][D], codebook [K][D] -> idx [N]; distances in double, first minimum wins */
void vq_argmin_ref(const float* flat_x, const float* cb, long long* idx, long long N, int D, int K) {
  for (long long n = 0; n < N; ++n) {
    double best = 1e300;
    int bi = 0;
    double x2 = 0.0;
    for (int d = 0; d < D; ++d) x2 += (double)flat_x[n * D + d] * flat_x[n * D + d];
    for (int k = 0; k < K; ++k) {
      double e2 = 0.0, dot = 0.0;
      for (int d = 0; d < D; ++d) {
        double e = cb[(long long)k * D + d];
        e2 += e * e;
        dot += (double)flat_x[n * D + d] * e;
      }
      double dist = x2 + e2 - 2.0 * dot;
      if (dist < best) { best = dist; bi = k; }
    }
    idx[n] = bi;
  }
}

/* logits [B][K][HW] (NCHW like the denoiser output), x_t [B*HW] int64 in/out, unmasked [B*HW] in/out,
 * u [B*HW], q [B*HW][K]:  changes = (u < 1/t) & ~unmasked; x0 = argmax_k softmax(l/temp)_k / q_k (first max) */
void psample_ref(const float* logits, long long* x_t, uint8_t* unmasked, int t, float temp, const float* u,
                 const float* q, int B, int HW, int K) {
  float inv_t = 1.0f / (float)t;
  for (int b = 0; b < B; ++b)
    for (int hw = 0; hw < HW; ++hw) {
      long long p = (long long)b * HW + hw;
      float mx = -INFINITY;
      for (int k = 0; k < K; ++k) {
        float l = logits[((long long)b * K + k) * HW + hw] / temp;
        if (l > mx) mx = l;
      }
      double se = 0.0;
      for (int k = 0; k < K; ++k) se += exp((double)(logits[((long long)b * K + k) * HW + hw] / temp - mx));
      double best = -1.0;
      int bi = 0;
      for (int k = 0; k < K; ++k) {
        double pk = exp((double)(logits[((long long)b * K + k) * HW + hw] / temp - mx)) / se;
        double r = pk / (double)q[p * K + k];
        if (r > best) { best = r; bi = k; }
      }
      if (u[p] < inv_t && !unmasked[p]) { unmasked[p] = 1; x_t[p] = bi; }
    }
}
