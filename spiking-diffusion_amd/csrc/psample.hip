// One reverse step of the absorbing-state discrete diffusion sampler, after the denoiser produced the logits:
//   R/snn_model/vq_diffusion.py:113-124 (where to unmask), :134-140 (temperature, Categorical sample, scatter).
//
//   changes  = (u < 1/t) & ~unmasked ;  unmasked |= changes
//   x0_hat   = Categorical(logits = logits/temp).sample()
//            = argmax_k softmax(l - logsumexp l)_k / q_k ,  q ~ Exp(1)      (torch.multinomial one-draw fast path)
//   x_t[changes] = x0_hat[changes]
//
// Noise sources: (a) injected u [B*HW] and q [B*HW*K] (parity mode: the host draws them with the reference's CPU
// generator in the reference's order), or (b) on-device Philox4x32-10 keyed by (seed, offset)
// (throughput mode).  One wave per latent position; lanes stride over the K classes; wave reductions by DPP shuffles.
#include "spk_common.h"
#include "den_common.h"
#include "psample_common.h"
#include "../../include/spkdiff.h"
#include <math.h>

namespace {

__device__ __forceinline__ bool __any_sync_quad(bool v) {      // OR over the 4 lanes of a quad
  int x = v ? 1 : 0;
  x |= __shfl_xor(x, 1);
  x |= __shfl_xor(x, 2);
  return x != 0;
}

// KPL = classes per lane (K <= 64 * KPL): 4 covers the reference's default codebook (--codebook_size 128, R/main.py:58);
// 8 / 16 / 32 are instantiated for larger codebooks (K <= 2048)
template <int KPL>
__global__ __launch_bounds__(256) void psample_kernel(const float* __restrict__ logits, long long* __restrict__ x_t,
                                                      uint8_t* __restrict__ unmasked, int t, float temp,
                                                      const float* __restrict__ u_in, const float* __restrict__ q_in,
                                                      unsigned long long seed, unsigned long long offset,
                                                      const unsigned long long* __restrict__ philox_state,
                                                      long long* __restrict__ x0_hat_out,
                                                      const int* __restrict__ active, const int* __restrict__ n_active,
                                                      int B, int HW, int K, float* __restrict__ next_input, float t_next) {
  // a captured (hipGraph) launch bakes its arguments: the per-call part of the Philox counter then comes from a
  // 2-word device buffer {seed, base offset} the host updates before each replay
  if (philox_state) { seed = philox_state[0]; offset += philox_state[1]; }
  const int lane = threadIdx.x & 63;
  // active-set form (spk_select_active): logits hold one slot per ACTIVE image (slot s = image active[s]); noise, x_t
  // and unmasked stay indexed by image, so the draws are those of the dense form
  const int Bn = active ? (*n_active < B ? *n_active : B) : B;
  const long long npos = (long long)Bn * HW;
  const float inv_t = 1.0f / (float)t;
  for (long long ps = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); ps < npos; ps += (long long)gridDim.x * 4) {
    const int b = (int)(ps / HW), hw = (int)(ps % HW);            // b = logits slot
    const long long p = active ? (long long)active[b] * HW + hw : ps;   // image position: noise / state index
    if (!x0_hat_out) {
      // only positions that change consume the sample (:140): skip the rest before the softmax (wave-uniform test;
      // the noise is counter-based / injected per position, so skipping a draw does not move any other)
      float u;
      if (u_in) u = u_in[p];
      else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)p * (unsigned long long)K, 0u, r); u = u01_open_right(r[0]); }
      if (!((u < inv_t) && !unmasked[p])) {
        // (dense form) the denoiser input of the next reverse step, cat(x_t, t - 1): this position keeps its token
        if (next_input && lane == 0) {
          next_input[((long long)b * 2 + 0) * HW + hw] = (float)x_t[p];
          next_input[((long long)b * 2 + 1) * HW + hw] = t_next;
        }
        continue;
      }
    }
    float l[KPL];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      // (unconditional load from a clamped index + select: hipcc waits for a conditional load on its own)
      const float lg = logits[((long long)b * K + (k < K ? k : K - 1)) * HW + hw];
      l[j] = k < K ? lg / temp : -INFINITY;
      mx = fmaxf(mx, l[j]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) se += (lane + 64 * j < K) ? expf(l[j] - mx) : 0.f;
    se = wave_sum(se);
    const float lse = mx + logf(se);
    // Categorical normalises logits, then .probs = softmax(normalised logits)
    float e[KPL];
    float mx2 = -INFINITY;
#pragma unroll
    for (int j = 0; j < KPL; ++j) { l[j] = l[j] - lse; mx2 = fmaxf(mx2, l[j]); }
    mx2 = wave_max(mx2);
    float se2 = 0.f;
#pragma unroll
    for (int j = 0; j < KPL; ++j) { e[j] = (lane + 64 * j < K) ? expf(l[j] - mx2) : 0.f; se2 += e[j]; }
    se2 = wave_sum(se2);
    float best = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const int k = lane + 64 * j;
      if (k < K) {
        float q;
        if (q_in) {
          q = q_in[p * K + k];
        } else {
          uint32_t r[4];
          philox4x32(seed, offset + (unsigned long long)(p * K + k), 1u, r);
          q = -logf(u01_open_left(r[0]));
        }
        const float ratio = (e[j] / se2) / q;
        if (ratio > best) { best = ratio; besti = k; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      float ob = __shfl_xor(best, off);
      int oi = __shfl_xor(besti, off);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) {
      float u;
      if (u_in) {
        u = u_in[p];
      } else {
        uint32_t r[4];
        philox4x32(seed, offset + (unsigned long long)p * (unsigned long long)K, 0u, r);
        u = u01_open_right(r[0]);
      }
      bool ch = (u < inv_t) && !unmasked[p];
      if (ch) { unmasked[p] = 1; x_t[p] = (long long)besti; }
      if (x0_hat_out) x0_hat_out[p] = (long long)besti;
      if (next_input && lane == 0) {
        next_input[((long long)b * 2 + 0) * HW + hw] = ch ? (float)besti : (float)x_t[p];
        next_input[((long long)b * 2 + 1) * HW + hw] = t_next;
      }
    }
  }
}

// Denoiser input map, R/snn_model/vq_diffusion.py:195-197:  cat(x, ones_like(x) * t[:,None,None,None]) -> [B,2,h,w]
// x comes either as float [B,1,h,w] (the module API) or as the int64 token state x_t of the sampler.
__global__ void den_input_kernel(const float* __restrict__ xf, const long long* __restrict__ xi,
                                 const long long* __restrict__ t_vec, long long t_scalar, float* __restrict__ out,
                                 const int* __restrict__ active, const int* __restrict__ n_active, int B, int HW) {
  const int Bn = active ? (*n_active < B ? *n_active : B) : B;
  const int total = Bn * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / HW, hw = i % HW;                            // output slot
    const int src = active ? active[b] * HW + hw : i;             // active-set form: slot b holds image active[b]
    const float x = xf ? xf[src] : (float)xi[src];
    const float tv = (float)(t_vec ? t_vec[active ? active[b] : b] : t_scalar);
    out[((long long)b * 2 + 0) * HW + hw] = x;
    out[((long long)b * 2 + 1) * HW + hw] = 1.0f * tv;
  }
}

// Which images does reverse step t touch?  R/snn_model/vq_diffusion.py:113-124 computes `changes` BEFORE the denoiser
// call and scatters x_0_hat only there (:140): for an image without a change at this step the denoiser output is never
// read and x_t / unmasked stay as they are.  This kernel evaluates the same test (same u: injected or the same Philox
// counters as spk_psample_step) per image and writes the ascending list of images with at least one change plus its
// length; the per-step kernels then run on that list only.  The list is ordered (deterministic slots).
// Eight threads per image (positions q, q + 8, ...: a Philox depth of HW / 8; the `unmasked` bytes of a thread are requested
// together), one-wave workgroups spread over the CUs; every image's flag goes to active[b], and the workgroup that finishes last
// (ticket) compacts the flags in place into the ordered list: slots are deterministic.  (The first form ran as ONE 1024-thread
// workgroup with a chain of conditional loads per thread: 10.7 us per reverse step.)  The ticket is n_active[1]: zero before the
// first call, left zero by every call (the caller's buffer, so calls on different streams with different buffers do not meet).

__global__ __launch_bounds__(64) void select_active_kernel(const uint8_t* __restrict__ unmasked, int t,
                                                           const float* __restrict__ u_in, unsigned long long seed,
                                                           unsigned long long offset,
                                                           const unsigned long long* __restrict__ philox_state,
                                                           int* __restrict__ active, int* __restrict__ n_active, int B, int HW, int K) {
  if (philox_state) { seed = philox_state[0]; offset += philox_state[1]; }
  const float inv_t = 1.0f / (float)t;
  const int lane = threadIdx.x;
  const int gid = blockIdx.x * 64 + lane;
  const int b = gid >> 3, q = gid & 7;
  bool any = false;
  if (b < B) {
    uint8_t um[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int hw = q + 8 * k;
      const uint8_t ld = unmasked[(long long)b * HW + (hw < HW ? hw : 0)];
      um[k] = hw < HW ? ld : (uint8_t)1;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (!um[k]) {
        const long long p = (long long)b * HW + q + 8 * k;
        float u;
        if (u_in) u = u_in[p];
        else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)p * (unsigned long long)K, 0u, r); u = u01_open_right(r[0]); }
        any = any || (u < inv_t);
      }
    }
    for (int hw = q + 64; hw < HW; hw += 8) {                   // (latents beyond 64 positions)
      const long long p = (long long)b * HW + hw;
      if (unmasked[p]) continue;
      float u;
      if (u_in) u = u_in[p];
      else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)p * (unsigned long long)K, 0u, r); u = u01_open_right(r[0]); }
      any = any || (u < inv_t);
    }
  }
  int f = any ? 1 : 0;
  f |= __shfl_xor(f, 1); f |= __shfl_xor(f, 2); f |= __shfl_xor(f, 4);
  if (q == 0 && b < B) active[b] = f;
  __threadfence();
  int last = 0;
  if (lane == 0) last = atomicAdd(reinterpret_cast<unsigned*>(n_active + 1), 1u) == gridDim.x - 1 ? 1 : 0;
  last = __shfl(last, 0);
  if (!last) return;
  __threadfence();
  int base = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int bi = b0 + lane;
    const int fl = bi < B ? __hip_atomic_load(&active[bi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const unsigned long long m = __ballot(fl != 0);
    // (every lane of the wave has read its flag before any lane writes: list entries land at or below b0 + lane)
    if (fl) active[base + __popcll(m & ((1ull << lane) - 1ull))] = bi;
    base += __popcll(m);
  }
  if (lane == 0) { n_active[0] = base; n_active[1] = 0; }
}

// Which POSITIONS of an active image does reverse step t need from each denoiser layer?  The sampler reads the logits only
// at the image's `changes` positions C (:134-140); conv6 is 3x3, so the last spiking layer is needed on dilate(C, 1), the
// one before it on dilate(C, 2), ... (3x3 convolutions, zero padding).  One wave per active slot; record (r - 1, slot)
// (64 bytes) describes radius r: bytes 0..47 the ascending list of needed positions below 48 (padded with its last entry),
// byte 48 the list length n, byte 49 its tile-count class ceil(n / 8) (1..6; a 32-row tile is two positions x 16 steps, four
// waves), byte 50 whether position 48 is needed.  (The last position of an odd latent is computed by the tail launch of the
// MFMA kernel either way.)  The workgroup that finishes last groups the slots of every radius by class (the MFMA kernel
// runs one item loop per class) and re-arms the ticket.  Buffer layout: den_common.h.
__global__ __launch_bounds__(256) void select_needed_kernel(const uint8_t* __restrict__ unmasked, int t,
                                                            const float* __restrict__ u_in, unsigned long long seed,
                                                            unsigned long long offset,
                                                            const unsigned long long* __restrict__ philox_state,
                                                            const int* __restrict__ active, const int* __restrict__ n_active,
                                                            uint8_t* __restrict__ need, int B, int H, int W, int R, int K) {
  __shared__ int s_cnt[8][8];
  __shared__ int s_last;
  if (philox_state) { seed = philox_state[0]; offset += philox_state[1]; }
  const int lane = threadIdx.x & 63, slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Bn = *n_active < B ? *n_active : B;
  const int HW = H * W;
  if (slot < Bn) {
    const int b = active[slot];
    bool ch = false;
    if (lane < HW) {
      const long long p = (long long)b * HW + lane;
      if (!unmasked[p]) {
        float u;
        if (u_in) u = u_in[p];
        else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)p * (unsigned long long)K, 0u, r); u = u01_open_right(r[0]); }
        ch = u < 1.0f / (float)t;
      }
    }
    unsigned long long m = __ballot(ch);
    unsigned long long col0 = 0;
    const unsigned long long all = HW >= 64 ? ~0ull : ((1ull << HW) - 1ull);
    for (int y = 0; y < H; ++y) col0 |= 1ull << (y * W);
    const unsigned long long colL = col0 << (W - 1);
    for (int r = 1; r <= R; ++r) {
      const unsigned long long hdil = (m | ((m << 1) & ~col0) | ((m >> 1) & ~colL)) & all;
      m = (hdil | (hdil << W) | (hdil >> W)) & all;
      const unsigned long long ml = m & ((1ull << 48) - 1ull);            // listed: positions 0..47
      const int n = __popcll(ml);
      uint8_t* rec = need + spk_need_off_rec(B, R, r - 1) + (long long)slot * 64;
      if (lane < 48) {
        if ((ml >> lane) & 1ull) rec[__popcll(ml & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        const int last = ml ? 63 - __clzll((long long)ml) : 0;
        if (lane >= n) rec[lane] = (uint8_t)last;
      }
      if (lane == 48) rec[48] = (uint8_t)n;
      if (lane == 49) rec[49] = (uint8_t)(n ? (n + 7) >> 3 : 1);
      if (lane == 50) rec[50] = (uint8_t)((m >> 48) & 1ull);
    }
  }
  // last workgroup to arrive: slots by class, per radius
  __threadfence();
  __syncthreads();
  unsigned* ticket = reinterpret_cast<unsigned*>(need);
  if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x < 64) s_cnt[threadIdx.x >> 3][threadIdx.x & 7] = 0;
  __syncthreads();
  for (int r = 0; r < R; ++r) {
    const uint8_t* recs = need + spk_need_off_rec(B, R, r);
    int* list = reinterpret_cast<int*>(need + spk_need_off_list(B, R, r));
    for (int s = threadIdx.x; s < Bn; s += blockDim.x) {
      int k = (int)__builtin_nontemporal_load(recs + (long long)s * 64 + 49) - 1;
      k = k < 0 ? 0 : (k > 5 ? 5 : k);
      list[k * B + atomicAdd(&s_cnt[r][k], 1)] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < 8 * R) {
    const int r = threadIdx.x >> 3, k = threadIdx.x & 7;
    reinterpret_cast<int*>(need + spk_need_off_cnt(r))[k] = k < 6 ? s_cnt[r][k] : 0;
  }
  if (threadIdx.x == 0) *ticket = 0u;
}

// The noise of one reverse step exactly as psample_kernel / select_active_kernel / select_needed_kernel draw it, written
// out: u[p] (stream 0, counter offset + p*K) and q[p*K + k] (stream 1, counter offset + p*K + k).  A debug / parity entry:
// the oracle is run on the dumped noise and must give the tokens of the Philox-mode sampler (the timed configuration).
__global__ __launch_bounds__(256) void philox_noise_kernel(unsigned long long seed, unsigned long long offset,
                                                           const unsigned long long* __restrict__ philox_state,
                                                           float* __restrict__ u_out, float* __restrict__ q_out,
                                                           long long npos, int K) {
  if (philox_state) { seed = philox_state[0]; offset += philox_state[1]; }
  const long long nq = q_out ? npos * K : 0;
  const long long total = nq > npos ? nq : npos;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    uint32_t r[4];
    if (u_out && i < npos) { philox4x32(seed, offset + (unsigned long long)i * (unsigned long long)K, 0u, r); u_out[i] = u01_open_right(r[0]); }
    if (i < nq) { philox4x32(seed, offset + (unsigned long long)i, 1u, r); q_out[i] = -logf(u01_open_left(r[0])); }
  }
}

}  // namespace

namespace {
// q_sample of the training step (R/snn_model/vq_diffusion.py:61-75): mask = u < t[b] / num_timesteps; x_t = mask ? mask_id : x_0;
// x_0_ignore = mask ? x_0 : -1.  One launch for what the module sequence spends eight on (expand, float, divide, compare, two
// fills, two selects); the uniforms u stay the framework's draw (rand_like), so the RNG stream is the reference's.
__global__ void q_sample_kernel(const float* __restrict__ x0, const long long* __restrict__ t, const float* __restrict__ u,
                                float* __restrict__ x_t, float* __restrict__ x0_ignore, uint8_t* __restrict__ mask, int B, int HW,
                                float num_timesteps, float mask_id) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  const int b = i / HW;
  const bool m = u[i] < (float)t[b] / num_timesteps;
  const float x = x0[i];
  x_t[i] = m ? mask_id : x;
  x0_ignore[i] = m ? x : -1.0f;
  if (mask) mask[i] = m ? 1 : 0;
}
}  // namespace

extern "C" int spk_q_sample(const float* x0, const long long* t, const float* u, float* x_t_out, float* x0_ignore_out,
                            uint8_t* mask_out_or_null, int B, int HW, int num_timesteps, float mask_id, hipStream_t stream) {
  if (!x0 || !t || !u || !x_t_out || !x0_ignore_out || B <= 0 || HW <= 0 || num_timesteps <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(q_sample_kernel, dim3((B * HW + 255) / 256), dim3(256), 0, stream, x0, t, u, x_t_out, x0_ignore_out,
                     mask_out_or_null, B, HW, (float)num_timesteps, mask_id);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_philox_noise(unsigned long long philox_seed, unsigned long long philox_offset,
                                const unsigned long long* philox_state_or_null, float* u_out_or_null, float* q_out_or_null,
                                int B, int HW, int K, hipStream_t stream) {
  if ((!u_out_or_null && !q_out_or_null) || B <= 0 || HW <= 0 || K <= 0) return SPK_ERR_ARG;
  const long long npos = (long long)B * HW;
  const long long total = q_out_or_null ? npos * K : npos;
  long long grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(philox_noise_kernel, dim3((int)grid), dim3(256), 0, stream, philox_seed, philox_offset,
                     philox_state_or_null, u_out_or_null, q_out_or_null, npos, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" long long spk_select_needed_bytes(int B, int R) {
  if (B <= 0 || R <= 0 || R > 8) return -1;
  return spk_need_off_rec(B, R, R);
}

extern "C" int spk_select_needed(const uint8_t* unmasked, int t, const float* u_or_null, unsigned long long philox_seed,
                                 unsigned long long philox_offset, const unsigned long long* philox_state_or_null,
                                 const int* active, const int* n_active, uint8_t* need_out, int B, int H, int W, int R,
                                 int K, hipStream_t stream) {
  if (!unmasked || !active || !n_active || !need_out || t <= 0 || B <= 0 || R <= 0 || R > 8 || K <= 0) return SPK_ERR_ARG;
  if (H != 7 || W != 7) return SPK_ERR_UNSUPPORTED;        // the record format lists positions 0..47 of a 49-position latent
  hipLaunchKernelGGL(select_needed_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, unmasked, t, u_or_null, philox_seed,
                     philox_offset, philox_state_or_null, active, n_active, need_out, B, H, W, R, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_select_active(const uint8_t* unmasked, int t, const float* u_or_null, unsigned long long philox_seed,
                                 unsigned long long philox_offset, const unsigned long long* philox_state_or_null,
                                 int* active_out, int* n_active_out, int B, int HW, int K, hipStream_t stream) {
  if (!unmasked || !active_out || !n_active_out || t <= 0 || B <= 0 || HW <= 0 || K <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(select_active_kernel, dim3((B * 8 + 63) / 64), dim3(64), 0, stream, unmasked, t, u_or_null, philox_seed,
                     philox_offset, philox_state_or_null, active_out, n_active_out, B, HW, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_den_build_input(const float* x_float_or_null, const long long* x_tokens_or_null,
                                   const long long* t_vec_or_null, long long t_scalar, float* out_b2hw, int B, int HW,
                                   const int* active_or_null, const int* n_active_or_null, hipStream_t stream) {
  if ((!x_float_or_null && !x_tokens_or_null) || !out_b2hw || B <= 0 || HW <= 0) return SPK_ERR_ARG;
  if ((active_or_null == nullptr) != (n_active_or_null == nullptr)) return SPK_ERR_ARG;
  int total = B * HW;
  hipLaunchKernelGGL(den_input_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, x_float_or_null,
                     x_tokens_or_null, t_vec_or_null, t_scalar, out_b2hw, active_or_null, n_active_or_null, B, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_psample_step(const float* logits_bkhw, long long* x_t_inout, uint8_t* unmasked_inout, int t,
                                float temp, const float* u_or_null, const float* q_or_null,
                                unsigned long long philox_seed, unsigned long long philox_offset,
                                const unsigned long long* philox_state_or_null, long long* x0_hat_out_or_null, int B,
                                int HW, int K, const int* active_or_null, const int* n_active_or_null,
                                float* next_input_b2hw_or_null, hipStream_t stream) {
  if (!logits_bkhw || !x_t_inout || !unmasked_inout || t <= 0 || !(temp > 0.f) || B <= 0 || HW <= 0 || K <= 0)
    return SPK_ERR_ARG;
  if (next_input_b2hw_or_null && active_or_null) return SPK_ERR_ARG;      // (the active-set form gathers its input by slot)
  if ((active_or_null == nullptr) != (n_active_or_null == nullptr) || (active_or_null && x0_hat_out_or_null))
    return SPK_ERR_ARG;
  if (K > 64 * 32) return SPK_ERR_UNSUPPORTED;
  long long npos = (long long)B * HW;
  int grid = (int)((npos + 3) / 4);
  if (grid > 4096) grid = 4096;
#define SPK_PSAMPLE_LAUNCH(KPL)                                                                                          \
  hipLaunchKernelGGL(psample_kernel<KPL>, dim3(grid), dim3(256), 0, stream, logits_bkhw, x_t_inout, unmasked_inout, t,   \
                     temp, u_or_null, q_or_null, philox_seed, philox_offset, philox_state_or_null, x0_hat_out_or_null,   \
                     active_or_null, n_active_or_null, B, HW, K, next_input_b2hw_or_null, (float)(t - 1))
  if (K <= 256) SPK_PSAMPLE_LAUNCH(4);
  else if (K <= 512) SPK_PSAMPLE_LAUNCH(8);
  else if (K <= 1024) SPK_PSAMPLE_LAUNCH(16);
  else SPK_PSAMPLE_LAUNCH(32);
#undef SPK_PSAMPLE_LAUNCH
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
