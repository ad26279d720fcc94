// The tail of one reverse-diffusion step of the dense sampler as ONE launch per image:
//
//   conv6 + mean over T   R/snn_model/vq_diffusion.py:185-187,205-206   (time-collapsed on spike counts, as den_mfma.hip)
//   p_sample              R/snn_model/vq_diffusion.py:113-124,134-140   (the arithmetic of psample.hip, same noise draws)
//   conv1 + BN + LIF of the NEXT step's input cat(x_t, t - 1)   :161-165,195-201   (the arithmetic of tinv_lif_kernel)
//
// One workgroup = one image, eight waves; wave w owns the 16-channel groups w, w + 8, ... of the logits (the reference's default
// codebook of 128 classes: one group per wave; round 6: any --codebook_size up to 512, R/main.py:58 -- conv6's output channels are
// zero-padded to a multiple of 16 in the packed weights, classes >= K are masked in the sampling).  The 64 (padded) rows of an image are
// two 32-row MFMA tiles; wave w multiplies both by the four int8 digit planes of its channel group (v_mfma_i32_32x32x32_i8,
// exact int32 sums, the packed weights of spk_den_pack_weight_i8 streamed from L2 eight taps ahead), the image's count records
// (15 - 20 KB) sit in LDS for the whole launch.  The logits of the image meet in LDS (never in HBM unless the caller asks for
// them), every wave then samples an eighth of the positions that change at this step, and once the image's tokens are final the
// workgroup evaluates the first denoiser layer for the next step (64 channels x 49 positions: a table look-up per neuron,
// spk_common.h) and writes its S32 spikes and spike counts.  Replaces three launches of the dense reverse step
// (conv3x3_counts_mfma_shared_kernel 33 us + psample_kernel 9 us + tinv_lif_kernel 13 us at B = 256) and the logits round trip.
//
// Bit-for-bit the results of those three kernels: same digit planes and fp64 recombination (one rounding), same softmax /
// exponential-race arithmetic and Philox counters, same fp64 dot product and look-up table.
#include "den_common.h"
#include "psample_common.h"
#include "../../include/spkdiff.h"
#include <math.h>

namespace {

constexpr int TCK = 32;                           // channels per K chunk
constexpr int TW_CHUNK = 9 * 2 * 32 * TCK;        // 18432 B of packed weights per (channel group, chunk)
constexpr int TNCH = 10;                          // 8 chunks of conv5 counts + 2 of conv1 counts (256 + 64 channels)
constexpr int TK_MAX = 512;                       // classes = conv6 output channels (KG = groups per wave = ceil(K / 128) <= 4)
#ifndef SPK_TAIL_PF
#define SPK_TAIL_PF 8                             // weight tiles are requested this many taps ahead of their MFMAs
#endif
#ifndef SPK_TAIL_ROT
#define SPK_TAIL_ROT 1
#endif
#ifndef SPK_TAIL_DBG
#define SPK_TAIL_DBG 0                            // timing experiments only (wrong results): 1 no K loop, 2 no token update, 4 no conv1
#endif

struct TailArgs {
  const uint8_t* c5; const uint8_t* c1;           // spike counts u8 [B][8][HW][32], [B][2][HW][32]
  const int8_t* wq; const double* scale; const double* bias;
  float* logits_out;                              // optional fp32 [B][128][HW]
  long long* x_t; uint8_t* unmasked; int t; float temp;
  const float* u_in; const float* q_in;
  unsigned long long seed, offset; const unsigned long long* philox_state;
  const float* w1; const float* b1; const float* bn1_a; const float* bn1_b;   // next step's conv1 (packed [9][2][64]); null: none
  uint8_t* x1_out; uint8_t* cnt1_out;             // S32 spikes [B][2][HW][16][16 B], counts u8 [B][2][HW][32]
  float t_next;
  int B, T;
  int K, ng;                                      // classes; 16-channel groups of the padded logits (ceil(K / 16))
  // active-set form (the untouched-image elimination, spk_select_active): workgroup = SLOT s of the list; counts / logits are indexed by
  // slot (what the active-set denoiser launches produced), tokens / unmasked / noise by IMAGE active[s]: the draws of the dense form
  const int* active; const int* n_active;
};

// KG = channel groups per wave (1: K <= 128, the reference's default; 2 .. 4: K <= 256 / 384 / 512)
template <int H, int W, int KG>
__global__ __launch_bounds__(512, 1) void step_tail_kernel(TailArgs a) {
  constexpr int HW = H * W;
  static_assert(HW <= 64, "an image is at most two 32-row tiles");
  constexpr int AV = HW * 2 + 1;                  // 16-byte vectors of a chunk's count records + one zero vector
  constexpr int TLP = 128 * KG + 4;               // LDS pitch of a logits row
  constexpr int NJ = 2 * KG;                      // classes per lane in the sampling (64 * NJ >= padded K)
  __shared__ v4i s_a[TNCH * AV];
  extern __shared__ __attribute__((aligned(16))) float s_logit_dyn[];   // [64][TLP]
  float (*s_logit)[TLP] = reinterpret_cast<float (*)[TLP]>(s_logit_dyn);
  __shared__ float s_tok[64];
  __shared__ int s_chg[64];
  __shared__ float s_th[16];
  __shared__ unsigned s_pat[18];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;                       // slot: index of the count records and of the logits
  int bi = b;                                     // image: index of tokens, unmasked and noise
  if (a.active) {
    const int n = *a.n_active < a.B ? *a.n_active : a.B;
    if (b >= n) return;                           // (uniform over the workgroup)
    bi = a.active[b];
  }
  const int K = a.K, ng = a.ng;
  const int row = lane & 31, half = lane >> 5;

  // ---- this wave's weight stream: tile (c, tap, j) = 1 KiB, 16 bytes per lane; requested D taps ahead of their MFMAs
  const int boff = (lane & 31) * TCK + 16 * (half ^ ((lane >> 4) & 1));
  // (the first group's stream starts here, ahead of the count records; a wave without a group -- K < 128 -- streams group 0's
  //  first tiles and drops them)
  const int8_t* wg = a.wq + (long long)(wave < ng ? wave : 0) * TNCH * TW_CHUNK + boff;
  // every workgroup reads the same 1.4 MB of packed weights: workgroups of one XCD (blocks k, k + 8, ...) start at different
  // chunks so that they do not all ask the L2 for the same lines at the same time (exact integer sums: any order)
  const int rot = SPK_TAIL_ROT ? (int)((blockIdx.x >> 3) % TNCH) : 0;
  auto chunk_of = [&](int i) -> int { const int c = i / 9 + rot; return c >= TNCH ? c - TNCH : c; };
  constexpr int NIT = TNCH * 9, D = SPK_TAIL_PF;
  v4i bq[D][2];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    bq[i][0] = *reinterpret_cast<const v4i*>(wg + chunk_of(i) * TW_CHUNK + ((i % 9) * 2 + 0) * 32 * TCK);
    bq[i][1] = *reinterpret_cast<const v4i*>(wg + chunk_of(i) * TW_CHUNK + ((i % 9) * 2 + 1) * 32 * TCK);
  }

  // ---- the image's count records -> LDS
  for (int e = tid; e < TNCH * HW * 2; e += 512) {
    const int c = e / (HW * 2), r = e - c * (HW * 2);
    const uint8_t* src = c < 8 ? a.c5 + (((long long)b * 8 + c) * HW) * TCK + r * 16
                               : a.c1 + (((long long)b * 2 + (c - 8)) * HW) * TCK + r * 16;
    s_a[c * AV + r] = *reinterpret_cast<const v4i*>(src);
  }
  if (tid < TNCH) s_a[tid * AV + HW * 2] = (v4i){0, 0, 0, 0};
  // ---- which positions change at this step, and the tokens as they are: one thread per position, up front (fetched row by row
  //      in the sampling loop these were seven dependent memory round trips per wave)
  unsigned long long seed = a.seed, offset = a.offset;
  if (a.philox_state) { seed = a.philox_state[0]; offset += a.philox_state[1]; }
  const float inv_t = 1.0f / (float)a.t;
  if (tid < HW) {
    const long long pi = (long long)bi * HW + tid;
    const uint8_t um = a.unmasked[pi];
    const long long tk = a.x_t[pi];
    float u;
    if (a.u_in) u = a.u_in[pi];
    else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)pi * (unsigned long long)K, 0u, r); u = u01_open_right(r[0]); }
    s_chg[tid] = ((u < inv_t) && !um) ? 1 : 0;
    s_tok[tid] = (float)tk;
  }
  // ---- per-channel constants of the next step's first layer, requested before the K loop
  const int col_e = lane & 31;
  float wreg[18];
  float al1 = 0.f, be1 = 0.f;
  double b01 = 0.0;
  if (a.x1_out) {
#pragma unroll
    for (int i = 0; i < 18; ++i) wreg[i] = a.w1[i * 64 + lane];        // packed [k * k][Cin = 2][Cout = 64]
    al1 = a.bn1_a[lane]; be1 = a.bn1_b[lane];
    b01 = a.b1 ? (double)a.b1[lane] : 0.0;
  } else {
#pragma unroll
    for (int i = 0; i < 18; ++i) wreg[i] = 0.f;
  }
  {
    constexpr unsigned thb[16] = SPK_LIF_CONST_TH_BITS, pat[18] = SPK_LIF_CONST_PATTERNS;
    if (tid < 16) s_th[tid] = __uint_as_float(thb[tid]);
    if (tid < 18) s_pat[tid] = pat[tid];
  }
  __syncthreads();

  // ---- conv6 on the counts: rows = positions (two tiles), columns = 16 channels x 4 digit planes of this wave's group
  int a_base[2];
  int y_[2], x_[2];
  bool rv[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int p = rt * 32 + row;
    rv[rt] = p < HW;
    y_[rt] = p / W; x_[rt] = p % W;
    a_base[rt] = p * 2 + half;
  }
#pragma unroll 1
  for (int gi = 0; gi < KG; ++gi) {
    const int g = wave + 8 * gi;                    // channel group of this pass (wave-uniform)
    if (g >= ng) break;
    const int co_e = g * 16 + (col_e & 15);
    const double sc_e = a.scale[co_e], bT_e = a.bias[co_e] * (double)a.T;
    // (KG > 1: everything the 90 unrolled steps derive from these is invariant over the groups -- hoisted out of this loop it was 120
    //  spilled registers; opaque copies keep the per-step address arithmetic where it is used)
    int rot_g = rot, ab_g[2] = {a_base[0], a_base[1]}, y_g[2] = {y_[0], y_[1]}, x_g[2] = {x_[0], x_[1]};
    if constexpr (KG > 1) asm volatile("" : "+s"(rot_g), "+v"(ab_g[0]), "+v"(ab_g[1]), "+v"(y_g[0]), "+v"(y_g[1]), "+v"(x_g[0]), "+v"(x_g[1]));
    auto chunk_g = [&](int i) -> int { const int c = i / 9 + rot_g; return c >= TNCH ? c - TNCH : c; };
    if (gi > 0) {                                   // (the first group's stream was started ahead of the count records)
      wg = a.wq + (long long)g * TNCH * TW_CHUNK + boff;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        bq[i][0] = *reinterpret_cast<const v4i*>(wg + chunk_g(i) * TW_CHUNK + ((i % 9) * 2 + 0) * 32 * TCK);
        bq[i][1] = *reinterpret_cast<const v4i*>(wg + chunk_g(i) * TW_CHUNK + ((i % 9) * 2 + 1) * 32 * TCK);
      }
    }
    v16i acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][j][r] = 0;
#pragma unroll
    for (int i = 0; i < ((SPK_TAIL_DBG & 1) ? D : NIT); ++i) {
      const int c = chunk_g(i), tap = i % 9;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const v4i b0 = bq[i % D][0], b1 = bq[i % D][1];
      if (i + D < NIT) {
        const int c2 = chunk_g(i + D), tap2 = (i + D) % 9;
        bq[i % D][0] = *reinterpret_cast<const v4i*>(wg + c2 * TW_CHUNK + (tap2 * 2 + 0) * 32 * TCK);
        bq[i % D][1] = *reinterpret_cast<const v4i*>(wg + c2 * TW_CHUNK + (tap2 * 2 + 1) * 32 * TCK);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int yy = y_g[rt] + dy, xx = x_g[rt] + dx;
        const bool ok = rv[rt] && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const v4i av = s_a[c * AV + (ok ? ab_g[rt] + (dy * W + dx) * 2 : HW * 2)];
        acc[rt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b0, acc[rt][0], 0, 0, 0);
        acc[rt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b1, acc[rt][1], 0, 0, 0);
      }
    }

    // ---- digit recombination (fp64, one rounding), mean over T, logits -> LDS (and to memory on request)
    {
      const int odd = col_e >> 4;
      const int co = co_e;
      const double sc = sc_e, bT = bT_e;
      const float invT = 1.0f / (float)a.T;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const v2u p01 = __builtin_amdgcn_permlane16_swap((unsigned)acc[rt][0][r], (unsigned)acc[rt][0][r + 8], false, false);
          const v2u p23 = __builtin_amdgcn_permlane16_swap((unsigned)acc[rt][1][r], (unsigned)acc[rt][1][r + 8], false, false);
          const long long hi = (long long)(int)p01[0] * 256 + (int)p01[1], lo = (long long)(int)p23[0] * 256 + (int)p23[1];
          const double s = fma((double)hi, 65536.0, (double)lo);
          const float xsum = (float)fma(s, sc, bT);
          const int rr = r + 8 * odd;
          const int orow = (rr & 3) + 8 * (rr >> 2) + 4 * half;
          const int op = rt * 32 + orow;
          const float lg = xsum * invT;
          s_logit[op][co] = lg;
          if (a.logits_out && op < HW && co < K) a.logits_out[((long long)b * K + co) * HW + op] = lg;
        }
      }
    }
  }
  __syncthreads();

  // ---- p_sample: wave w takes positions w, w + 8, ...; only positions that change at this step consume a sample (:140)
  for (int p = wave; p < ((SPK_TAIL_DBG & 2) ? 0 : HW); p += 8) {
    const long long pi = (long long)bi * HW + p;
    if (!s_chg[p]) continue;                                            // (wave-uniform; s_tok[p] holds the token it keeps)
    // (classes k >= K -- the zero-padded output channels of conv6 and the lanes beyond them -- are masked exactly as
    //  psample_kernel masks them: -inf logits, zero terms)
    float l[NJ], e[NJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = lane + 64 * j;
      l[j] = k < K ? s_logit[p][k] / a.temp : -INFINITY;
      mx = fmaxf(mx, l[j]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) se += (lane + 64 * j < K) ? expf(l[j] - mx) : 0.f;
    se = wave_sum(se);
    const float lse = mx + logf(se);
    float mx2 = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { l[j] = l[j] - lse; mx2 = fmaxf(mx2, l[j]); }
    mx2 = wave_max(mx2);
    float se2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { e[j] = (lane + 64 * j < K) ? expf(l[j] - mx2) : 0.f; se2 += e[j]; }
    se2 = wave_sum(se2);
    float best = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int k = lane + 64 * j;
      if (k < K) {
        float q;
        if (a.q_in) q = a.q_in[pi * K + k];
        else { uint32_t r[4]; philox4x32(seed, offset + (unsigned long long)(pi * K + k), 1u, r); q = -logf(u01_open_left(r[0])); }
        const float ratio = (e[j] / se2) / q;
        if (ratio > best) { best = ratio; besti = k; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(besti, off);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) {
      a.unmasked[pi] = 1;
      a.x_t[pi] = (long long)besti;
      s_tok[p] = (float)besti;
    }
  }
  if (!a.x1_out || (SPK_TAIL_DBG & 4)) return;                         // (uniform: the last reverse step has no successor)
  __syncthreads();

  // ---- the next step's first layer: conv1(cat(x_t, t - 1)) + BN + LIF from the reset state -> S32 spikes + spike counts
  {
    const int co = lane, pl = wave;                                    // 64 output channels, eight positions per pass
    const float al = al1, be = be1;
    const double b0 = b01;
    for (int p0 = 0; p0 < HW; p0 += 8) {
      const int opr = p0 + pl;
      const bool ok = opr < HW;
      const int op = ok ? opr : HW - 1;
      const int oy = op / W, ox = op - oy * W;
      double accd = b0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox - 1 + kx;
          const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
          const float tk = s_tok[in ? iy * W + ix : 0];
          const float x0 = in ? tk : 0.0f, x1 = in ? a.t_next : 0.0f;
          accd = fma((double)x0, (double)wreg[(ky * 3 + kx) * 2 + 0], accd);
          accd = fma((double)x1, (double)wreg[(ky * 3 + kx) * 2 + 1], accd);
        }
      }
      const float y0 = fmaf((float)accd, al, be);
      const unsigned mybits = spk_lif_const_input_bits16(y0, s_th, s_pat);
      if (ok) a.cnt1_out[(((long long)b * 2 + (co >> 5)) * HW + op) * 32 + (co & 31)] = (uint8_t)__popc(mybits);
      const unsigned bitsv = spk_transpose16_rows(mybits, lane);
      const int tl = lane & 15, co16 = co & ~15;
      if (!ok) continue;
      unsigned lo8 = bitsv & 0xffu, hi8 = (bitsv >> 8) & 0xffu;
      auto spread8 = [](unsigned x) -> unsigned {                       // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
        x = (x | (x << 12)) & 0x000f000fu;
        x = (x | (x << 6)) & 0x03030303u;
        x = (x | (x << 3)) & 0x11111111u;
        return x << 1;
      };
      uint2 o;
      o.x = spread8(lo8);
      o.y = spread8(hi8);
      uint8_t* dst = a.x1_out + ((((long long)b * 2 + (co16 >> 5)) * HW + op) * 16 + tl) * 16 + ((co16 & 31) >> 1);
      *reinterpret_cast<uint2*>(dst) = o;
    }
  }
}

}  // namespace

extern "C" int spk_den_step_tail(const uint8_t* cnt5, int nch5, const uint8_t* cnt1, int nch1, const int8_t* wq,
                                 const double* scale, const double* bias_d, float* logits_out_or_null, long long* x_t_inout,
                                 uint8_t* unmasked_inout, int t, float temp, const float* u_or_null, const float* q_or_null,
                                 unsigned long long philox_seed, unsigned long long philox_offset,
                                 const unsigned long long* philox_state_or_null, const float* conv1_w_packed_or_null,
                                 const float* conv1_bias_or_null, const float* bn1_a, const float* bn1_b,
                                 uint8_t* x1_s32_out_or_null, uint8_t* cnt1_out_or_null, int T, int B, int H, int W, int K,
                                 const int* active_or_null, const int* n_active_or_null, hipStream_t stream) {
  if (!cnt5 || !cnt1 || !wq || !scale || !bias_d || !x_t_inout || !unmasked_inout || t <= 0 || !(temp > 0.f) || B <= 0 || T <= 0)
    return SPK_ERR_ARG;
  if ((x1_s32_out_or_null == nullptr) != (cnt1_out_or_null == nullptr)) return SPK_ERR_ARG;
  if ((active_or_null == nullptr) != (n_active_or_null == nullptr)) return SPK_ERR_ARG;
  // (the next step's first layer belongs to the NEXT step's active set, which only spk_select_active after this update knows)
  if (active_or_null && x1_s32_out_or_null) return SPK_ERR_UNSUPPORTED;
  if (x1_s32_out_or_null && (!conv1_w_packed_or_null || !bn1_a || !bn1_b)) return SPK_ERR_ARG;
  if (nch5 != 8 || nch1 != 2 || K < 1 || K > TK_MAX || T > 127 || !((H == 7 && W == 7) || (H == 8 && W == 8))) return SPK_ERR_UNSUPPORTED;
  // the fused first layer writes sixteen 16-byte step records per position and scans with the module-default LIF constants
  // (spk_lif_const_input_bits16): any other step count would write outside x1_s32_out (T < 16) or give wrong spikes (T > 16)
  if (x1_s32_out_or_null && T != 16) return SPK_ERR_UNSUPPORTED;
  TailArgs a;
  a.c5 = cnt5; a.c1 = cnt1; a.wq = wq; a.scale = scale; a.bias = bias_d; a.logits_out = logits_out_or_null;
  a.x_t = x_t_inout; a.unmasked = unmasked_inout; a.t = t; a.temp = temp; a.u_in = u_or_null; a.q_in = q_or_null;
  a.seed = philox_seed; a.offset = philox_offset; a.philox_state = philox_state_or_null;
  a.w1 = conv1_w_packed_or_null; a.b1 = conv1_bias_or_null; a.bn1_a = bn1_a; a.bn1_b = bn1_b;
  a.x1_out = x1_s32_out_or_null; a.cnt1_out = cnt1_out_or_null; a.t_next = (float)(t - 1);
  a.B = B; a.T = T;
  a.active = active_or_null; a.n_active = n_active_or_null;
  a.K = K; a.ng = (K + 15) / 16;                    // wq / scale / bias_d hold ng * 16 channels (zero weights beyond K)
  const int kg = (a.ng + 7) / 8;
#define SPK_TAIL_LAUNCH(H_, W_, KG_)                                                                                          \
  hipLaunchKernelGGL((step_tail_kernel<H_, W_, KG_>), dim3(B), dim3(512), (size_t)64 * (128 * KG_ + 4) * sizeof(float), stream, a)
  if (H == 7) {
    if (kg == 1) SPK_TAIL_LAUNCH(7, 7, 1); else if (kg == 2) SPK_TAIL_LAUNCH(7, 7, 2);
    else if (kg == 3) SPK_TAIL_LAUNCH(7, 7, 3); else SPK_TAIL_LAUNCH(7, 7, 4);
  } else {
    if (kg == 1) SPK_TAIL_LAUNCH(8, 8, 1); else if (kg == 2) SPK_TAIL_LAUNCH(8, 8, 2);
    else if (kg == 3) SPK_TAIL_LAUNCH(8, 8, 3); else SPK_TAIL_LAUNCH(8, 8, 4);
  }
#undef SPK_TAIL_LAUNCH
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
