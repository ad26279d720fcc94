// Generic (Conv2d | ConvTranspose2d) + BN + LIF over binary spike tensors on the matrix cores, for the spiking
// VQ-VAE layers whose input is spikes: Encoder conv2/conv3, Decoder convT1/convT2 and the final convT + membrane
// read-out (R/snn_model/vae_model.py:115-124,139-155,186).  Same exact int8 digit-plane arithmetic and the same
// accumulator epilogue (permlane recombination, paired LIF scan, bit-matrix transpose, 16-byte spike stores) as the
// denoiser kernel (den_mfma.hip); what differs is the data movement: these layers have small K (64..576) and large
// spatial extents (up to 28x28), so there is nothing to keep resident -- every wave owns one task
//     4 output positions of one sub-pixel class  x  16 time steps  x  16 output channels (x 4 digit planes)
// gathers its A fragments (16 B per lane: 16 channels of one input position and time step) and its weight fragments
// straight from L2/HBM, issues 2 x 2 MFMAs per K step, and runs the epilogue.  No LDS, no barriers; latency is hidden
// by occupancy (~100 registers per wave).  Transposed stride-2 convolutions are handled as their 4 sub-pixel classes:
// all 4 positions of a task share (oy % stride, ox % stride), so the set of contributing taps is wave-uniform.
//
// Layouts: input / output spikes plain PTC u8 [B][H*W][T=16][C]; packed weights [ceil(Cout/16)][ceil(Cin/32)][k*k]
// [2 column tiles][32 cols][32 k] int8 (spk_pack_conv_weight_i8; Cin / Cout are zero-padded to 32 / 16).
#include "spk_common.h"
#include "../../include/spkdiff.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int T16 = 16;

// (the 16x16 bit transpose of den_common.h; this file keeps to spk_common.h, so it carries the same few lines)
__device__ __forceinline__ unsigned transpose16_rows_g(unsigned x, int lane) {
  unsigned y;
  // (opaque copy of the lane id: the three per-lane constants of a round are recomputed here -- three vector instructions --
  //  instead of being hoisted out of the caller's loops, where eight of them stayed live across the K loop: 256 registers + spills)
  int ln = lane;
  asm volatile("" : "+v"(ln));
#define SPK_TR16_ROUND_G(S, LOW)                                                                     \
  do {                                                                                               \
    const unsigned sh = (unsigned)ln & (unsigned)(S);            /* 0 or S */                         \
    const unsigned keep = (unsigned)(LOW) << sh;                                                     \
    const unsigned amt = (32u - (unsigned)(S)) + 2u * sh;        /* 32 - S, or 32 + S = S (mod 32) */ \
    const unsigned yr = __builtin_amdgcn_alignbit(y, y, amt);                                        \
    x = (x & keep) | (yr & ~keep);                                                                   \
  } while (0)
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true), 0x141, 0xF, 0xF, true);
  SPK_TR16_ROUND_G(8, 0x00FFu);
  y = __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  SPK_TR16_ROUND_G(4, 0x0F0Fu);
  y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
  SPK_TR16_ROUND_G(2, 0x3333u);
  y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
  SPK_TR16_ROUND_G(1, 0x5555u);
#undef SPK_TR16_ROUND_G
  return x;
}

__device__ __forceinline__ unsigned spread8_g(unsigned x) {        // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
  x = (x | (x << 12)) & 0x000f000fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}

struct GArgs {
  uint8_t* out_s32;       // MODE_LIF, optional: nibble-packed "S32" spikes [B][Cout/32][Ho*Wo][16][16 B] (den_mfma_fp6v2.hip)
  const uint8_t* in;      // PTC [B][H*W][16][Cin]
  const int8_t* wq; const double* scale; const double* bias; const float* bn_a; const float* bn_b;
  float* v_io;            // [B][Cout][Ho*Wo] or null
  uint8_t* out_ptc;       // MODE_LIF: PTC [B][Ho*Wo][16][Cout]
  const float* coef;      // MODE_MEMOUT: [16]
  float* out_f32;         // MODE_MEMOUT: [B][Cout][Ho*Wo]; MODE_LIF (optional): [B][Ho*Wo][Cout] = sum_t coef[t] * spike[t]
  uint8_t* out_u8;        // MODE_MEMOUT, optional
  int apply_tanh;
  int B, H, W, Cin, Cout, Ho, Wo, k, stride, pad, transposed;
};

// sub-pixel classes: transposed -> stride*stride classes over the INPUT grid (class-local grid Hc x Wc);
// plain conv -> one class over the output grid.
template <int MODE>
__global__ __launch_bounds__(256) void conv_mfma_gather_kernel(GArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ncls = a.transposed ? a.stride * a.stride : 1;
  // class-local output grid
  const int G = (a.Cout + 15) >> 4;
  const int nchunks = (a.Cin + 31) >> 5;
  const int KK = a.k * a.k;
  const long long task = (long long)blockIdx.x * 4 + wave_s;
  // task -> (b, class, position group of 4, channel group): channel group fastest (waves of a block share A rows)
  const int g = (int)(task % G);
  long long r1 = task / G;
  // positions per class
  int cls, Hc, Wc, py = 0, px = 0;
  long long pg;
  {
    // all classes have ceil-divided grids; enumerate (b, cls, pg) with per-class group counts computed on the fly
    const int Hc0 = a.transposed ? (a.Ho + a.stride - 1) / a.stride : a.Ho;
    const int Wc0 = a.transposed ? (a.Wo + a.stride - 1) / a.stride : a.Wo;
    const long long groups_per_cls = ((long long)Hc0 * Wc0 + 3) / 4;      // upper bound shared by all classes
    pg = r1 % groups_per_cls; r1 /= groups_per_cls;
    cls = (int)(r1 % ncls); r1 /= ncls;
    if (a.transposed) { py = cls / a.stride; px = cls % a.stride; }
    Hc = a.transposed ? (a.Ho - py + a.stride - 1) / a.stride : a.Ho;
    Wc = a.transposed ? (a.Wo - px + a.stride - 1) / a.stride : a.Wo;
  }
  const int b = (int)r1;
  if (b >= a.B) return;
  const int npos_c = Hc * Wc;
  if (pg * 4 >= npos_c) return;                                        // wave-uniform

  // this lane's A rows: tile i (0,1), position h, time t
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  int oy[2], ox[2];
  bool pvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (int)pg * 4 + 2 * i + hsel;
    pvalid[i] = q < npos_c;
    const int qy = pvalid[i] ? q / Wc : 0, qx = pvalid[i] ? q % Wc : 0;
    oy[i] = a.transposed ? qy * a.stride + py : qy;
    ox[i] = a.transposed ? qx * a.stride + px : qx;
  }
  const int boff = (lane & 31) * 32 + 16 * (half ^ ((lane >> 4) & 1));
  const int HWi = a.H * a.W;
  const uint8_t* inb = a.in + (long long)b * HWi * T16 * a.Cin;

  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0; acc[i][1][r] = 0; }

  for (int ky = 0; ky < a.k; ++ky) {
    // wave-uniform class test (all 4 positions share oy % stride): does this kernel row contribute at all?
    if (a.transposed && ((py + a.pad - ky) % a.stride + a.stride) % a.stride != 0) continue;
    for (int kx = 0; kx < a.k; ++kx) {
      if (a.transposed && ((px + a.pad - kx) % a.stride + a.stride) % a.stride != 0) continue;
      const int tap = ky * a.k + kx;
      // input position of each tile's row for this tap
      long long ioff[2];
      bool ok[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int iy, ix;
        if (a.transposed) {
          const int ty = oy[i] + a.pad - ky, tx = ox[i] + a.pad - kx;
          iy = ty / a.stride; ix = tx / a.stride;
          ok[i] = pvalid[i] && ty >= 0 && tx >= 0 && iy < a.H && ix < a.W;
        } else {
          iy = oy[i] * a.stride - a.pad + ky; ix = ox[i] * a.stride - a.pad + kx;
          ok[i] = pvalid[i] && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
        }
        ioff[i] = ((long long)(iy * a.W + ix) * T16 + tt) * a.Cin + 16 * half;
      }
      for (int c = 0; c < nchunks; ++c) {
        const bool chan_ok = c * 32 + 16 * half < a.Cin;              // Cin = 16: the upper k-half is zero padding
        v4i av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool use = ok[i] && chan_ok;
          const v4i ld = *reinterpret_cast<const v4i*>(inb + (use ? ioff[i] + c * 32 : 0));    // see gather2_body
          av[i] = use ? ld : (v4i){0, 0, 0, 0};
        }
        const int8_t* wp = a.wq + ((((long long)g * nchunks + c) * KK + tap) * 2) * 1024 + boff;
        const v4i b0 = *reinterpret_cast<const v4i*>(wp);
        const v4i b1 = *reinterpret_cast<const v4i*>(wp + 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], b0, acc[i][0], 0, 0, 0);
          acc[i][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[i], b1, acc[i][1], 0, 0, 0);
        }
      }
    }
  }

  // ---------------- epilogue (see den_mfma.hip for the scheme) -------------------------------------------------
  const int col = lane & 31, ch = col & 15, odd = col >> 4;
  const int co = g * 16 + ch;
  const bool co_ok = co < a.Cout;
  const double sc = a.scale[g * 16 + ch], bi = a.bias[g * 16 + ch];     // padded to 16 per group by the packer
  // Pairwise exchange (den_mfma_fp6.hip): v_permlane16_swap(acc[0][ct][r], acc[1][ct][r]) leaves the even lane with both
  // digits of column tile ct of row tile 0 and the odd lane with those of row tile 1 -- 32 swaps give every lane all four
  // digits of ONE neuron for all 16 steps; both lane parities recombine and scan their own tile.
  float xs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const v2u p01 = __builtin_amdgcn_permlane16_swap((unsigned)acc[0][0][r], (unsigned)acc[1][0][r], false, false);
    const v2u p23 = __builtin_amdgcn_permlane16_swap((unsigned)acc[0][1][r], (unsigned)acc[1][1][r], false, false);
    const int hi = (int)p01[0] * 256 + (int)p01[1], lo = (int)p23[0] * 256 + (int)p23[1];
    const double s = fma((double)hi, 65536.0, (double)lo);
    xs[r] = (float)fma(s, sc, bi);
  }
  // even lanes own tile 0, odd lanes tile 1; accumulator lane-half = position within the tile
  const int q = (int)pg * 4 + 2 * odd + half;
  const bool pos_ok = q < npos_c;
  const int qy = pos_ok ? q / Wc : 0, qx = pos_ok ? q % Wc : 0;
  const int opos = (a.transposed ? qy * a.stride + py : qy) * a.Wo + (a.transposed ? qx * a.stride + px : qx);
  const long long HWo = (long long)a.Ho * a.Wo;
  if (MODE == SPK_MODE_LIF) {
    const float bn_a = co_ok ? a.bn_a[co] : 0.f, bn_b = co_ok ? a.bn_b[co] : 0.f;
    const long long vidx = ((long long)b * a.Cout + (co_ok ? co : 0)) * HWo + opos;
    float v = (a.v_io && pos_ok && co_ok) ? a.v_io[vidx] : 0.f;
    unsigned mybits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xv = xs[r];
      const bool s = spk_lif_step_default(v, fmaf(xv, bn_a, bn_b)) && pos_ok && co_ok;
      mybits |= s ? (1u << r) : 0u;
    }
    if (a.v_io && pos_ok && co_ok) a.v_io[vidx] = v;
    if (a.out_f32 && pos_ok && co_ok) {                            // time-collapsed spikes for a linear read-out layer
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) m = m + ((mybits >> r) & 1u ? a.coef[r] : 0.f);
      a.out_f32[((long long)b * HWo + opos) * a.Cout + co] = m;
    }
    const unsigned bitsv = transpose16_rows_g(mybits, lane);      // lane t of each 16-lane row: 16 channel bits of step t
    if (pos_ok && a.out_s32) {                                     // this task's 16 channels = half of a 32-channel record
      uint2 o2;
      o2.x = spread8_g(bitsv & 0xffu);
      o2.y = spread8_g((bitsv >> 8) & 0xffu);
      *reinterpret_cast<uint2*>(a.out_s32 + ((((long long)b * (a.Cout >> 5) + (g >> 1)) * HWo + opos) * T16 + (lane & 15)) * 16 +
                                8 * (g & 1)) = o2;
    }
    if (pos_ok && a.out_ptc) {
      uint4 o;
      o.x = ((bitsv & 0xfu) * 0x00204081u) & 0x01010101u;
      o.y = (((bitsv >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
      o.z = (((bitsv >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
      o.w = (((bitsv >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
      uint8_t* dst = a.out_ptc + (((long long)b * HWo + opos) * T16 + (lane & 15)) * a.Cout + g * 16;
      if (g * 16 + 16 <= a.Cout) {
        *reinterpret_cast<uint4*>(dst) = o;
      } else {                                                     // ragged last channel group
        const uint8_t* ob = reinterpret_cast<const uint8_t*>(&o);
        for (int c2 = 0; g * 16 + c2 < a.Cout; ++c2) dst[c2] = ob[c2];
      }
    }
  } else {  // SPK_MODE_MEMOUT: sum_t x[t] * coef[t]  (+ tanh, + uint8), R/snn_model/snn_layers.py:36-41, R/main.py:399-401
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = m + xs[r] * a.coef[r];
    if (pos_ok && co_ok) {
      const float pv = a.apply_tanh ? tanhf(m) : m;
      const long long oidx = ((long long)b * a.Cout + co) * HWo + opos;
      if (a.out_f32) a.out_f32[oidx] = pv;
      if (a.out_u8) a.out_u8[oidx] = (uint8_t)(fminf(fmaxf(pv + 0.5f, 0.0f), 1.0f) * 255.0f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ specialised form
// Same task, geometry known at compile time (KS x KS taps, stride S, transposed or not, NCH 32-channel K chunks,
// pad = KS/2, sub-pixel class PY,PX): the valid (tap, chunk) steps are a compile-time list, every A and B fragment
// of a task is requested before the first MFMA waits (up to 18 + 18 loads of 16 B per lane in flight per wave
// instead of one dependent L2 round trip per K step), and the task decode has no integer division by runtime values
// other than one per wave.  grid.y = image x class, grid.x*4 + wave = (position group, channel group).
template <int MODE, int KS, int S, bool TR, int NCH, int PY, int PX>
__device__ __forceinline__ void gather2_body(const GArgs& a, int b, int g, int pg, int lane) {
  constexpr int PAD = KS / 2;
  constexpr int KK = KS * KS;
  const int Hc = TR ? (a.Ho - PY + S - 1) / S : a.Ho;
  const int Wc = TR ? (a.Wo - PX + S - 1) / S : a.Wo;
  const int npos_c = Hc * Wc;
  if (pg * 4 >= npos_c) return;
  const float inv_wc = 1.0f / (float)Wc;
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  int oy[2], ox[2];
  bool pvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = pg * 4 + 2 * i + hsel;
    pvalid[i] = q < npos_c;
    const int qy = (int)(((float)q + 0.5f) * inv_wc), qx = q - qy * Wc;      // q < 2^20: exact
    oy[i] = TR ? qy * S + PY : qy;
    ox[i] = TR ? qx * S + PX : qx;
  }
  const int boff = (lane & 31) * 32 + 16 * (half ^ ((lane >> 4) & 1));
  const uint8_t* inb = a.in + (long long)b * a.H * a.W * T16 * a.Cin;
  const int8_t* wg = a.wq + (long long)g * NCH * KK * 2048 + boff;
  const bool chan_hi_ok = 16 * half < a.Cin;                                  // Cin = 16: upper k-half is zero padding

  v4i av[KK * NCH][2], bv[KK * NCH][2];
  // ---- request everything
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      constexpr int dummy = 0; (void)dummy;
      const bool tap_ok = !TR || ((((PY + PAD - ky) % S + S) % S == 0) && (((PX + PAD - kx) % S + S) % S == 0));
      if (tap_ok) {
        long long ioff[2];
        bool ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int iy, ix;
          if (TR) {
            const int ty = oy[i] + PAD - ky, tx = ox[i] + PAD - kx;
            iy = ty / S; ix = tx / S;
            ok[i] = pvalid[i] && ty >= 0 && tx >= 0 && iy < a.H && ix < a.W;
          } else {
            iy = oy[i] * S - PAD + ky; ix = ox[i] * S - PAD + kx;
            ok[i] = pvalid[i] && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
          }
          ioff[i] = ((long long)(iy * a.W + ix) * T16 + tt) * a.Cin + 16 * half;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int st = (ky * KS + kx) * NCH + c;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            // UNCONDITIONAL load from a clamped address + select afterwards: a load under a per-lane condition makes
            // hipcc branch around it and drain vmcnt(0) per element, which serialises the whole gather
            const bool use = ok[i] && (c * 32 + 16 * half < a.Cin) && chan_hi_ok;
            const v4i ld = *reinterpret_cast<const v4i*>(inb + (use ? ioff[i] + c * 32 : 0));
            av[st][i] = use ? ld : (v4i){0, 0, 0, 0};
          }
          const int8_t* wp = wg + ((long long)c * KK + (ky * KS + kx)) * 2048;
          bv[st][0] = *reinterpret_cast<const v4i*>(wp);
          bv[st][1] = *reinterpret_cast<const v4i*>(wp + 1024);
        }
      }
    }
  }
  // ---- consume
  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0; acc[i][1][r] = 0; }
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const bool tap_ok = !TR || ((((PY + PAD - ky) % S + S) % S == 0) && (((PX + PAD - kx) % S + S) % S == 0));
      if (tap_ok) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int st = (ky * KS + kx) * NCH + c;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[st][i], bv[st][0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[st][i], bv[st][1], acc[i][1], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- epilogue (identical to the generic kernel)
  const int col = lane & 31, ch = col & 15, odd = col >> 4;
  const int co = g * 16 + ch;
  const bool co_ok = co < a.Cout;
  const double sc = a.scale[g * 16 + ch], bi = a.bias[g * 16 + ch];
  // Pairwise exchange (den_mfma_fp6.hip): v_permlane16_swap(acc[0][ct][r], acc[1][ct][r]) leaves the even lane with both
  // digits of column tile ct of row tile 0 and the odd lane with those of row tile 1 -- 32 swaps give every lane all four
  // digits of ONE neuron for all 16 steps; both lane parities recombine and scan their own tile.
  float xs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const v2u p01 = __builtin_amdgcn_permlane16_swap((unsigned)acc[0][0][r], (unsigned)acc[1][0][r], false, false);
    const v2u p23 = __builtin_amdgcn_permlane16_swap((unsigned)acc[0][1][r], (unsigned)acc[1][1][r], false, false);
    const int hi = (int)p01[0] * 256 + (int)p01[1], lo = (int)p23[0] * 256 + (int)p23[1];
    const double s = fma((double)hi, 65536.0, (double)lo);
    xs[r] = (float)fma(s, sc, bi);
  }
  const int q = pg * 4 + 2 * odd + half;
  const bool pos_ok = q < npos_c;
  const int qy = (int)(((float)q + 0.5f) * inv_wc), qx = q - qy * Wc;
  const int opos = (TR ? qy * S + PY : qy) * a.Wo + (TR ? qx * S + PX : qx);
  const long long HWo = (long long)a.Ho * a.Wo;
  if (MODE == SPK_MODE_LIF) {
    const float bn_a = co_ok ? a.bn_a[co] : 0.f, bn_b = co_ok ? a.bn_b[co] : 0.f;
    const long long vidx = ((long long)b * a.Cout + (co_ok ? co : 0)) * HWo + (pos_ok ? opos : 0);
    float v = (a.v_io && pos_ok && co_ok) ? a.v_io[vidx] : 0.f;
    unsigned mybits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float xv = xs[r];
      const bool s = spk_lif_step_default(v, fmaf(xv, bn_a, bn_b)) && pos_ok && co_ok;
      mybits |= s ? (1u << r) : 0u;
    }
    if (a.v_io && pos_ok && co_ok) a.v_io[vidx] = v;
    if (a.out_f32 && pos_ok && co_ok) {                            // time-collapsed spikes for a linear read-out layer
      float m = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) m = m + ((mybits >> r) & 1u ? a.coef[r] : 0.f);
      a.out_f32[((long long)b * HWo + opos) * a.Cout + co] = m;
    }
    const unsigned bitsv = transpose16_rows_g(mybits, lane);
    if (pos_ok && a.out_s32) {                                     // this task's 16 channels = half of a 32-channel record
      uint2 o2;
      o2.x = spread8_g(bitsv & 0xffu);
      o2.y = spread8_g((bitsv >> 8) & 0xffu);
      *reinterpret_cast<uint2*>(a.out_s32 + ((((long long)b * (a.Cout >> 5) + (g >> 1)) * HWo + opos) * T16 + (lane & 15)) * 16 +
                                8 * (g & 1)) = o2;
    }
    if (pos_ok && a.out_ptc) {
      uint4 o;
      o.x = ((bitsv & 0xfu) * 0x00204081u) & 0x01010101u;
      o.y = (((bitsv >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
      o.z = (((bitsv >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
      o.w = (((bitsv >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
      uint8_t* dst = a.out_ptc + (((long long)b * HWo + opos) * T16 + (lane & 15)) * a.Cout + g * 16;
      if (g * 16 + 16 <= a.Cout) {
        *reinterpret_cast<uint4*>(dst) = o;
      } else {
        const uint8_t* ob = reinterpret_cast<const uint8_t*>(&o);
        for (int c2 = 0; g * 16 + c2 < a.Cout; ++c2) dst[c2] = ob[c2];
      }
    }
  } else {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) m = m + xs[r] * a.coef[r];
    if (pos_ok && co_ok) {
      const float pv = a.apply_tanh ? tanhf(m) : m;
      const long long oidx = ((long long)b * a.Cout + co) * HWo + opos;
      if (a.out_f32) a.out_f32[oidx] = pv;
      if (a.out_u8) a.out_u8[oidx] = (uint8_t)(fminf(fmaxf(pv + 0.5f, 0.0f), 1.0f) * 255.0f);
    }
  }
}

// OCC = waves per SIMD the register allocation must leave room for.  The kernel hides L2 latency by occupancy alone, and the
// "request everything first" body of the larger geometries sits at 200-220 registers (2 waves per SIMD); asking for 3 makes
// hipcc pack accumulators and fragments into 167 without spilling where the tap list is short enough (decoder convT1 /
// convT2, encoder conv3: 0.745 -> 0.604 ms, 0.248 -> 0.215 ms at B = 1024), and spills where it is not (encoder conv2,
// decoder convT3: those keep 2).
template <int MODE, int KS, int S, bool TR, int NCH, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_mfma_gather2_kernel(GArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NCLS = TR ? S * S : 1;
  const int G = (a.Cout + 15) >> 4;
  const int b = blockIdx.y / NCLS, cls = blockIdx.y - b * NCLS;
  const int wt = blockIdx.x * 4 + wave_s;
  const int pg = wt / G, g = wt - pg * G;
  if (TR && S == 2) {
    switch (cls) {
      case 0: gather2_body<MODE, KS, S, TR, NCH, 0, 0>(a, b, g, pg, lane); break;
      case 1: gather2_body<MODE, KS, S, TR, NCH, 0, 1>(a, b, g, pg, lane); break;
      case 2: gather2_body<MODE, KS, S, TR, NCH, 1, 0>(a, b, g, pg, lane); break;
      default: gather2_body<MODE, KS, S, TR, NCH, 1, 1>(a, b, g, pg, lane); break;
    }
  } else {
    gather2_body<MODE, KS, S, TR, NCH, 0, 0>(a, b, g, pg, lane);
  }
}

template <int MODE, int KS, int S, bool TR, int NCH, int OCC = 2>
int launch_gather2(const GArgs& a, hipStream_t stream) {
  constexpr int NCLS = TR ? S * S : 1;
  const int Hc0 = TR ? (a.Ho + S - 1) / S : a.Ho, Wc0 = TR ? (a.Wo + S - 1) / S : a.Wo;
  const int groups = (Hc0 * Wc0 + 3) / 4, G = (a.Cout + 15) / 16;
  const long long by = (long long)a.B * NCLS;
  if (by > 65535) return SPK_ERR_UNSUPPORTED;
  dim3 grid((groups * G + 3) / 4, (unsigned)by), blk(256);
  hipLaunchKernelGGL((conv_mfma_gather2_kernel<MODE, KS, S, TR, NCH, OCC>), grid, blk, 0, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// returns SPK_ERR_UNSUPPORTED when no specialisation matches (the caller then uses the generic kernel)
template <int MODE>
int dispatch_gather2(const GArgs& a, hipStream_t stream) {
  const int nch = (a.Cin + 31) / 32;
  if (a.pad != a.k / 2 || a.B > 16000) return SPK_ERR_UNSUPPORTED;
  if (!a.transposed && a.k == 3 && a.stride == 2 && nch == 1) return launch_gather2<MODE, 3, 2, false, 1>(a, stream);   // enc conv2
#ifndef SPK_G2_OCC_C3
#define SPK_G2_OCC_C3 6         // waves per SIMD the 1x1 layer's register allocation leaves room for: its tasks are one dependent round trip each, the
                                // launch is waves in flight.  3 (round 3: 90 registers, five waves) / 6 (80 registers, 8 dwords of scratch) / 8 (64
                                // registers, 629 dwords of scratch): 43.7 / 35.8 / 580 us at B = 1024 (profiles/r5_ab_kernel_variants.txt (9))
#endif
  if (!a.transposed && a.k == 1 && a.stride == 1 && nch == 2) return launch_gather2<MODE, 1, 1, false, 2, SPK_G2_OCC_C3>(a, stream);   // enc conv3
  if (a.transposed && a.k == 3 && a.stride == 2 && nch == 1) return launch_gather2<MODE, 3, 2, true, 1, 3>(a, stream);     // dec convT1
  if (a.transposed && a.k == 3 && a.stride == 2 && nch == 2) return launch_gather2<MODE, 3, 2, true, 2, 3>(a, stream);     // dec convT2
  if (a.transposed && a.k == 3 && a.stride == 1 && nch == 1) return launch_gather2<MODE, 3, 1, true, 1>(a, stream);     // dec convT3
  return SPK_ERR_UNSUPPORTED;
}


// ------------------------------------------------------------------------------------------------ collapsed read-out
// The decoder's last layer is linear and followed by the membrane read-out sum_t coef[t] * x[t]
// (R/snn_model/vae_model.py:152-154,186; R/snn_model/snn_layers.py:36-41):
//     sum_t coef[t] * (W * s_t + bias) = W * (sum_t coef[t] * s_t) + bias * sum_t coef[t],
// so the producing layer hands over m = sum_t coef[t] * s_t (fp32 [B][H*W][Cin], spk_conv_mfma_fused_fwd) and this kernel
// convolves it ONCE instead of sixteen spike frames: 16x less arithmetic, and the [B][H*W][T][Cin] spike tensor between
// the two layers is never written or read.  fp32 throughout; the result differs from the frame-by-frame sum by fp32
// round-off only (~1e-7; pixels are compared at 1e-4).  One workgroup = RB output rows of one image: the RB + k - 1 input
// rows sit in LDS; a thread = one output pixel x one output channel x half of the input channels.
constexpr int RO_RB = 4;
struct ROArgs {
  const float* x; const float* w; const float* bias; float coef_sum;
  float* out_f32; uint8_t* out_u8; int apply_tanh;
  int B, H, W, Cin, Cout, k, pad, transposed;
};

__global__ __launch_bounds__(256) void readout_collapsed_kernel(ROArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ro_lds[];
  const int nband = (a.H + RO_RB - 1) / RO_RB;
  const int b = blockIdx.x / nband, y0 = (blockIdx.x % nband) * RO_RB;
  const int KK = a.k * a.k;
  const int rows = RO_RB + a.k - 1;
  const int Cp = a.Cin + 4;                                   // row pitch in floats (bank spread)
  float* sx = ro_lds;                                         // [rows][W][Cp]
  float* sw = ro_lds + rows * a.W * Cp;                       // [Cout][KK][Cin]  (taps as a plain correlation)
  for (int i = threadIdx.x; i < a.Cout * KK * a.Cin; i += 256) {
    const int ci = i % a.Cin, tap = (i / a.Cin) % KK, co = i / (a.Cin * KK);
    // correlation form: out[y][x] += X[y + dy - pad'][x + dx - pad'] * wc[dy][dx]; a stride-1 transposed convolution is the
    // correlation with the flipped kernel and pad' = k - 1 - pad
    const int src_tap = a.transposed ? KK - 1 - tap : tap;
    sw[i] = a.transposed ? a.w[((long long)ci * a.Cout + co) * KK + src_tap] : a.w[((long long)co * a.Cin + ci) * KK + src_tap];
  }
  const int padc = a.transposed ? a.k - 1 - a.pad : a.pad;
  const int q4 = a.Cin >> 2;
  for (int i = threadIdx.x; i < rows * a.W * q4; i += 256) {
    const int c4 = i % q4, xx = (i / q4) % a.W, r = i / (q4 * a.W);
    const int yy = y0 + r - padc;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yy >= 0 && yy < a.H) v = *reinterpret_cast<const float4*>(a.x + (((long long)b * a.H + yy) * a.W + xx) * a.Cin + 4 * c4);
    *reinterpret_cast<float4*>(sx + (r * a.W + xx) * Cp + 4 * c4) = v;
  }
  __syncthreads();
  const int npix = RO_RB * a.W;
  const int hc = a.Cin >> 1;                                  // input channels per thread
  for (int u = threadIdx.x >> 1; u < npix * a.Cout; u += 128) {
    const int part = threadIdx.x & 1;
    const int co = u / npix, pix = u % npix;
    const int ry = pix / a.W, ox = pix % a.W;
    float acc = 0.f;
    for (int dy = 0; dy < a.k; ++dy) {
      for (int dx = 0; dx < a.k; ++dx) {
        const int xx = ox + dx - padc;
        if (xx < 0 || xx >= a.W) continue;
        const float* xp = sx + ((ry + dy) * a.W + xx) * Cp + part * hc;
        const float* wp = sw + (co * KK + dy * a.k + dx) * a.Cin + part * hc;
        for (int c = 0; c < hc; c += 4) {
          const float4 xv = *reinterpret_cast<const float4*>(xp + c), wv = *reinterpret_cast<const float4*>(wp + c);
          acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc); acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
      }
    }
    acc += __shfl_xor(acc, 1);
    const int oy = y0 + ry;
    if (part == 0 && oy < a.H) {
      const float m = acc + (a.bias ? a.bias[co] * a.coef_sum : 0.f);
      const float pv = a.apply_tanh ? tanhf(m) : m;
      const long long oidx = ((long long)b * a.Cout + co) * a.H * a.W + (long long)oy * a.W + ox;
      if (a.out_f32) a.out_f32[oidx] = pv;
      if (a.out_u8) a.out_u8[oidx] = (uint8_t)(fminf(fmaxf(pv + 0.5f, 0.0f), 1.0f) * 255.0f);
    }
  }
}

// The same read-out for 3x3 kernels and CIN input channels (the decoder's last layer), bound by the LDS reads of the generic
// form above (two 16-byte reads per four FMAs).  Here a thread computes FOUR adjacent output pixels for a quarter of the input
// channels: per kernel row the six input columns they share are read once (12 reads) and every weight vector serves four
// pixels (6 reads): 54 reads per 288 FMAs.  The image rows sit in LDS with one zero column on either side (no edge tests).
// (WC: the image width as a compile-time constant, 0 = a.W: with a runtime width the two index decodes per staged chunk are integer
//  divisions by a runtime value, ~80 vector instructions each -- rocprofv3 --pmc counted 16.9 M vector instructions per launch at B = 1024
//  for 3.6 M wave-FMAs)
template <int CIN, int RB, int WC>
__global__ __launch_bounds__(256) void readout_collapsed_k3_kernel(ROArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ro_lds[];
  constexpr int Cp = CIN + 4, CQ = CIN / 4, ROWS = RB + 2;
  const int nband = (a.H + RB - 1) / RB;
  const int b = blockIdx.x / nband, y0 = (blockIdx.x % nband) * RB;
  const int Wp = (WC ? WC : a.W) + 2;
  float* sx = ro_lds;                                         // [ROWS][W + 2][Cp]
  float* sw = ro_lds + ROWS * Wp * Cp;                        // [Cout][9][CIN]  (taps as a plain correlation)
  for (int i = threadIdx.x; i < a.Cout * 9 * CIN; i += 256) {
    const int ci = i % CIN, tap = (i / CIN) % 9, co = i / (CIN * 9);
    const int src_tap = a.transposed ? 8 - tap : tap;
    sw[i] = a.transposed ? a.w[((long long)ci * a.Cout + co) * 9 + src_tap] : a.w[((long long)co * CIN + ci) * 9 + src_tap];
  }
  constexpr int q4 = CIN / 4;
  // every load of the band is requested before the first LDS write waits for one (W <= 32: at most NLD per thread)
  constexpr int NLD = (ROWS * 34 * q4 + 255) / 256;
  const int nld = ROWS * Wp * q4;
  float4 stg[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int c4 = i % q4, xc = (i / q4) % Wp, r = i / (q4 * Wp);
    const int yy = y0 + r - 1, xx = xc - 1;                   // (pad' = 1 either way: pad == 1)
    const bool ok = i < nld && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const float4 ld = *reinterpret_cast<const float4*>(a.x + (ok ? (((long long)b * a.H + yy) * a.W + xx) * CIN + 4 * c4 : 0));
    stg[j] = ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int c4 = i % q4, xc = (i / q4) % Wp, r = i / (q4 * Wp);
    if (i < nld) *reinterpret_cast<float4*>(sx + (r * Wp + xc) * Cp + 4 * c4) = stg[j];
  }
  __syncthreads();
  const int part = threadIdx.x & 3, u = threadIdx.x >> 2;
  const int nq = (WC ? WC : a.W) >> 2;
  const int quad = u % nq, ry = u / nq;
  const bool active = ry < RB;                                // (no barrier below: idle threads just skip the stores)
  const int ryc = active ? ry : 0;
  for (int co = 0; co < a.Cout; ++co) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      float4 xr[6][CQ / 4];
      const float* xp = sx + ((ryc + dy) * Wp + 4 * quad) * Cp + part * CQ;
#pragma unroll
      for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int c = 0; c < CQ / 4; ++c) xr[j][c] = *reinterpret_cast<const float4*>(xp + j * Cp + 4 * c);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* wp = sw + (co * 9 + dy * 3 + dx) * CIN + part * CQ;
#pragma unroll
        for (int c = 0; c < CQ / 4; ++c) {
          const float4 wv = *reinterpret_cast<const float4*>(wp + 4 * c);
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const float4 xv = xr[px + dx][c];
            acc[px] = fmaf(xv.x, wv.x, acc[px]); acc[px] = fmaf(xv.y, wv.y, acc[px]);
            acc[px] = fmaf(xv.z, wv.z, acc[px]); acc[px] = fmaf(xv.w, wv.w, acc[px]);
          }
        }
      }
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[px] += __shfl_xor(acc[px], 1);
      acc[px] += __shfl_xor(acc[px], 2);
    }
    const int oy = y0 + ry;
    if (part == 0 && active && oy < a.H) {
      const float bs = a.bias ? a.bias[co] * a.coef_sum : 0.f;
      float pv[4];
      unsigned pk = 0;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float m = acc[px] + bs;
        pv[px] = a.apply_tanh ? tanhf(m) : m;
        pk |= (unsigned)(uint8_t)(fminf(fmaxf(pv[px] + 0.5f, 0.0f), 1.0f) * 255.0f) << (8 * px);
      }
      const long long oidx = ((long long)b * a.Cout + co) * a.H * a.W + (long long)oy * a.W + 4 * quad;
      if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + oidx) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      if (a.out_u8) *reinterpret_cast<unsigned*>(a.out_u8 + oidx) = pk;
    }
  }
}

}  // namespace

extern "C" int spk_readout_collapsed_fwd(const float* x_bpc, const float* w, const float* bias, float coef_sum,
                                         float* out_f32, uint8_t* out_u8, int apply_tanh, int B, int H, int W, int Cin,
                                         int Cout, int k, int pad, int transposed, hipStream_t stream) {
  if (!x_bpc || !w || (!out_f32 && !out_u8) || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || pad < 0)
    return SPK_ERR_ARG;
  if ((k & 1) == 0 || pad != k / 2 || (Cin % 8) != 0) return SPK_ERR_UNSUPPORTED;       // "same" geometry, stride 1
  const size_t lds = ((size_t)(RO_RB + k - 1) * W * (Cin + 4) + (size_t)Cout * k * k * Cin) * sizeof(float);
  if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
  ROArgs a;
  a.x = x_bpc; a.w = w; a.bias = bias; a.coef_sum = coef_sum; a.out_f32 = out_f32; a.out_u8 = out_u8; a.apply_tanh = apply_tanh;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.k = k; a.pad = pad; a.transposed = transposed;
  if (k == 3 && Cin == 32 && (W % 4) == 0 && W <= 32 && Cout <= 8) {
    const bool r7 = (H % 7) == 0;                              // 28 rows: four bands of seven; else bands of eight
    const int rb = r7 ? 7 : 8;
    if ((W / 4) * rb * 4 <= 256) {
      const size_t lds3 = ((size_t)(rb + 2) * (W + 2) * (32 + 4) + (size_t)Cout * 9 * 32) * sizeof(float);
      const long long nb = (long long)B * ((H + rb - 1) / rb);
      if (lds3 <= 64 * 1024 && nb <= 0x7fffffffLL) {
        if (r7 && W == 28) hipLaunchKernelGGL((readout_collapsed_k3_kernel<32, 7, 28>), dim3((unsigned)nb), dim3(256), lds3, stream, a);
        else if (r7) hipLaunchKernelGGL((readout_collapsed_k3_kernel<32, 7, 0>), dim3((unsigned)nb), dim3(256), lds3, stream, a);
        else if (W == 32) hipLaunchKernelGGL((readout_collapsed_k3_kernel<32, 8, 32>), dim3((unsigned)nb), dim3(256), lds3, stream, a);
        else hipLaunchKernelGGL((readout_collapsed_k3_kernel<32, 8, 0>), dim3((unsigned)nb), dim3(256), lds3, stream, a);
        SPK_LAUNCH_CHECK();
        return SPK_OK;
      }
    }
  }
  const long long blocks = (long long)B * ((H + RO_RB - 1) / RO_RB);
  if (blocks > 0x7fffffffLL) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(readout_collapsed_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

// LIF form with nibble-packed "S32" output (Cout % 32 == 0): the input layout of the fp6 kernels (vae_fp6.hip)
extern "C" int spk_conv_mfma_fused_lif_s32(const uint8_t* in_ptc, const int8_t* wq, const double* scale, const double* bias_d,
                                           const float* bn_a, const float* bn_b, float* v_inout, uint8_t* out_s32, int T, int B,
                                           int H, int W, int Cin, int Cout, int k, int stride, int pad, int transposed,
                                           int out_pad, hipStream_t stream) {
  if (!in_ptc || !wq || !scale || !bias_d || !bn_a || !bn_b || !out_s32 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 ||
      Cout <= 0 || k <= 0 || stride <= 0 || pad < 0)
    return SPK_ERR_ARG;
  if (T != T16 || (Cin % 16) != 0 || (Cout % 32) != 0) return SPK_ERR_UNSUPPORTED;
  GArgs a;
  a.out_s32 = out_s32;
  a.in = in_ptc; a.wq = wq; a.scale = scale; a.bias = bias_d; a.bn_a = bn_a; a.bn_b = bn_b; a.v_io = v_inout;
  a.out_ptc = nullptr; a.coef = nullptr; a.out_f32 = nullptr; a.out_u8 = nullptr; a.apply_tanh = 0;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.k = k; a.stride = stride; a.pad = pad;
  a.transposed = transposed;
  a.Ho = spk_conv_out_size(H, k, stride, pad, transposed, out_pad);
  a.Wo = spk_conv_out_size(W, k, stride, pad, transposed, out_pad);
  if (a.Ho <= 0 || a.Wo <= 0) return SPK_ERR_ARG;
  if (dispatch_gather2<SPK_MODE_LIF>(a, stream) == SPK_OK) return SPK_OK;
  const int ncls = transposed ? stride * stride : 1;
  const int Hc0 = transposed ? (a.Ho + stride - 1) / stride : a.Ho;
  const int Wc0 = transposed ? (a.Wo + stride - 1) / stride : a.Wo;
  const long long tasks = (long long)B * ncls * (((long long)Hc0 * Wc0 + 3) / 4) * ((Cout + 15) / 16);
  const long long blocks = (tasks + 3) / 4;
  if (blocks > 0x7fffffffLL) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(conv_mfma_gather_kernel<SPK_MODE_LIF>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

namespace {

// ------------------------------------------------------------------------------------------------ weight packing
// one block per PADDED output channel (ceil(Cout/16)*16): zero digits / zero bias for the padding channels
__global__ __launch_bounds__(256) void pack_i8_generic_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                              int8_t* __restrict__ wq, double* __restrict__ scale,
                                                              double* __restrict__ bias_d, int Cout, int Cin, int k,
                                                              int transposed) {
  __shared__ float smax[256];
  const int co = blockIdx.x, KK = k * k, n = Cin * KK;
  const int nchunks = (Cin + 31) >> 5, g = co >> 4, ch = co & 15;
  const bool real = co < Cout;
  auto wat = [&](int ci, int tap) -> float {
    return transposed ? w[((long long)ci * Cout + co) * KK + tap] : w[((long long)co * Cin + ci) * KK + tap];
  };
  float m = 0.f;
  if (real)
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(wat(i / KK, i % KK)));
  smax[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    __syncthreads();
  }
  m = smax[0];
  int e = 0;
  if (m > 0.f) frexpf(m, &e);
  const int sh = 30 - e;
  if (threadIdx.x == 0) { scale[co] = ldexp(1.0, -sh); bias_d[co] = (real && bias) ? (double)bias[co] : 0.0; }
  const int npad = nchunks * 32 * KK;
  for (int i = threadIdx.x; i < npad; i += 256) {
    const int ci = i / KK, tap = i % KK;
    long long q = (real && ci < Cin) ? (long long)rint(ldexp((double)wat(ci, tap), sh)) : 0;
    int dg[4];
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      int r = (int)(((q + 128) & 255) - 128);
      dg[d] = r;
      q = (q - r) >> 8;
    }
    const int c = ci >> 5, kk = ci & 31;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int ct = d >> 1, colw = (d & 1) * 16 + ch;
      wq[((((long long)(g * nchunks + c) * KK + tap) * 2 + ct) * 32 + colw) * 32 + (kk ^ (colw & 16))] = (int8_t)dg[d];
    }
  }
}

}  // namespace

extern "C" long long spk_conv_packed_weight_i8_bytes(int Cout, int Cin, int k) {
  if (Cout <= 0 || Cin <= 0 || k <= 0) return -1;
  return (long long)((Cout + 15) / 16) * ((Cin + 31) / 32) * k * k * 2048;
}

extern "C" int spk_pack_conv_weight_i8(const float* w, const float* bias, int8_t* wq, double* scale, double* bias_d,
                                       int Cout, int Cin, int k, int transposed, hipStream_t stream) {
  if (!w || !wq || !scale || !bias_d || Cout <= 0 || Cin <= 0 || k <= 0) return SPK_ERR_ARG;
  const int cpad = ((Cout + 15) / 16) * 16;
  hipLaunchKernelGGL(pack_i8_generic_kernel, dim3(cpad), dim3(256), 0, stream, w, bias, wq, scale, bias_d, Cout, Cin, k,
                     transposed);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_conv_mfma_fused_fwd(const uint8_t* in_ptc, const int8_t* wq, const double* scale,
                                       const double* bias_d, const float* bn_a, const float* bn_b, float* v_inout,
                                       uint8_t* out_ptc, const float* coef, float* out_f32, uint8_t* out_u8,
                                       int apply_tanh, int mode, int T, int B, int H, int W, int Cin, int Cout, int k,
                                       int stride, int pad, int transposed, int out_pad, hipStream_t stream) {
  if (!in_ptc || !wq || !scale || !bias_d || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 ||
      stride <= 0 || pad < 0)
    return SPK_ERR_ARG;
  if (T != T16 || (Cin % 16) != 0) return SPK_ERR_UNSUPPORTED;
  GArgs a;
  a.out_s32 = nullptr;
  a.in = in_ptc; a.wq = wq; a.scale = scale; a.bias = bias_d; a.bn_a = bn_a; a.bn_b = bn_b; a.v_io = v_inout;
  a.out_ptc = out_ptc; a.coef = coef; a.out_f32 = out_f32; a.out_u8 = out_u8; a.apply_tanh = apply_tanh;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.k = k; a.stride = stride; a.pad = pad;
  a.transposed = transposed;
  a.Ho = spk_conv_out_size(H, k, stride, pad, transposed, out_pad);
  a.Wo = spk_conv_out_size(W, k, stride, pad, transposed, out_pad);
  if (a.Ho <= 0 || a.Wo <= 0) return SPK_ERR_ARG;
  const int ncls = transposed ? stride * stride : 1;
  const int Hc0 = transposed ? (a.Ho + stride - 1) / stride : a.Ho;
  const int Wc0 = transposed ? (a.Wo + stride - 1) / stride : a.Wo;
  const long long groups = ((long long)Hc0 * Wc0 + 3) / 4;
  const long long tasks = (long long)B * ncls * groups * ((Cout + 15) / 16);
  const long long blocks = (tasks + 3) / 4;
  if (blocks > 0x7fffffffLL) return SPK_ERR_UNSUPPORTED;
  dim3 grid((unsigned)blocks), blk(256);
  if (mode == SPK_MODE_LIF) {
    // (coef AND out_f32 given: out_f32 receives sum_t coef[t] * spike[t] as [B][Ho*Wo][Cout]; out_ptc may then be null)
    if (!bn_a || !bn_b || (!out_ptc && !(coef && out_f32)) || (out_f32 && !coef)) return SPK_ERR_ARG;
    if (!coef) a.out_f32 = nullptr;
    if (dispatch_gather2<SPK_MODE_LIF>(a, stream) == SPK_OK) return SPK_OK;          // compile-time geometry
    hipLaunchKernelGGL(conv_mfma_gather_kernel<SPK_MODE_LIF>, grid, blk, 0, stream, a);
  } else if (mode == SPK_MODE_MEMOUT) {
    if (!coef || (!out_f32 && !out_u8)) return SPK_ERR_ARG;
    if (dispatch_gather2<SPK_MODE_MEMOUT>(a, stream) == SPK_OK) return SPK_OK;
    hipLaunchKernelGGL(conv_mfma_gather_kernel<SPK_MODE_MEMOUT>, grid, blk, 0, stream, a);
  } else {
    return SPK_ERR_UNSUPPORTED;
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
