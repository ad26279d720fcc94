// Direct (non-MFMA) convolution kernels of the SNN path.
//
//   spk_pack_conv_weight      [Cout,Cin,k,k] / [Cin,Cout,k,k] (ConvT)  ->  packed [k*k][Cin][Cout]
//   spk_conv2d_fwd            layer.Conv2d 'm'-mode body     SJ/activation_based/layer.py:164-173
//   spk_conv_transpose2d_fwd  layer.ConvTranspose2d body     SJ/activation_based/layer.py:316-325
//   spk_conv_fused_fwd        (Conv|ConvT) [+ BN + LIF | raw | membrane read-out | time mean] fused:
//                             R/snn_model/vae_model.py:109-124 (Encoder), :34-38 (poisson), :139-155 (Decoder),
//                             R/snn_model/vq_diffusion.py:161-187,200-206 (denoiser; MFMA kernel replaces conv2-5)
//
// Numerics contract of every convolution in this library: the pre-activation is the CORRECTLY ROUNDED fp32
// value of the exact dot product (+ bias), obtained by accumulating in fp64 (products of two fp32 are exact in
// fp64).  It is therefore independent of accumulation order, bit-reproducible, and at most half an ulp away
// from the real number that oneDNN's fp32 result approximates.  BN is y = fma(x, a, b) (fixture F7), LIF is the
// reference's arithmetic (spk_common.h).
//
// Spike tensors between fused layers are u8 {0,1} in "PTC" layout [B][H][W][T][C] (position, time, channel):
// the T steps of one neuron's inputs are adjacent, and channels are the unit-stride dimension that both the
// direct kernels (lanes = output channels) and the MFMA kernel (K = taps x channels) read as 4..16 B vectors.
#include "den_common.h"
#include "../../include/spkdiff.h"
#include <math.h>
#include <string.h>

namespace {

__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int kk,
                                   int transposed) {
  int total = Cout * Cin * kk;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int co = i % Cout;
    int ci = (i / Cout) % Cin;
    int tap = i / (Cout * Cin);
    // Conv2d weight [Cout][Cin][kk]; ConvTranspose2d weight [Cin][Cout][kk]
    out[i] = transposed ? w[((long long)ci * Cout + co) * kk + tap] : w[((long long)co * Cin + ci) * kk + tap];
  }
}

// ---------------------------------------------------------------- generic fp32 NCHW conv (API surface)
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void conv_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        long long M, int Cin, int H, int W, int Cout, int Ho, int Wo,
                                                        int k, int stride, int pad) {
  const long long total = M * Cout * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int ox = (int)(i % Wo);
    long long r = i / Wo;
    int oy = (int)(r % Ho); r /= Ho;
    int co = (int)(r % Cout);
    long long m = r / Cout;
    double acc = bias ? (double)bias[co] : 0.0;
    const float* xm = x + m * Cin * H * W;
    for (int ky = 0; ky < k; ++ky) {
      int iy;
      if (TRANSPOSED) {
        int ty = oy + pad - ky;
        if (ty < 0 || ty % stride) continue;
        iy = ty / stride;
      } else {
        iy = oy * stride - pad + ky;
      }
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        int ix;
        if (TRANSPOSED) {
          int tx = ox + pad - kx;
          if (tx < 0 || tx % stride) continue;
          ix = tx / stride;
        } else {
          ix = ox * stride - pad + kx;
        }
        if (ix < 0 || ix >= W) continue;
        const float* xp = xm + (long long)iy * W + ix;
        const int tap = ky * k + kx;
        for (int ci = 0; ci < Cin; ++ci) {
          float wv = TRANSPOSED ? w[((long long)ci * Cout + co) * k * k + tap] : w[((long long)co * Cin + ci) * k * k + tap];
          acc += (double)xp[(long long)ci * H * W] * (double)wv;
        }
      }
    }
    y[i] = (float)acc;
  }
}

// ---------------------------------------------------------------- fused direct kernel
struct FusedArgs {
  const void* in0;      // PTC: u8 [B,H,W,T,C0]; TINV: fp32 [B,Cin,H,W]; SEQ: fp32 [T,B,Cin,H,W]
  const uint8_t* in1;   // optional second PTC source concatenated after in0 along channels (C1 channels)
  int C0, C1;
  const float* wt;      // packed [k*k][Cin][Cout], Cin = C0 + C1
  const float* bias;    // [Cout] or null
  const float* bn_a;    // [Cout]  (MODE_LIF)
  const float* bn_b;
  float* v_io;          // [B,Cout,Ho,Wo] membrane potential in/out, or null (= fresh state, not written back)
  uint8_t* out_ptc;     // MODE_LIF: spikes u8 [B,Ho,Wo,T,Cout] or null
  float* out_f32;       // MODE_LIF: spikes fp32 [T,B,Cout,Ho,Wo] or null; MODE_RAW: x [T,B,Cout,Ho,Wo];
                        // MODE_MEMOUT / MODE_MEAN: [B,Cout,Ho,Wo]
  float* out_pre;       // optional BN output: TINV [B,Cout,Ho,Wo]; else [T,B,Cout,Ho,Wo]
  uint8_t* out_u8;      // MODE_MEMOUT: uint8 image [B,Cout,Ho,Wo] or null
  const float* coef;    // MODE_MEMOUT: [T]
  int apply_tanh;
  int T, B, H, W, Cout, Ho, Wo, k, stride, pad;
  uint8_t* out_cnt;     // MODE_LIF, optional: spike counts over T, u8 [B][Cout/32][Ho*Wo][32]
  int chunk0, chunk1, chunk_out;   // channel chunking of the PTC tensors ([B][C/chunk][HW][T][chunk]); chunk == C: plain
  int out_c4;           // 64 / 32: out_ptc is nibble-packed fp4 ("C4" [B][Cout/64][HW][T][32 B] / "S32" [B][Cout/32][HW][T][16 B])
  const int* n_dyn;     // optional device-side batch count (<= B): only the first *n_dyn images are processed
};

template <int INKIND, bool TRANSPOSED, int MODE>
__global__ __launch_bounds__(256) void conv_fused_kernel(FusedArgs a) {
  constexpr bool TINV = INKIND == SPK_IN_TINV;
  const int Cin = a.C0 + a.C1;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;     // buffers and strides stay sized by a.B
  const long long total = (long long)Bn * a.Ho * a.Wo * a.Cout;
  const int T = a.T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % a.Cout);
    long long r = i / a.Cout;
    const int ox = (int)(r % a.Wo); r /= a.Wo;
    const int oy = (int)(r % a.Ho);
    const int b = (int)(r / a.Ho);

    double acc[TINV ? 1 : SPK_MAX_T];
    const double b0 = a.bias ? (double)a.bias[co] : 0.0;
#pragma unroll
    for (int t = 0; t < (TINV ? 1 : SPK_MAX_T); ++t) acc[t] = b0;

    for (int ky = 0; ky < a.k; ++ky) {
      int iy;
      if (TRANSPOSED) {
        int ty = oy + a.pad - ky;
        if (ty < 0 || ty % a.stride) continue;
        iy = ty / a.stride;
      } else {
        iy = oy * a.stride - a.pad + ky;
      }
      if (iy < 0 || iy >= a.H) continue;
      for (int kx = 0; kx < a.k; ++kx) {
        int ix;
        if (TRANSPOSED) {
          int tx = ox + a.pad - kx;
          if (tx < 0 || tx % a.stride) continue;
          ix = tx / a.stride;
        } else {
          ix = ox * a.stride - a.pad + kx;
        }
        if (ix < 0 || ix >= a.W) continue;
        const float* wp = a.wt + ((long long)(ky * a.k + kx) * Cin) * a.Cout + co;
        if constexpr (TINV) {
          const float* xp = reinterpret_cast<const float*>(a.in0) + ((long long)b * Cin * a.H + iy) * a.W + ix;
          for (int ci = 0; ci < Cin; ++ci)
            acc[0] += (double)xp[(long long)ci * a.H * a.W] * (double)wp[(long long)ci * a.Cout];
        } else if constexpr (INKIND == SPK_IN_SEQ) {
          const long long chw = (long long)a.H * a.W;
          const float* xp = reinterpret_cast<const float*>(a.in0) + ((long long)b * Cin * a.H + iy) * a.W + ix;
          const long long ts = (long long)a.B * Cin * chw;
          for (int ci = 0; ci < Cin; ++ci) {
            const double wv = (double)wp[(long long)ci * a.Cout];
#pragma unroll
            for (int t = 0; t < SPK_MAX_T; ++t)
              if (t < T) acc[t] += (double)xp[t * ts + ci * chw] * wv;
          }
        } else {
          const int ipos = iy * a.W + ix, HWi = a.H * a.W;
          for (int src = 0; src < 2; ++src) {
            const int Cs = src ? a.C1 : a.C0;
            if (Cs == 0) continue;
            const int chk = src ? a.chunk1 : a.chunk0;
            const uint8_t* sbase = (src ? a.in1 : reinterpret_cast<const uint8_t*>(a.in0)) + (long long)b * HWi * T * Cs;
            const float* wsrc = wp + (src ? (long long)a.C0 * a.Cout : 0);
            for (int ci = 0; ci < Cs; ci += 4) {
              // [C/chunk][HW][T][chunk]: 4 consecutive channels never straddle a chunk (chunk % 4 == 0)
              const uint8_t* sp = sbase + ((long long)(ci / chk) * HWi + ipos) * T * chk + (ci % chk);
              const float w0 = wsrc[(long long)(ci + 0) * a.Cout], w1 = wsrc[(long long)(ci + 1) * a.Cout];
              const float w2 = wsrc[(long long)(ci + 2) * a.Cout], w3 = wsrc[(long long)(ci + 3) * a.Cout];
#pragma unroll
              for (int t = 0; t < SPK_MAX_T; ++t) {
                if (t < T) {
                  const uint32_t s4 = *reinterpret_cast<const uint32_t*>(sp + t * chk);
                  if (s4) {
                    // exact: the selected weights are added in fp64
                    acc[t] += (s4 & 0x000000ffu) ? (double)w0 : 0.0;
                    acc[t] += (s4 & 0x0000ff00u) ? (double)w1 : 0.0;
                    acc[t] += (s4 & 0x00ff0000u) ? (double)w2 : 0.0;
                    acc[t] += (s4 & 0xff000000u) ? (double)w3 : 0.0;
                  }
                }
              }
            }
          }
        }
      }
    }

    const long long plane = (long long)a.Ho * a.Wo;
    const long long o_bchw = (((long long)b * a.Cout + co) * a.Ho + oy) * a.Wo + ox;     // [B,Cout,Ho,Wo]
    const long long tstride = (long long)a.B * a.Cout * plane;                            // [T,B,Cout,Ho,Wo]

    if constexpr (MODE == SPK_MODE_LIF) {
      const float al = a.bn_a[co], be = a.bn_b[co];
      float v = a.v_io ? a.v_io[o_bchw] : 0.0f;
      float y0 = 0.f;
      int nspk = 0;
      unsigned mybits = 0;
      if constexpr (TINV) {
        y0 = fmaf((float)acc[0], al, be);
        if (a.out_pre) a.out_pre[o_bchw] = y0;
      }
      const long long o_ptc = ((((long long)b * (a.Cout / a.chunk_out) + co / a.chunk_out) * plane + oy * a.Wo + ox) * T) *
                                  a.chunk_out + co % a.chunk_out;
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t) {
        if (t < T) {
          float y;
          if constexpr (TINV) {
            y = y0;
          } else {
            y = fmaf((float)acc[t], al, be);
            if (a.out_pre) a.out_pre[o_bchw + t * tstride] = y;
          }
          const bool s = spk_lif_step_default(v, y);
          nspk += s ? 1 : 0;
          if (a.out_c4) {
            mybits |= s ? (1u << t) : 0u;                  // stored after the scan (below)
          } else if (a.out_ptc) {
            a.out_ptc[o_ptc + (long long)t * a.chunk_out] = (uint8_t)s;
          }
          if (a.out_f32) a.out_f32[o_bchw + t * tstride] = s ? 1.0f : 0.0f;
        }
      }
      if (a.out_c4) {
        // nibble-packed fp4 output: lanes are consecutive output channels (Cout % 16 == 0, so the 16 lanes of a DPP row
        // are 16 channels of ONE position); a 16x16 bit transpose per row gives lane t the 16 channel bits of step t =
        // 8 bytes of the (position, t) record -- one 8-byte store per lane instead of a byte store per channel pair and step
        const unsigned bitsv = spk_transpose16_rows(mybits, (int)(threadIdx.x & 63));
        const int tl = (int)(threadIdx.x & 15), co16 = co & ~15;
        if (tl < T) {
          auto spread8 = [](unsigned x) -> unsigned {        // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
            x = (x | (x << 12)) & 0x000f000fu;
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;
            return x << 1;
          };
          uint2 o;
          o.x = spread8(bitsv & 0xffu);
          o.y = spread8((bitsv >> 8) & 0xffu);
          uint8_t* dst = a.out_ptc + ((((long long)b * (a.Cout / a.out_c4) + (co16 / a.out_c4)) * plane + oy * a.Wo + ox) * T + tl) *
                                         (a.out_c4 >> 1) + ((co16 % a.out_c4) >> 1);
          *reinterpret_cast<uint2*>(dst) = o;
        }
      }
      if (a.v_io) a.v_io[o_bchw] = v;
      if (a.out_cnt) a.out_cnt[(((long long)b * (a.Cout >> 5) + (co >> 5)) * plane + oy * a.Wo + ox) * 32 + (co & 31)] = (uint8_t)nspk;
    } else if constexpr (MODE == SPK_MODE_RAW) {
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t)
        if (t < T) a.out_f32[o_bchw + t * tstride] = (float)acc[TINV ? 0 : t];
    } else if constexpr (MODE == SPK_MODE_MEMOUT) {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t)
        if (t < T) m = m + (float)acc[TINV ? 0 : t] * a.coef[t];          // torch.sum(x * coef, dim=0)
      const float p = a.apply_tanh ? tanhf(m) : m;
      if (a.out_f32) a.out_f32[o_bchw] = p;
      if (a.out_u8) {
        // R/main.py:401: np.array(np.clip(pred + 0.5, 0, 1) * 255, dtype=np.uint8)  (truncating cast)
        float q = fminf(fmaxf(p + 0.5f, 0.0f), 1.0f) * 255.0f;
        a.out_u8[o_bchw] = (uint8_t)q;
      }
    } else {  // SPK_MODE_MEAN: torch.sum(x6, dim=0) / T     R/snn_model/vq_diffusion.py:206
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < SPK_MAX_T; ++t)
        if (t < T) m = m + (float)acc[TINV ? 0 : t];
      a.out_f32[o_bchw] = m / (float)T;
    }
  }
}

// ---------------------------------------------------------------- time-invariant input + BN + LIF, lean form
// The first layer of the encoder and the spike generator see the SAME frame at every step (R/main.py:309,
// vae_model.py:54-56): one dot product per neuron, then the 16-step scan.  The generic kernel above carries every mode and
// layout in one body (1 800 instructions, scalar registers spilled to lanes); this one is the case the pipeline runs --
// plain convolution, T = 16, Cout a multiple of 16 that divides 256, spikes out as plain u8 PTC or nibble-packed S32 / C4,
// optional membrane state and spike counts -- with the same arithmetic in the same order (fp64 accumulation over ky, kx,
// ci; BN fma; the reference's LIF step).  A thread keeps its output channel for the whole launch and walks positions.
struct TinvArgs {
  const float* x; const float* wt; const float* bias; const float* bn_a; const float* bn_b;
  float* v_io; uint8_t* out; uint8_t* out_cnt;
  int B, Cin, H, W, Cout, Ho, Wo, k, stride, pad;
  int out_c4;           // 0: u8 PTC [B][HW][16][Cout]; 32 / 64: S32 / C4
  const int* n_dyn;
};

// the sixteen spike bits of a neuron -> counts + the layer's output format.  Lanes are consecutive output channels (Cout % 16 == 0: the 16
// lanes of a DPP row are 16 channels of ONE position); a 16x16 bit transpose per row gives lane t the 16 channel bits of step t
__device__ __forceinline__ void tinv_store(const TinvArgs& a, unsigned mybits, int b, int op, int co, int lane, bool ok, int plane) {
  if (a.out_cnt && ok) a.out_cnt[(((long long)b * (a.Cout >> 5) + (co >> 5)) * plane + op) * 32 + (co & 31)] = (uint8_t)__popc(mybits);
  const unsigned bitsv = spk_transpose16_rows(mybits, lane);
  const int tl = lane & 15, co16 = co & ~15;
  if (!ok) return;
  if (a.out_c4) {
    auto spread8 = [](unsigned x) -> unsigned {        // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
      x = (x | (x << 12)) & 0x000f000fu;
      x = (x | (x << 6)) & 0x03030303u;
      x = (x | (x << 3)) & 0x11111111u;
      return x << 1;
    };
    uint2 o;
    o.x = spread8(bitsv & 0xffu);
    o.y = spread8((bitsv >> 8) & 0xffu);
    uint8_t* dst = a.out + ((((long long)b * (a.Cout / a.out_c4) + (co16 / a.out_c4)) * plane + op) * 16 + tl) * (a.out_c4 >> 1) +
                   ((co16 % a.out_c4) >> 1);
    *reinterpret_cast<uint2*>(dst) = o;
  } else {
    uint4 o;
    o.x = ((bitsv & 0xfu) * 0x00204081u) & 0x01010101u;
    o.y = (((bitsv >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
    o.z = (((bitsv >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
    o.w = (((bitsv >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
    *reinterpret_cast<uint4*>(a.out + (((long long)b * plane + op) * 16 + tl) * a.Cout + co16) = o;
  }
}

// KC > 0: k = KS and Cin = KC are compile-time and the thread's KS * KS * KC weights live in registers
template <int KS, int KC>
__global__ __launch_bounds__(256) void tinv_lif_kernel(TinvArgs a) {
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  const int co = threadIdx.x % a.Cout, pl = threadIdx.x / a.Cout, ppb = 256 / a.Cout;     // positions per block step
  const int plane = a.Ho * a.Wo;
  const int npos = Bn * plane;
  const float al = a.bn_a[co], be = a.bn_b[co];
  const double b0 = a.bias ? (double)a.bias[co] : 0.0;
  const int lane = threadIdx.x & 63;
  constexpr int NW = KC > 0 ? KS * KS * KC : 1;
  float wreg[NW];
  if constexpr (KC > 0) {
#pragma unroll
    for (int i = 0; i < NW; ++i) wreg[i] = a.wt[i * a.Cout + co];            // packed [k*k][Cin][Cout]
  }
  const int HWi = a.H * a.W;
  const bool small_idx = npos < (1 << 24);
  const float inv_plane = 1.0f / (float)plane, inv_wo = 1.0f / (float)a.Wo;
  // stateless calls (v = 0 at step 0): the spike train of a constant input is a table look-up (spk_common.h)
  __shared__ float s_th[16];
  __shared__ unsigned s_pat[18];
  {
    constexpr unsigned thb[16] = SPK_LIF_CONST_TH_BITS, pat[18] = SPK_LIF_CONST_PATTERNS;
    if (threadIdx.x < 16) s_th[threadIdx.x] = __uint_as_float(thb[threadIdx.x]);
    if (threadIdx.x < 18) s_pat[threadIdx.x] = pat[threadIdx.x];
  }
  __syncthreads();
  for (int p0 = blockIdx.x * ppb; p0 < npos; p0 += gridDim.x * ppb) {
    const int pg = p0 + pl;
    const bool ok = pg < npos;
    const int pc = ok ? pg : npos - 1;
    // (quotients by reciprocal multiplication + one correction step: exact for operands below 2^24; a 32-bit integer division
    //  is ~40 vector instructions, two of them were a quarter of this kernel's work per neuron)
    int b, op, oy, ox;
    if (small_idx) {
      b = (int)((float)pc * inv_plane);
      op = pc - b * plane;
      if (op < 0) { --b; op += plane; } else if (op >= plane) { ++b; op -= plane; }
      oy = (int)((float)op * inv_wo);
      ox = op - oy * a.Wo;
      if (ox < 0) { --oy; ox += a.Wo; } else if (ox >= a.Wo) { ++oy; ox -= a.Wo; }
    } else {
      b = pc / plane; op = pc - b * plane;
      oy = op / a.Wo; ox = op - oy * a.Wo;
    }
    double acc = b0;
    if constexpr (KC > 0) {
      const float* xb = a.x + (long long)b * KC * HWi;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int ix = ox * a.stride - a.pad + kx;
          const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          const int off = in ? iy * a.W + ix : 0;
#pragma unroll
          for (int ci = 0; ci < KC; ++ci) {
            // (an out-of-image tap adds an exact zero: same sum as skipping it.  Unconditional load from a clamped address +
            //  select: a load under a per-lane condition makes hipcc branch around it and wait for it alone -- KS * KS * KC
            //  round trips one after the other, 18 us of the denoiser's conv1 launch)
            const float ld = xb[ci * HWi + off];
            const float xv = in ? ld : 0.0f;
            acc = fma((double)xv, (double)wreg[(ky * KS + kx) * KC + ci], acc);   // (fp32 x fp32 is exact in fp64: == acc + x * w)
          }
        }
      }
    } else {
      for (int ky = 0; ky < a.k; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        if (iy < 0 || iy >= a.H) continue;
        for (int kx = 0; kx < a.k; ++kx) {
          const int ix = ox * a.stride - a.pad + kx;
          if (ix < 0 || ix >= a.W) continue;
          const float* wp = a.wt + (long long)(ky * a.k + kx) * a.Cin * a.Cout + co;
          const float* xp = a.x + ((long long)b * a.Cin * a.H + iy) * a.W + ix;
          for (int ci = 0; ci < a.Cin; ++ci) acc += (double)xp[(long long)ci * a.H * a.W] * (double)wp[(long long)ci * a.Cout];
        }
      }
    }
    const float y0 = fmaf((float)acc, al, be);
    const long long o_bchw = ((long long)b * a.Cout + co) * plane + op;
    unsigned mybits = 0;
    if (a.v_io) {                                            // (uniform) carried membrane state: the sixteen steps
      float v = ok ? a.v_io[o_bchw] : 0.0f;
#pragma unroll
      for (int t = 0; t < 16; ++t) mybits |= spk_lif_step_default(v, y0) ? (1u << t) : 0u;
      if (ok) a.v_io[o_bchw] = v;
    } else {
      mybits = spk_lif_const_input_bits16(y0, s_th, s_pat);
    }
    tinv_store(a, mybits, b, op, co, lane, ok, plane);
  }
}

// Round 5: the same layer with the per-POSITION work done once per position.  In tinv_lif_kernel a thread is a (position, channel) pair
// and every one of the Cout lanes of a position repeats its index arithmetic, bounds tests, pixel requests and fp32 -> fp64 conversions:
// 225 vector instructions per wave and pair of positions for 18 multiply-adds each (rocprofv3 --pmc: 22.6 M vector instructions per launch
// of the encoder's first layer at B = 1024).  Here 128 threads of the block first take one position each of a chunk of 128 -- indices,
// KS * KS * KC pixels, converted -- and leave them in LDS as doubles; then every thread keeps its channel, walks the chunk's positions and
// reads a position's pixels with broadcast LDS reads.  Same arithmetic in the same order (fp64 accumulation over ky, kx, ci from the bias;
// an out-of-image tap adds an exact zero).  Stateless calls only (v = 0 at step 0: the spike train is the table look-up).
#ifndef SPK_TINV_PB
#define SPK_TINV_PB 64          // positions per chunk (<= 256): 32 / 64 / 128 / 256 measured below (profiles/r5_ab_kernel_variants.txt (9))
#endif
template <int KS, int KC>
__global__ __launch_bounds__(256) void tinv_lif_staged_kernel(TinvArgs a) {
  constexpr int NX = KS * KS * KC, PB = SPK_TINV_PB;
  __shared__ double sX[PB][NX];
  __shared__ int sB[PB], sOp[PB];
  __shared__ float s_th[16];
  __shared__ unsigned s_pat[18];
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  const int co = threadIdx.x % a.Cout, pl = threadIdx.x / a.Cout, ppb = 256 / a.Cout;
  const int plane = a.Ho * a.Wo;
  const int npos = Bn * plane;
  const float al = a.bn_a[co], be = a.bn_b[co];
  const double b0 = a.bias ? (double)a.bias[co] : 0.0;
  const int lane = threadIdx.x & 63;
  double wreg[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) wreg[i] = (double)a.wt[i * a.Cout + co];      // packed [k*k][Cin][Cout]
  const int HWi = a.H * a.W;
  {
    constexpr unsigned thb[16] = SPK_LIF_CONST_TH_BITS, pat[18] = SPK_LIF_CONST_PATTERNS;
    if (threadIdx.x < 16) s_th[threadIdx.x] = __uint_as_float(thb[threadIdx.x]);
    if (threadIdx.x < 18) s_pat[threadIdx.x] = pat[threadIdx.x];
  }
  for (int c0 = blockIdx.x * PB; c0 < npos; c0 += gridDim.x * PB) {
    __syncthreads();                                         // the previous chunk's readers are done (first trip: the tables are in)
    if (threadIdx.x < PB) {
      const int pg = c0 + (int)threadIdx.x;
      const int pc = pg < npos ? pg : npos - 1;
      const int b = pc / plane, op = pc - b * plane;
      const int oy = op / a.Wo, ox = op - oy * a.Wo;
      sB[threadIdx.x] = b; sOp[threadIdx.x] = op;
      const float* xb = a.x + (long long)b * KC * HWi;
      float xv[NX];
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int ix = ox * a.stride - a.pad + kx;
          const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          const int off = in ? iy * a.W + ix : 0;
#pragma unroll
          for (int ci = 0; ci < KC; ++ci) {
            const float ld = xb[ci * HWi + off];             // (unconditional request from a clamped address + select, as above)
            xv[(ky * KS + kx) * KC + ci] = in ? ld : 0.0f;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) sX[threadIdx.x][i] = (double)xv[i];
    }
    __syncthreads();
    for (int j = pl; j < PB; j += ppb) {
      const bool ok = c0 + j < npos;
      double acc = b0;
#pragma unroll
      for (int i = 0; i < NX; ++i) acc = fma(sX[j][i], wreg[i], acc);          // (fp32 x fp32 is exact in fp64: == acc + x * w)
      const float y0 = fmaf((float)acc, al, be);
      const unsigned mybits = spk_lif_const_input_bits16(y0, s_th, s_pat);
      tinv_store(a, mybits, sB[j], sOp[j], co, lane, ok, plane);
    }
  }
}

// Round 5: the decoder's front end -- embedding gather, spike generator (1x1 convolution of the code vector + BN + LIF from the reset state
// on a constant input) and the nibble packing of its spikes -- as a table look-up by TOKEN.  The generator's input at a position is one of
// the K codebook rows, so its sixteen-step spike train per output channel is one of K patterns: spikegen_table_kernel computes them with
// the arithmetic of tinv_lif_kernel (fp64 dot product over the code's components from the bias, BN fma, the constant-input look-up; row K
// = an out-of-range token, whose embedding is NaN: no spikes), spikegen_expand_kernel writes a position's S32 records from its token's row.
// Replaces three launches (embedding 6 us, generator 12 us, PTC -> S32 6 us at B = 1024) and 13 MB of intermediate spikes.
__global__ __launch_bounds__(256) void spikegen_table_kernel(const float* __restrict__ cb, const float* __restrict__ wt, const float* __restrict__ bias,
                                                             const float* __restrict__ bn_a, const float* __restrict__ bn_b,
                                                             unsigned short* __restrict__ table, int K, int D, int Cout) {
  __shared__ float s_th[16];
  __shared__ unsigned s_pat[18];
  {
    constexpr unsigned thb[16] = SPK_LIF_CONST_TH_BITS, pat[18] = SPK_LIF_CONST_PATTERNS;
    if (threadIdx.x < 16) s_th[threadIdx.x] = __uint_as_float(thb[threadIdx.x]);
    if (threadIdx.x < 18) s_pat[threadIdx.x] = pat[threadIdx.x];
  }
  __syncthreads();
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (K + 1) * Cout; i += gridDim.x * blockDim.x) {
    const int k = i / Cout, co = i - k * Cout;
    double acc = bias ? (double)bias[co] : 0.0;
    for (int c0 = 0; c0 < D; c0 += 16) {                     // (sixteen components requested at once; beyond D: exact zeros)
      float xv[16], wv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int ci = c0 + j;
        const bool in = ci < D;
        const float xl = cb[(k < K ? k : 0) * D + (in ? ci : 0)], wl = wt[(in ? ci : 0) * Cout + co];      // packed [1][D][Cout]
        xv[j] = in ? (k < K ? xl : __builtin_nanf("")) : 0.0f;
        wv[j] = in ? wl : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (c0 + j < D) acc = fma((double)xv[j], (double)wv[j], acc);
    }
    const float y0 = fmaf((float)acc, bn_a[co], bn_b[co]);
    table[i] = (unsigned short)spk_lif_const_input_bits16(y0, s_th, s_pat);
  }
}

template <int COUT>
__global__ __launch_bounds__(256) void spikegen_expand_kernel(const long long* __restrict__ tok, const unsigned short* __restrict__ table,
                                                              uint8_t* __restrict__ out, long long npos, int K) {
  auto spread8 = [](unsigned x) -> unsigned {          // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x << 1;
  };
  const long long total = npos * 16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i >> 4;
    const int t = (int)(i & 15);
    const long long tk = tok[p];
    const int row = (tk >= 0 && tk < K) ? (int)tk : K;
    const uint4* rp = reinterpret_cast<const uint4*>(table + (long long)row * COUT);     // COUT u16 patterns: 32 / 64 bytes
    unsigned mask = 0;
#pragma unroll
    for (int q = 0; q < COUT / 8; ++q) {
      const uint4 v = rp[q];
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mask |= ((w4[j] >> t) & 1u) << (8 * q + 2 * j);
        mask |= ((w4[j] >> (16 + t)) & 1u) << (8 * q + 2 * j + 1);
      }
    }
    uint4 o;
    o.x = spread8(mask & 0xffu);
    o.y = spread8((mask >> 8) & 0xffu);
    o.z = COUT > 16 ? spread8((mask >> 16) & 0xffu) : 0u;
    o.w = COUT > 16 ? spread8((mask >> 24) & 0xffu) : 0u;
    *reinterpret_cast<uint4*>(out + i * 16) = o;                                          // [position][t][16 B]
  }
}

inline int grid_for(long long work_items) {
  long long g = (work_items + 255) / 256;
  const long long cap = 256 * 8 * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// RAW output of a spike-input layer with few output channels, one thread per (output position, STEP): the module-API form of the
// decoder's read-out layer (R/main.py:397: `pred = model.decoder(quantized)` returns the per-step convolution [T,B,C,28,28], memout and
// tanh are main.py's own calls).  conv_fused_kernel keeps all T accumulators of an output in one thread: B x 784 x C threads -- 49
// workgroups at R/main.py's B = 16, 228 us.  Here T x as many threads each walk the taps once; a tap is one 16-byte read per 16 input
// channels of the (position, step) record, spikes select weights into an fp64 sum in the same (ky, kx, ci) order (adding the 0.0 of a
// silent channel changes no fp64 sum: the same values bit for bit).  Plain PTC input [B][H][W][T][Cin], Cin % 16 == 0, Cout <= 4.
template <bool TRANSPOSED, int CO>
__global__ __launch_bounds__(256) void conv_raw_steps_kernel(FusedArgs a) {
  const int Cin = a.C0, T = a.T;
  const long long total = (long long)a.B * a.Ho * a.Wo * T;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    long long r = i / T;
    const int ox = (int)(r % a.Wo); r /= a.Wo;
    const int oy = (int)(r % a.Ho);
    const int b = (int)(r / a.Ho);
    double acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = a.bias ? (double)a.bias[co] : 0.0;
    for (int ky = 0; ky < a.k; ++ky) {
      int iy;
      if (TRANSPOSED) {
        const int ty = oy + a.pad - ky;
        if (ty < 0 || ty % a.stride) continue;
        iy = ty / a.stride;
      } else {
        iy = oy * a.stride - a.pad + ky;
      }
      if (iy < 0 || iy >= a.H) continue;
      for (int kx = 0; kx < a.k; ++kx) {
        int ix;
        if (TRANSPOSED) {
          const int tx = ox + a.pad - kx;
          if (tx < 0 || tx % a.stride) continue;
          ix = tx / a.stride;
        } else {
          ix = ox * a.stride - a.pad + kx;
        }
        if (ix < 0 || ix >= a.W) continue;
        const float* wp = a.wt + (long long)(ky * a.k + kx) * Cin * CO;
        const uint8_t* sp = reinterpret_cast<const uint8_t*>(a.in0) + ((((long long)b * a.H + iy) * a.W + ix) * T + t) * Cin;
        for (int c16 = 0; c16 < Cin; c16 += 16) {
          const uint4 s16 = *reinterpret_cast<const uint4*>(sp + c16);
          const unsigned w4[4] = {s16.x, s16.y, s16.z, s16.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (w4[q] == 0u) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if ((w4[q] >> (8 * e)) & 0xffu) {
                const int ci = c16 + 4 * q + e;
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] += (double)wp[(long long)ci * CO + co];
              }
            }
          }
        }
      }
    }
    const long long plane = (long long)a.Ho * a.Wo;
#pragma unroll
    for (int co = 0; co < CO; ++co)
      a.out_f32[(((long long)t * a.B + b) * CO + co) * plane + (long long)oy * a.Wo + ox] = (float)acc[co];
  }
}

template <int INKIND, bool TR>
int launch_mode(const FusedArgs& a, int mode, hipStream_t stream) {
  const long long total = (long long)a.B * a.Ho * a.Wo * a.Cout;
  dim3 g(grid_for(total)), blk(256);
  switch (mode) {
    case SPK_MODE_LIF: hipLaunchKernelGGL((conv_fused_kernel<INKIND, TR, SPK_MODE_LIF>), g, blk, 0, stream, a); break;
    case SPK_MODE_RAW: hipLaunchKernelGGL((conv_fused_kernel<INKIND, TR, SPK_MODE_RAW>), g, blk, 0, stream, a); break;
    case SPK_MODE_MEMOUT: hipLaunchKernelGGL((conv_fused_kernel<INKIND, TR, SPK_MODE_MEMOUT>), g, blk, 0, stream, a); break;
    case SPK_MODE_MEAN: hipLaunchKernelGGL((conv_fused_kernel<INKIND, TR, SPK_MODE_MEAN>), g, blk, 0, stream, a); break;
    default: return SPK_ERR_UNSUPPORTED;
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

}  // namespace

extern "C" int spk_lif_const_input_table(float* thresholds16, unsigned* patterns18) {
  if (!thresholds16 || !patterns18) return SPK_ERR_ARG;
  const unsigned thb[16] = SPK_LIF_CONST_TH_BITS, pat[18] = SPK_LIF_CONST_PATTERNS;
  for (int i = 0; i < 16; ++i) memcpy(&thresholds16[i], &thb[i], 4);
  for (int i = 0; i < 18; ++i) patterns18[i] = pat[i];
  return SPK_OK;
}

extern "C" int spk_conv_out_size(int in, int k, int stride, int pad, int transposed, int out_pad) {
  return transposed ? (in - 1) * stride - 2 * pad + k + out_pad : (in + 2 * pad - k) / stride + 1;
}

extern "C" int spk_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int k, int transposed,
                                    hipStream_t stream) {
  if (!w || !packed || Cout <= 0 || Cin <= 0 || k <= 0) return SPK_ERR_ARG;
  int total = Cout * Cin * k * k;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, packed, Cout, Cin, k * k,
                     transposed);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int Cin, int H,
                              int W, int Cout, int k, int stride, int pad, hipStream_t stream) {
  if (!x || !w || !y || M <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0) return SPK_ERR_ARG;
  int Ho = spk_conv_out_size(H, k, stride, pad, 0, 0), Wo = spk_conv_out_size(W, k, stride, pad, 0, 0);
  if (Ho <= 0 || Wo <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(conv_nchw_kernel<false>, dim3(grid_for(M * Cout * Ho * Wo)), dim3(256), 0, stream, x, w, bias, y,
                     M, Cin, H, W, Cout, Ho, Wo, k, stride, pad);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_conv_transpose2d_fwd(const float* x, const float* w, const float* bias, float* y, long long M,
                                        int Cin, int H, int W, int Cout, int k, int stride, int pad, int out_pad,
                                        hipStream_t stream) {
  if (!x || !w || !y || M <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0 || out_pad < 0)
    return SPK_ERR_ARG;
  int Ho = spk_conv_out_size(H, k, stride, pad, 1, out_pad), Wo = spk_conv_out_size(W, k, stride, pad, 1, out_pad);
  if (Ho <= 0 || Wo <= 0) return SPK_ERR_ARG;
  hipLaunchKernelGGL(conv_nchw_kernel<true>, dim3(grid_for(M * Cout * Ho * Wo)), dim3(256), 0, stream, x, w, bias, y,
                     M, Cin, H, W, Cout, Ho, Wo, k, stride, pad);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_conv_fused_fwd(const void* in0, const uint8_t* in1, int C0, int C1, int in_kind,
                                  const float* w_packed, const float* bias, const float* bn_a, const float* bn_b,
                                  float* v_inout, uint8_t* out_ptc, float* out_f32, float* out_pre, uint8_t* out_u8,
                                  const float* coef, int apply_tanh, int mode, int T, int B, int H, int W, int Cout,
                                  int k, int stride, int pad, int transposed, int out_pad, int chunk0, int chunk1,
                                  int chunk_out, uint8_t* out_counts, const int* n_dyn_or_null, hipStream_t stream) {
  if (!in0 || !w_packed || T <= 0 || T > SPK_MAX_T || B <= 0 || C0 <= 0 || C1 < 0 || Cout <= 0 || k <= 0 ||
      stride <= 0 || pad < 0)
    return SPK_ERR_ARG;
  if (in_kind < 0 || in_kind > 2) return SPK_ERR_ARG;
  if (in_kind == SPK_IN_PTC && ((C0 % 4) || (C1 % 4))) return SPK_ERR_UNSUPPORTED;   // u32 spike loads
  if (chunk0 <= 0) chunk0 = C0;
  if (chunk1 <= 0) chunk1 = C1 > 0 ? C1 : 4;
  const int out_c4 = chunk_out == SPK_CHUNK_C4 ? 64 : (chunk_out == SPK_CHUNK_S32 ? 32 : 0);
  if (out_c4 && (mode != SPK_MODE_LIF || !out_ptc || (Cout % out_c4))) return SPK_ERR_ARG;
  if (chunk_out <= 0) chunk_out = Cout;
  if (in_kind == SPK_IN_PTC && ((chunk0 % 4) || (C0 % chunk0) || (C1 > 0 && ((chunk1 % 4) || (C1 % chunk1)))))
    return SPK_ERR_ARG;
  if (Cout % chunk_out) return SPK_ERR_ARG;
  if (in_kind != SPK_IN_PTC && C1 != 0) return SPK_ERR_UNSUPPORTED;
  if (in_kind == SPK_IN_TINV && transposed) return SPK_ERR_UNSUPPORTED;
  if (C1 > 0 && !in1) return SPK_ERR_ARG;
  if (mode == SPK_MODE_LIF && (!bn_a || !bn_b || (!out_ptc && !out_f32))) return SPK_ERR_ARG;
  if ((mode == SPK_MODE_RAW || mode == SPK_MODE_MEAN) && !out_f32) return SPK_ERR_ARG;
  if (mode == SPK_MODE_MEMOUT && (!coef || (!out_f32 && !out_u8))) return SPK_ERR_ARG;
  FusedArgs a;
  a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.wt = w_packed; a.bias = bias; a.bn_a = bn_a; a.bn_b = bn_b;
  a.v_io = v_inout; a.out_ptc = out_ptc; a.out_f32 = out_f32; a.out_pre = out_pre; a.out_u8 = out_u8; a.coef = coef;
  a.chunk0 = chunk0; a.chunk1 = chunk1; a.chunk_out = chunk_out; a.out_cnt = out_counts; a.out_c4 = out_c4;
  a.n_dyn = n_dyn_or_null;
  if (out_counts && (mode != SPK_MODE_LIF || (Cout % 32))) return SPK_ERR_ARG;
  a.apply_tanh = apply_tanh; a.T = T; a.B = B; a.H = H; a.W = W; a.Cout = Cout;
  a.Ho = spk_conv_out_size(H, k, stride, pad, transposed, out_pad);
  a.Wo = spk_conv_out_size(W, k, stride, pad, transposed, out_pad);
  a.k = k; a.stride = stride; a.pad = pad;
  if (a.Ho <= 0 || a.Wo <= 0) return SPK_ERR_ARG;
  if (in_kind == SPK_IN_TINV && mode == SPK_MODE_LIF && T == 16 && out_ptc && !out_f32 && !out_pre && (Cout % 16) == 0 &&
      (256 % Cout) == 0 && (out_c4 || chunk_out == Cout) && (long long)B * a.Ho * a.Wo < (1ll << 30)) {
    TinvArgs t;
    t.x = reinterpret_cast<const float*>(in0); t.wt = w_packed; t.bias = bias; t.bn_a = bn_a; t.bn_b = bn_b; t.v_io = v_inout;
    t.out = out_ptc; t.out_cnt = out_counts; t.B = B; t.Cin = C0; t.H = H; t.W = W; t.Cout = Cout; t.Ho = a.Ho; t.Wo = a.Wo;
    t.k = k; t.stride = stride; t.pad = pad; t.out_c4 = out_c4; t.n_dyn = n_dyn_or_null;
    const long long steps = ((long long)B * a.Ho * a.Wo + (256 / Cout) - 1) / (256 / Cout);
    const long long cap = 256 * 16;
    const dim3 tg((unsigned)(steps < cap ? steps : cap)), tb(256);
#ifndef SPK_TINV_STAGED
#define SPK_TINV_STAGED 1       // 1: stateless 3x3 calls take tinv_lif_staged_kernel (per-position work once per position, pixels through LDS)
#endif
    const long long chunks = ((long long)B * a.Ho * a.Wo + SPK_TINV_PB - 1) / SPK_TINV_PB;
    const dim3 sg((unsigned)(chunks < 256 * 8 ? chunks : 256 * 8));
    if (SPK_TINV_STAGED && k == 3 && C0 == 1 && !v_inout) hipLaunchKernelGGL((tinv_lif_staged_kernel<3, 1>), sg, tb, 0, stream, t);
    else if (SPK_TINV_STAGED && k == 3 && C0 == 3 && !v_inout) hipLaunchKernelGGL((tinv_lif_staged_kernel<3, 3>), sg, tb, 0, stream, t);
    else if (SPK_TINV_STAGED && k == 3 && C0 == 2 && !v_inout) hipLaunchKernelGGL((tinv_lif_staged_kernel<3, 2>), sg, tb, 0, stream, t);
    else if (SPK_TINV_STAGED && k == 1 && C0 == 16 && !v_inout) hipLaunchKernelGGL((tinv_lif_staged_kernel<1, 16>), sg, tb, 0, stream, t);   // spike generator
    else if (k == 3 && C0 == 1) hipLaunchKernelGGL((tinv_lif_kernel<3, 1>), tg, tb, 0, stream, t);       // encoder conv1
    else if (k == 3 && C0 == 3) hipLaunchKernelGGL((tinv_lif_kernel<3, 3>), tg, tb, 0, stream, t);       // ... on RGB
    else if (k == 3 && C0 == 2) hipLaunchKernelGGL((tinv_lif_kernel<3, 2>), tg, tb, 0, stream, t);       // denoiser conv1
    else hipLaunchKernelGGL((tinv_lif_kernel<0, 0>), tg, tb, 0, stream, t);                               // (spike generator: 1x1, 16 channels -- measured faster with the weights left in L1)
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  if (in_kind == SPK_IN_PTC && mode == SPK_MODE_RAW && Cout <= 4 && Cout != 2 && C1 == 0 && chunk0 == C0 && (C0 % 16) == 0 && !n_dyn_or_null) {
    // few output channels, per-step output (the decoder's read-out layer through the module API): one thread per (position, step)
    const long long total = (long long)B * a.Ho * a.Wo * T;
    const dim3 g(grid_for(total)), blk(256);
#define SPK_RAW_STEPS(TR_, CO_) hipLaunchKernelGGL((conv_raw_steps_kernel<TR_, CO_>), g, blk, 0, stream, a)
    if (transposed) { if (Cout == 1) SPK_RAW_STEPS(true, 1); else if (Cout == 3) SPK_RAW_STEPS(true, 3); else SPK_RAW_STEPS(true, 4); }
    else            { if (Cout == 1) SPK_RAW_STEPS(false, 1); else if (Cout == 3) SPK_RAW_STEPS(false, 3); else SPK_RAW_STEPS(false, 4); }
#undef SPK_RAW_STEPS
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  if (in_kind == SPK_IN_TINV) return launch_mode<SPK_IN_TINV, false>(a, mode, stream);
  if (in_kind == SPK_IN_SEQ)
    return transposed ? launch_mode<SPK_IN_SEQ, true>(a, mode, stream) : launch_mode<SPK_IN_SEQ, false>(a, mode, stream);
  return transposed ? launch_mode<SPK_IN_PTC, true>(a, mode, stream) : launch_mode<SPK_IN_PTC, false>(a, mode, stream);
}

extern "C" long long spk_spikegen_table_bytes(int K, int Cout) {
  if (K <= 0 || (Cout != 16 && Cout != 32)) return -1;
  return (long long)(K + 1) * Cout * 2;
}

extern "C" int spk_spikegen_tokens_s32(const long long* tokens, const float* codebook, const float* w_packed, const float* bias,
                                       const float* bn_a, const float* bn_b, unsigned short* table_ws, int build_table,
                                       uint8_t* out_s32, int T, long long n_positions, int K, int D, int Cout, hipStream_t stream) {
  if (!tokens || !codebook || !w_packed || !bn_a || !bn_b || !table_ws || !out_s32 || n_positions <= 0 || K <= 0 || D <= 0)
    return SPK_ERR_ARG;
  if (T != 16 || (Cout != 16 && Cout != 32)) return SPK_ERR_UNSUPPORTED;
  if (build_table) {                                          // (0: table_ws still holds the table of these weights and this codebook)
    hipLaunchKernelGGL(spikegen_table_kernel, dim3(((K + 1) * Cout + 255) / 256), dim3(256), 0, stream, codebook, w_packed, bias, bn_a, bn_b,
                       table_ws, K, D, Cout);
    SPK_LAUNCH_CHECK();
  }
  const long long blocks = (n_positions * 16 + 255) / 256;
  const dim3 g((unsigned)(blocks < 256 * 32 ? blocks : 256 * 32));
  if (Cout == 16) hipLaunchKernelGGL((spikegen_expand_kernel<16>), g, dim3(256), 0, stream, tokens, table_ws, out_s32, n_positions, K);
  else hipLaunchKernelGGL((spikegen_expand_kernel<32>), g, dim3(256), 0, stream, tokens, table_ws, out_s32, n_positions, K);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
