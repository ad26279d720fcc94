// Training convolutions of the spiking VQ-VAE (R/snn_model/vae_model.py:101-159: stride-2 Conv2d, 1x1 Conv2d, stride-2 and
// stride-1 ConvTranspose2d; what R/main.py:118-146 runs through cuDNN forward and backward): forward, data gradient and weight
// gradient of channels-last fp32 tensors on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: fp32 products, fp32 accumulation --
// the arithmetic of an fp32 GEMM, no operand is narrowed), SURVEY.md §8 f2.
//
// Every one of the four data-path operators is ONE of two gathers over a "class grid" q:
//   form 0 (regular):     out[n, q, co] = sum over (tap, c) of in[n, q * s + (k - pad), c] * W[tap][c][co]
//       = Conv2d forward, ConvTranspose2d data gradient
//   form 1 (transposed):  out[n, s * q + p, co] = sum over the taps k with (p + pad - k) % s == 0 and c of
//                                                 in[n, q + (p + pad - k) / s, c] * W[tap][c][co]      (p = sub-pixel class)
//       = ConvTranspose2d forward, Conv2d data gradient
// with the weight tensor addressed through three strides (tap, reduced channel, output channel), so that Conv2d's
// [Cout][k][k][Cin] and ConvTranspose2d's [Cin][k][k][Cout] storage (channels-last parameters, spkdiff/fused.py) and both
// transpositions are the same code.  A workgroup (eight waves) stages ALL k * k weight taps of its column tiles in LDS once, in
// B-fragment order ([tap][c / 8][c / 4 % 2][co][c % 4]: one 16-byte read = the B operands of four MFMAs; the class / tap tables come
// from the host as a kernel argument), and its waves walk items of 32 output positions of one sub-pixel class: a lane reads 16
// bytes of its row's input record per eight reduced channels (the A operands of the same four MFMAs; the K order inside a group
// of eight is a permutation both operands share), the next tap's records requested before this tap's MFMAs.  Rows of a tile share
// the tap list: no multiplication by structural zeros in the transposed form.
//
// Weight gradient: D[tap][cu][cv] = sum over (n, q) of U[n, q * s - pad + k][cu] * V[n, q][cv]  (Conv2d: U = input, V = gy;
// ConvTranspose2d: U = gy, V = input -- the tensor on the finer grid is U).  Rows = cv, columns = cu, K = positions; the 32x32 tiles
// (tap, cu / 32, cv / 32) are dealt to four waves, a workgroup owns a range of coarse-grid rows and stages their operand rows through
// LDS a block ahead (conv_train_wgrad_lds_kernel; the direct-from-global form conv_train_wgrad_kernel takes the shapes whose rows do not
// fit), two halves of four waves whose tiles are added through LDS, and a second launch adds the workgroups' partial tiles in index
// order (deterministic) and scatters them through the gradient's strides; the bias gradient (a column sum of gy) comes from the staged
// rows.
//
// Sides of one to four channels (the 1 -> 32 / 3 -> 32 first layer, the 32 -> 1 / 32 -> 3 read-out layer) are vector kernels: there is no
// matrix in them.  Every variant that was built and measured slower stays behind a knob below, with its numbers
// (profiles/r5_ab_conv_train.txt holds the A/B records, ablations, counters and phase stamps).
#include "spk_common.h"
#include "../../include/spkdiff.h"
#include <mutex>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

#ifndef CT_STAGE_A
#define CT_STAGE_A 0                               // 1: the gather kernel's input records go through a per-wave LDS transpose (coalesced 16-byte
#endif                                             // requests: LPR lanes per row instead of one).  Built because "lane = row" touches 32 - 64 cache
                                                   // lines per request; measured SLOWER (same box, N = 512: dec.convT2 forward 80.3 against 66.0 us,
                                                   // sum of the ten matrix-path launches of a VQ-VAE iteration 289 against 265 us,
                                                   // profiles/r5_ab_conv_train.txt): the LDS tiles cost a workgroup per CU, and the address unit
                                                   // was not what set the pace (the no-load ablation, CT_DBG 4, runs the same launch in 72 us)
#ifndef CT_WGRAD_LDS
#define CT_WGRAD_LDS 1                             // 0: the weight gradient always reads its operands from global memory (the first form)
#endif
#ifndef CT_C1_ROWS_CAP
#define CT_C1_ROWS_CAP 4096                        // workgroups of the few-channel gather kernels (at most; each walks output rows)
#endif
#ifndef CT_C1W_CAP
#define CT_C1W_CAP 1024                            // workgroups of the few-channel weight gradient (at most)
#endif
#ifndef CT_WL_FUSE_BIAS
#define CT_WL_FUSE_BIAS 1                          // LDS-staged weight gradient: bias column sums from the staged rows (0: a second pass over gy)
#endif
#ifndef CT_WL_PINGPONG
#define CT_WL_PINGPONG 0                           // 1: LDS-staged weight gradient with the two halves of a workgroup alternating between their MFMA
#endif                                             // phase and their request / LDS-write phase (one buffer per half, two barriers per block).  Built
                                                   // because the in-step form's parts add up; measured SLOWER (same box: dec.convT2 109.5 against 87.2 us,
                                                   // enc.conv2 46.8 against 40.6, dec.convT1 53.9 against 45.7): a wave alone on its SIMD does not
                                                   // issue its MFMAs twice as fast -- every step waits for its own LDS reads with nobody to fill in
#ifndef CT_SUBPIXEL
#define CT_SUBPIXEL 0                              // 1: the stride-2 transposed form takes ALL FOUR sub-pixel classes of 32 coarse positions per item
                                                   // (conv_train_gather_sub_kernel).  Built to amortise the per-item index arithmetic; measured SLOWER
                                                   // (same box, us per call: dec.convT2 forward 69.1 against 64.4, enc.conv2 data gradient 35.6 against
                                                   // 29.6, dec.convT1 forward 33.0 against 23.4): 182 registers put one workgroup on a CU instead of two
#endif
#ifndef CT_BIG_ITEMS
#define CT_BIG_ITEMS 2048                          // a wave takes all column tiles of its 32 rows from this many items on (below: one column tile)
#endif
#ifndef CT_DBG
#define CT_DBG 0                                   // timing experiments only (results are wrong; 32 / 64 / 128: the LDS-staged weight gradient without
                                                   // its bias sums / MFMAs / global requests): 1 wgrad without operand loads, 2 wgrad without
#endif                                             // MFMAs, 4 gather without input loads, 8 gather without MFMAs, 16 gather without stores
constexpr int MAX_TAPS = 16;                       // k <= 4
constexpr int GATHER_LDS_MAX = 150 * 1024;

struct GArgs {
  const float* in; const float* w; const float* bias; float* out;
  int N, Hi, Wi, Cred, Ho, Wo, Cout;
  int k, stride, pad, form;
  long long w_tap, w_red, w_out;
};

// taps of sub-pixel class (py, px): index into the k x k kernel and the input offset of the tap
struct TapList { int n; int tap[MAX_TAPS], dy[MAX_TAPS], dx[MAX_TAPS]; };

__host__ __device__ inline void build_taps(TapList& tl, int k, int stride, int pad, int form, int py, int px) {
  int n = 0;
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) {
      int dy, dx;
      bool ok = true;
      if (!form) { dy = ky - pad; dx = kx - pad; }
      else {
        const int ty = py + pad - ky, tx = px + pad - kx;
        ok = (ty % stride == 0) && (tx % stride == 0);
        dy = ty / stride; dx = tx / stride;
      }
      if (ok && n < MAX_TAPS) { tl.tap[n] = ky * k + kx; tl.dy[n] = dy; tl.dx[n] = dx; ++n; }
    }
  tl.n = n;
}

// All k * k weight taps of the workgroup's CN column tiles into LDS in B-fragment order ([tap][c / 8][c / 4 % 2][co][c % 4]).
template <int J, int CN, int NTH>
__device__ __forceinline__ void stage_weights(const GArgs& a, float4* sB, const int co0, const int tid) {
  constexpr int CP = CN * 32, CRED = J * 8;
  const int ntap = a.k * a.k;
  if (CT_DBG & 256) return;                        // (timing experiment: no weight staging)
  // staging: 16-byte loads along whichever weight dimension is contiguous, SB requests in flight per thread (the weights are
  // L2-resident; one request at a time cost 30+ us per launch)
  constexpr int SB = 6;
  const bool al = ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0) && (a.w_tap & 3) == 0;
  if (al && a.w_red == 1 && (a.w_out & 3) == 0) {
    // four consecutive reduced channels = one LDS slot
    const int total = ntap * CP * (CRED / 4);
    for (int e0 = tid; e0 < total; e0 += NTH * SB) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * NTH;
        const int c4 = e % (CRED / 4), r = e / (CRED / 4), co = r % CP, t = r / CP;
        v[u] = (e < total && co0 + co < a.Cout) ? *reinterpret_cast<const float4*>(a.w + t * a.w_tap + (co0 + co) * a.w_out + 4 * c4)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * NTH;
        const int c4 = e % (CRED / 4), r = e / (CRED / 4), co = r % CP, t = r / CP;
        if (e < total) sB[((t * J + (c4 >> 1)) * 2 + (c4 & 1)) * CP + co] = v[u];
      }
    }
  } else if (al && a.w_out == 1 && (a.w_red & 3) == 0 && (a.Cout & 3) == 0) {
    // four consecutive output channels = component c % 4 of four neighbouring LDS slots
    const int total = ntap * CRED * (CP / 4);
    for (int e0 = tid; e0 < total; e0 += NTH * SB) {
      float4 v[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * NTH;
        const int q4 = e % (CP / 4), r = e / (CP / 4), c = r % CRED, t = r / CRED;
        v[u] = (e < total && co0 + 4 * q4 < a.Cout) ? *reinterpret_cast<const float4*>(a.w + t * a.w_tap + c * a.w_red + co0 + 4 * q4)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * NTH;
        const int q4 = e % (CP / 4), r = e / (CP / 4), c = r % CRED, t = r / CRED;
        if (e < total) {
          float* d = reinterpret_cast<float*>(sB) + ((((t * J + (c >> 3)) * 2 + ((c >> 2) & 1)) * CP + 4 * q4) << 2) + (c & 3);
          d[0] = v[u].x; d[4] = v[u].y; d[8] = v[u].z; d[12] = v[u].w;
        }
      }
    }
  } else {
    for (int e = tid; e < ntap * CRED * CP; e += NTH) {
      const int co = e % CP, r = e / CP, c = r % CRED, t = r / CRED;
      const float v = co0 + co < a.Cout ? a.w[t * a.w_tap + c * a.w_red + (co0 + co) * a.w_out] : 0.f;
      reinterpret_cast<float*>(sB)[((((t * J + (c >> 3)) * 2 + ((c >> 2) & 1)) * CP + co) << 2) + (c & 3)] = v;
    }
  }
}

// J = Cred / 8, CN = column tiles (32 output channels each) of a wave, RM = row tiles (32 output positions each) of a wave.
// blockIdx.y = first column tile of the workgroup (the host splits the column tiles over workgroups when the layer has too few
// rows to fill the chip otherwise).  A workgroup stages ALL k * k taps once and walks (sub-pixel class, row group) items: the
// classes of the transposed form have 1 .. ceil(k / s)^2 taps each, so a workgroup per class would leave the chip waiting for
// the largest class.
constexpr int MAX_CLASSES = 4;                      // (transposed form: stride <= 2; the table travels as a kernel argument)
struct ClassTab { int ncls, cs; int first[MAX_CLASSES + 1]; int Qh[MAX_CLASSES], Qw[MAX_CLASSES]; TapList tl[MAX_CLASSES]; };

// (the class / tap tables are built on the host: one thread building them in front of a workgroup barrier took 5.6 us of a 55 us launch)
inline void build_class_tab(ClassTab& ct, const GArgs& a, int rows) {
  const int cs = a.form ? a.stride : 1;
  ct.ncls = cs * cs; ct.cs = cs;
  int first = 0;
  for (int c = 0; c < cs * cs; ++c) {
    const int py = c / cs, px = c % cs;
    build_taps(ct.tl[c], a.k, a.stride, a.pad, a.form, py, px);
    ct.Qh[c] = (a.Ho - py + cs - 1) / cs;
    ct.Qw[c] = (a.Wo - px + cs - 1) / cs;
    ct.first[c] = first;
    first += (int)(((long long)a.N * ct.Qh[c] * ct.Qw[c] + rows - 1) / rows);
  }
  for (int c = cs * cs; c <= MAX_CLASSES; ++c) ct.first[c] = first;
}

template <int J, int CN, int RM>
__global__ __launch_bounds__(512) void conv_train_gather_kernel(GArgs a, ClassTab ct_arg) {
  constexpr int NTH = 512, NWV = 8;                 // eight waves share one staged weight image
  extern __shared__ __attribute__((aligned(16))) float4 sB[];       // [tap][J][2][CN * 32]
  __shared__ ClassTab ct;                           // (LDS copy of the host-built table: published by the barrier behind the weight staging)
  if (threadIdx.x < sizeof(ClassTab) / 4) reinterpret_cast<int*>(&ct)[threadIdx.x] = reinterpret_cast<const int*>(&ct_arg)[threadIdx.x];
  constexpr int CP = CN * 32, CRED = J * 8, ROWS = RM * 32;
  constexpr bool STAGE = CT_STAGE_A && (J == 1 || J == 2 || J == 4 || J == 8);       // (a power of two lanes per row)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cs = a.form ? a.stride : 1;            // output step per class-grid step
  const int ai = a.form ? 1 : a.stride;            // input step per class-grid step
  const int co0 = blockIdx.y * CP;                 // first output channel of this workgroup
  unsigned long long stamp[4] = {0, 0, 0, 0};      // (CT_DBG 512: s_memrealtime at start / tables built / weights staged / items done -> a.out)
  if (CT_DBG & 512) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const int ntap = a.k * a.k;
  float4* const sA = sB + ntap * J * 2 * CP;          // STAGE: [8 waves][32 rows][Cred / 4] operand tiles behind the weights
  if (CT_DBG & 512) stamp[1] = __builtin_amdgcn_s_memrealtime();
  stage_weights<J, CN, NTH>(a, sB, co0, tid);
  __syncthreads();
  if (CT_DBG & 512) stamp[2] = __builtin_amdgcn_s_memrealtime();

  const int nitems = ct.first[cs * cs];
  const int r = lane & 31, h = lane >> 5;
  for (int item = blockIdx.x * NWV + wave; item < nitems; item += gridDim.x * NWV) {
    int cls = 0;
    while (item >= ct.first[cls + 1]) ++cls;
    const TapList& tl = ct.tl[cls];
    const int py = cls / cs, px = cls % cs;
    const int Qh = ct.Qh[cls], Qw = ct.Qw[cls];
    const long long M = (long long)a.N * Qh * Qw;
    const int gi = item - ct.first[cls];
    const int nt = tl.n;
    int n_[RM], qy_[RM], qx_[RM], opos[RM];
    bool rv[RM];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      const long long R = (long long)gi * ROWS + rm * 32 + r;
      rv[rm] = R < M;
      const int Rc = rv[rm] ? (int)R : 0;
      qx_[rm] = Rc % Qw;
      const int t = Rc / Qw;
      qy_[rm] = t % Qh;
      n_[rm] = t / Qh;
      opos[rm] = rv[rm] ? (n_[rm] * a.Ho + qy_[rm] * cs + py) * a.Wo + qx_[rm] * cs + px : -1;
    }
    v16f acc[RM][CN];
#pragma unroll
    for (int rm = 0; rm < RM; ++rm)
#pragma unroll
      for (int cn = 0; cn < CN; ++cn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[rm][cn][i] = 0.f;
    float4 A[2][RM][J];                             // [buffer][row tile][j]: the next tap's records are requested before this tap's MFMAs
    auto load_a = [&](int t, float4 (&dst)[RM][J]) {
      const int dy = tl.dy[t], dx = tl.dx[t];
#pragma unroll
      for (int rm = 0; rm < RM; ++rm) {
        const int iy = qy_[rm] * ai + dy, ix = qx_[rm] * ai + dx;
        const bool ok = rv[rm] && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const float4* p = reinterpret_cast<const float4*>(a.in + (((long long)n_[rm] * a.Hi + iy) * a.Wi + ix) * CRED + 4 * h);
#pragma unroll
        for (int j = 0; j < J; ++j) dst[rm][j] = ok ? p[2 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto mma = [&](int t, const float4 (&src)[RM][J]) {
      const int tap = tl.tap[t];
      // (reading the weight fragments of step j + 1 in front of the MFMAs of step j behind scheduling barriers was measured SLOWER --
      //  dec.convT2 forward 78 against 61 us: the barriers also pin the input requests; left to hipcc)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        float4 B[CN];
#pragma unroll
        for (int cn = 0; cn < CN; ++cn) B[cn] = sB[((tap * J + j) * 2 + h) * CP + cn * 32 + r];
        // consecutive MFMAs write different accumulators
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int rm = 0; rm < RM; ++rm)
#pragma unroll
            for (int cn = 0; cn < CN; ++cn) {
              const float av = q == 0 ? src[rm][j].x : q == 1 ? src[rm][j].y : q == 2 ? src[rm][j].z : src[rm][j].w;
              const float bv = q == 0 ? B[cn].x : q == 1 ? B[cn].y : q == 2 ? B[cn].z : B[cn].w;
              if (CT_DBG & 8) acc[rm][cn][0] += av + bv;
              else acc[rm][cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[rm][cn], 0, 0, 0);
            }
      }
    };
    if constexpr (STAGE) {
      // Coalesced requests + a per-wave LDS transpose.  With "lane = row" a 16-byte request per lane touches 32 - 64 cache lines
      // per instruction and the CU's one texture-address unit, not the matrix pipe, set the pace (measured: 2x the MFMA time).
      // Here LPR = Cred / 4 consecutive lanes cover one row's record (8 full lines per instruction), the wave writes the 32 rows
      // to its own LDS tile (chunks XOR-swizzled by row: conflict-free both ways) and reads its operands back row-wise.  LDS
      // operations of a wave execute in order; the asm statements only keep the compiler from reordering across them.
      static_assert(RM == 1, "staged form: one row tile per wave");
      constexpr int LPR = 2 * J, RPI = 64 / LPR, SH = LPR >= 8 ? 0 : (LPR == 4 ? 1 : 2), MASK = LPR >= 8 ? 7 : LPR - 1;
      float4* const sAw = sA + wave * (32 * LPR);
      const int wrow = lane / LPR, wchunk = lane % LPR;
      float4 G[J];
      auto gload = [&](int t) {
        const int iy = qy_[0] * ai + tl.dy[t], ix = qx_[0] * ai + tl.dx[t];
        const bool ok = rv[0] && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        const int off = ok ? ((n_[0] * a.Hi + iy) * a.Wi + ix) * CRED : -1;
#pragma unroll
        for (int i = 0; i < J; ++i) {
          const int o = __shfl(off, i * RPI + wrow);
          if (CT_DBG & 4) G[i] = make_float4((float)o, 1.f, 2.f, (float)lane);
          else
          G[i] = o >= 0 ? *reinterpret_cast<const float4*>(a.in + o + 4 * wchunk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto stage = [&]() {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < J; ++i) {
          const int row = i * RPI + wrow;
          sAw[row * LPR + (wchunk ^ ((row >> SH) & MASK))] = G[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < J; ++j) A[0][0][j] = sAw[r * LPR + ((2 * j + h) ^ ((r >> SH) & MASK))];
        asm volatile("" ::: "memory");
      };
      if (nt > 0) gload(0);
      for (int t = 0; t < nt; ++t) {
        stage();
        if (t + 1 < nt) gload(t + 1);
        mma(t, A[0]);
      }
    } else {
      if (nt > 0) load_a(0, A[0]);
      int t = 0;
      for (; t + 1 < nt; t += 2) {
        load_a(t + 1, A[1]);
        mma(t, A[0]);
        if (t + 2 < nt) load_a(t + 2, A[0]);
        mma(t + 1, A[1]);
      }
      if (t < nt) mma(t, A[0]);
    }

    // D[i][j]: j = lane & 31, i = 8 * (reg / 4) + 4 * (lane / 32) + reg % 4
#pragma unroll
    for (int rm = 0; rm < RM; ++rm) {
      int orow[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) orow[i] = __shfl(opos[rm], 8 * (i >> 2) + 4 * h + (i & 3));
#pragma unroll
      for (int cn = 0; cn < CN; ++cn) {
        const int co = co0 + cn * 32 + r;
        const float bv = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        if (co < a.Cout) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (orow[i] >= 0 && (!(CT_DBG & 16) || acc[rm][cn][i] == 12345.f)) a.out[(long long)orow[i] * a.Cout + co] = acc[rm][cn][i] + bv;
        }
      }
    }
  }
  if (CT_DBG & 512) {
    __syncthreads();
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.out) + 4 * (blockIdx.y * gridDim.x + blockIdx.x);
      o[0] = stamp[0]; o[1] = stamp[1]; o[2] = stamp[2]; o[3] = stamp[3];
    }
  }
}

// The transposed form at stride 2 with ALL FOUR sub-pixel classes of a coarse position in one item.  Class (py, px) of coarse
// position q reads in[q + d] for the taps with (p + pad - k) even, d = (p + pad - k) / 2: over the four classes the k * k taps meet
// only a handful of DISTINCT offsets d (four for k = 3: every input record feeds one tap of up to four classes).  An item = 32 coarse
// positions: per distinct offset one 16-byte-per-lane read of the input records and, for every class that has a tap there, the
// 4 J MFMAs of that tap into the class's accumulator -- 9 taps x 4 J MFMAs per four input tiles and ONE set of index arithmetic,
// where the class-per-item form above spends a tile, a prologue and an epilogue per 1 - 4 taps (550 vector instructions per 72
// MFMAs: 28 % of the matrix pipe, profiles/r5_ab_conv_train.txt (2)).
constexpr int SUB_MAXOFF = 9;
struct SubTab { int noff; int dy[SUB_MAXOFF], dx[SUB_MAXOFF]; int tap[SUB_MAXOFF][4]; };

inline void build_sub_tab(SubTab& tb, const GArgs& a) {
  int n = 0;
  for (int c = 0; c < 4; ++c) {
    const int py = c >> 1, px = c & 1;
    for (int ky = 0; ky < a.k; ++ky)
      for (int kx = 0; kx < a.k; ++kx) {
        const int ty = py + a.pad - ky, tx = px + a.pad - kx;
        if ((ty & 1) || (tx & 1)) continue;
        const int dy = ty / 2, dx = tx / 2;
        int o = 0;
        while (o < n && !(tb.dy[o] == dy && tb.dx[o] == dx)) ++o;
        if (o == n) {
          tb.dy[n] = dy; tb.dx[n] = dx;
          for (int q = 0; q < 4; ++q) tb.tap[n][q] = -1;
          ++n;
        }
        tb.tap[o][c] = ky * a.k + kx;
      }
  }
  tb.noff = n;
}

template <int J, int CN>
__global__ __launch_bounds__(512) void conv_train_gather_sub_kernel(GArgs a, SubTab tb) {
  constexpr int NTH = 512, NWV = 8;
  extern __shared__ __attribute__((aligned(16))) float4 sB[];       // [tap][J][2][CN * 32]
  constexpr int CP = CN * 32, CRED = J * 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.y * CP;
  stage_weights<J, CN, NTH>(a, sB, co0, tid);
  __syncthreads();

  const int Qh = (a.Ho + 1) >> 1, Qw = (a.Wo + 1) >> 1;
  const long long M = (long long)a.N * Qh * Qw;
  const int nitems = (int)((M + 31) / 32);
  const int r = lane & 31, h = lane >> 5;
  const int noff = tb.noff;
  for (int item = blockIdx.x * NWV + wave; item < nitems; item += gridDim.x * NWV) {
    const long long R = (long long)item * 32 + r;
    const bool rv = R < M;
    const int Rc = rv ? (int)R : 0;
    const int qx = Rc % Qw, t1 = Rc / Qw, qy = t1 % Qh, n = t1 / Qh;
    const int opos = rv ? (n * a.Ho + 2 * qy) * a.Wo + 2 * qx : -1;           // class (0, 0); class (py, px) is py * Wo + px further
    const int edge = (2 * qy + 1 < a.Ho ? 1 : 0) | (2 * qx + 1 < a.Wo ? 2 : 0);   // bit 0: the odd row exists, bit 1: the odd column
    v16f acc[4][CN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int cn = 0; cn < CN; ++cn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][cn][i] = 0.f;
    float4 A[2][J];
    auto load_a = [&](int o, float4 (&dst)[J]) {
      const int iy = qy + tb.dy[o], ix = qx + tb.dx[o];
      const bool ok = rv && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
      const float4* p = reinterpret_cast<const float4*>(a.in + (((long long)n * a.Hi + iy) * a.Wi + ix) * CRED + 4 * h);
#pragma unroll
      for (int j = 0; j < J; ++j) dst[j] = ok ? p[2 * j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto mma = [&](int o, const float4 (&src)[J]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int tap = tb.tap[o][c];                  // (wave-uniform)
        if (tap < 0) continue;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          float4 B[CN];
#pragma unroll
          for (int cn = 0; cn < CN; ++cn) B[cn] = sB[((tap * J + j) * 2 + h) * CP + cn * 32 + r];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int cn = 0; cn < CN; ++cn) {
              const float av = q == 0 ? src[j].x : q == 1 ? src[j].y : q == 2 ? src[j].z : src[j].w;
              const float bv = q == 0 ? B[cn].x : q == 1 ? B[cn].y : q == 2 ? B[cn].z : B[cn].w;
              acc[c][cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c][cn], 0, 0, 0);
            }
        }
      }
    };
    if (noff > 0) load_a(0, A[0]);
    int o = 0;
    for (; o + 1 < noff; o += 2) {
      load_a(o + 1, A[1]);
      mma(o, A[0]);
      if (o + 2 < noff) load_a(o + 2, A[0]);
      mma(o + 1, A[1]);
    }
    if (o < noff) mma(o, A[0]);

    int orow[16], oedge[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int src = 8 * (i >> 2) + 4 * h + (i & 3);
      orow[i] = __shfl(opos, src);
      oedge[i] = __shfl(edge, src);
    }
#pragma unroll
    for (int cn = 0; cn < CN; ++cn) {
      const int co = co0 + cn * 32 + r;
      if (co < a.Cout) {
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int need = (c >> 1) | ((c & 1) << 1);                 // the edge bits class c needs
          float* const oc = a.out + ((long long)(c >> 1) * a.Wo + (c & 1)) * a.Cout + co;
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (orow[i] >= 0 && (oedge[i] & need) == need) oc[(long long)orow[i] * a.Cout] = acc[c][cn][i] + bv;
        }
      }
    }
  }
}

// ---- one to four output channels (the 32 -> 1 read-out layer; 32 -> 3 on RGB): LP = Cred / 4 lanes per output position ----
// (The two one-channel gather kernels walk ROWS of the output grid -- the row's image and y are wave-uniform, a thread's x and channel
//  quad come from its index by a shift or one small division -- and read every tap UNCONDITIONALLY from a clamped address, with a
//  select: a load under a per-lane condition makes hipcc branch around it and wait for it alone, k * k round trips one after the
//  other.  Measured, kernel time under rocprofv3: the read-out forward 34.9 -> 29.6 us, the one-reduced-channel form unchanged.)
template <int LP, int CO>
__global__ __launch_bounds__(256) void conv_train_c1out_kernel(GArgs a) {
  __shared__ float4 sW[MAX_TAPS * CO * LP];        // [tap][co][Cred / 4]
  const int tid = threadIdx.x;
  const int nt = a.k * a.k;
  for (int e = tid; e < nt * CO * LP * 4; e += 256) {
    const int c = e % (LP * 4), r = e / (LP * 4), co = r % CO, t = r / CO;
    reinterpret_cast<float*>(sW)[e] = a.w[t * a.w_tap + c * a.w_red + co * a.w_out];
  }
  __syncthreads();
  constexpr int PER = 256 / LP;                     // positions of a row per pass
  const int cq = tid % LP, xl = tid / LP;
  float b0[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) b0[co] = a.bias ? a.bias[co] : 0.f;
  const int sgn = a.form ? -1 : 1, ai = a.form ? 1 : a.stride;
  const int rows = a.N * a.Ho;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / a.Ho, oy = row - n * a.Ho;  // (wave-uniform)
    const float* inb = a.in + (long long)n * a.Hi * a.Wi * (LP * 4) + 4 * cq;
    for (int x0 = 0; x0 < a.Wo; x0 += PER) {        // (every lane of a position group takes part in the exchange below)
      const int ox = x0 + xl;
      const bool pv = ox < a.Wo;
      float acc[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[co] = 0.f;
      for (int ky = 0; ky < a.k; ++ky) {
        const int iy = oy * ai + sgn * (ky - a.pad);
        const bool yok = iy >= 0 && iy < a.Hi;     // (uniform)
        const int iyc = yok ? iy : 0;
#pragma unroll 3
        for (int kx = 0; kx < a.k; ++kx) {
          const int ix = ox * ai + sgn * (kx - a.pad);
          const bool ok = pv && yok && ix >= 0 && ix < a.Wi;
          const int ixc = ok ? ix : 0;
          const float4 ld = *reinterpret_cast<const float4*>(inb + ((long long)iyc * a.Wi + ixc) * (LP * 4));
          const float4 x = ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            const float4 w = sW[((ky * a.k + kx) * CO + co) * LP + cq];
            acc[co] = fmaf(x.x, w.x, acc[co]); acc[co] = fmaf(x.y, w.y, acc[co]);
            acc[co] = fmaf(x.z, w.z, acc[co]); acc[co] = fmaf(x.w, w.w, acc[co]);
          }
        }
      }
#pragma unroll
      for (int co = 0; co < CO; ++co) {
#pragma unroll
        for (int m = LP / 2; m >= 1; m >>= 1) acc[co] += __shfl_xor(acc[co], m);
        if (pv && cq == 0) a.out[((long long)row * a.Wo + ox) * CO + co] = acc[co] + b0[co];
      }
    }
  }
}

// ---- one to four reduced channels (the 1 -> 32 / 3 -> 32 first layer forward, the read-out layer's data gradient): regular form ----
template <int CR>
__global__ __launch_bounds__(256) void conv_train_c1in_kernel(GArgs a) {
  extern __shared__ float sW1[];                    // [tap][ci][Cout]
  const int tid = threadIdx.x, C = a.Cout, CQ = C >> 2;
  const int nt = a.k * a.k;
  for (int e = tid; e < nt * CR * C; e += 256) {
    const int c = e % C, r = e / C, ci = r % CR, t = r / CR;
    sW1[e] = a.w[t * a.w_tap + ci * a.w_red + c * a.w_out];
  }
  __syncthreads();
  const int rows = a.N * a.Ho, per_row = a.Wo * CQ;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / a.Ho, oy = row - n * a.Ho;  // (wave-uniform)
    const float* inb = a.in + (long long)n * a.Hi * a.Wi * CR;
    for (int idx = tid; idx < per_row; idx += 256) {
      const int ox = idx / CQ, cq = idx - ox * CQ;
      float4 acc = a.bias ? *reinterpret_cast<const float4*>(a.bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int ky = 0; ky < a.k; ++ky) {
        const int iy = oy * a.stride + ky - a.pad;
        const bool yok = iy >= 0 && iy < a.Hi;     // (uniform)
        const int iyc = yok ? iy : 0;
#pragma unroll 3
        for (int kx = 0; kx < a.k; ++kx) {
          const int ix = ox * a.stride + kx - a.pad;
          const bool ok = yok && ix >= 0 && ix < a.Wi;
          const float* ip = inb + (iyc * a.Wi + (ok ? ix : 0)) * CR;
#pragma unroll
          for (int ci = 0; ci < CR; ++ci) {
            const float ld = ip[ci];
            const float x = ok ? ld : 0.f;
            const float4 w = *reinterpret_cast<const float4*>(sW1 + ((ky * a.k + kx) * CR + ci) * C + 4 * cq);
            acc.x = fmaf(x, w.x, acc.x); acc.y = fmaf(x, w.y, acc.y); acc.z = fmaf(x, w.z, acc.z); acc.w = fmaf(x, w.w, acc.w);
          }
        }
      }
      *reinterpret_cast<float4*>(a.out + ((long long)row * a.Wo + ox) * C + 4 * cq) = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WArgs {
  const float* u; const float* v; float* part;
  int N, Hu, Wu, Cu, Hv, Wv, Cv;
  int k, stride, pad;
  int bias_from;                                    // 0: none, 1: column sums of v, 2: column sums of u
  int nwg, psize;                                   // workgroups, floats per workgroup in `part`
  float* gw; float* gb;
  long long g_tap, g_u, g_v;
};

// NTW = tiles per wave (tile ti = (wave & 3) + 4 * i; ti = (tap * CUT + cut) * CVT + cvt).  SPLIT = 2: eight waves, waves 4..7 take
// the second half of the workgroup's positions and their tiles are added to those of waves 0..3 through LDS (a fixed order) --
// twice the loads in flight per CU for the same number of partial tiles.  UNR steps (two positions each) are requested before
// the first of their MFMAs: one step's operands are ~10 dword loads per lane, and without the unrolling the loop runs at the
// latency of one load per five MFMAs.
template <int NTW, int SPLIT>
__global__ __launch_bounds__(256 * SPLIT) void conv_train_wgrad_kernel(WArgs a) {
  constexpr int UNR = NTW <= 5 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) float s_x[];        // SPLIT 2: [4 waves][NTW][16][64] floats; bias sums: 256 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave & 3, part_id = wave >> 2;
  const int CUT = (a.Cu + 31) >> 5, CVT = (a.Cv + 31) >> 5;
  const int ntile = a.k * a.k * CUT * CVT;
  const long long Ms = (long long)a.N * a.Hv * a.Wv;
  // rows of the v grid (n, qy) are dealt to the (workgroup, half) parts: a wave's position loop then runs along one row, where
  // a tile's operand address moves by a constant and only the row's two ends need a bounds test
  const int NR = a.N * a.Hv;
  const int rows_per = (NR + a.nwg * SPLIT - 1) / (a.nwg * SPLIT);
  const int r0 = (blockIdx.x * SPLIT + part_id) * rows_per;
  const int r1 = r0 + rows_per < NR ? r0 + rows_per : NR;
  const int rc = lane & 31, kk = lane >> 5;
  int t_dy[NTW], t_dx[NTW], t_cu[NTW], t_cv[NTW];
  bool t_ok[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int ti = wq + 4 * i;
    t_ok[i] = ti < ntile;
    const int tc = t_ok[i] ? ti : 0;
    const int cvt = tc % CVT, r2 = tc / CVT, cut = r2 % CUT, tap = r2 / CUT;
    t_dy[i] = tap / a.k - a.pad;
    t_dx[i] = tap % a.k - a.pad;
    t_cu[i] = cut * 32 + rc;
    t_cv[i] = cvt * 32 + rc;
  }
  const bool a_shared = (4 % CVT) == 0;              // every tile of a wave has the same row tile: one A operand per step
  v16f acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bool cu_ok[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) cu_ok[i] = t_ok[i] && t_cu[i] < a.Cu;
  const int sCu = a.stride * a.Cu;
  for (int row = r0; row < r1; ++row) {
    const int n = row / a.Hv, qy = row - n * a.Hv;   // (wave-uniform)
    const float* vb = a.v + (long long)row * a.Wv * a.Cv;
    const float* ub[NTW];
    bool rok[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int iy = qy * a.stride + t_dy[i];
      rok[i] = cu_ok[i] && iy >= 0 && iy < a.Hu;
      ub[i] = a.u + (((long long)n * a.Hu + iy) * a.Wu + t_dx[i]) * a.Cu + t_cu[i];
    }
    for (int qx0 = 0; qx0 < a.Wv; qx0 += 2 * UNR) {
      float av[UNR][NTW], bv[UNR][NTW];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (qx0 + 2 * u < a.Wv) {                    // (wave-uniform: the last batch of a row may be short)
          const int qx = qx0 + 2 * u + kk;
          const bool pv = qx < a.Wv;
          const int xo = qx * sCu, xb = qx * a.stride;
          if (CT_DBG & 1) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) { av[u][i] = (float)(lane + u); bv[u][i] = (float)(qx + i); }
            continue;
          }
          if (a_shared) {
            const float x = (pv && t_cv[0] < a.Cv) ? vb[qx * a.Cv + t_cv[0]] : 0.f;
#pragma unroll
            for (int i = 0; i < NTW; ++i) av[u][i] = x;
          } else {
#pragma unroll
            for (int i = 0; i < NTW; ++i) av[u][i] = (pv && t_ok[i] && t_cv[i] < a.Cv) ? vb[qx * a.Cv + t_cv[i]] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            const int ix = xb + t_dx[i];
            bv[u][i] = (pv && rok[i] && ix >= 0 && ix < a.Wu) ? ub[i][xo] : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (qx0 + 2 * u < a.Wv) {
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            if (CT_DBG & 2) acc[i][0] += av[u][i] + bv[u][i];
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][i], acc[i], 0, 0, 0);
          }
        }
    }
  }
  if (SPLIT == 2) {
    if (part_id == 1) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_x[((wq * NTW + i) * 16 + r) * 64 + lane] = acc[i][r];
    }
    __syncthreads();
    if (part_id == 0) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += s_x[((wq * NTW + i) * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  float* part = a.part + (long long)blockIdx.x * a.psize;
  if (part_id == 0) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      if (t_ok[i]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(long long)(wq + 4 * i) * 1024 + r * 64 + lane] = acc[i][r];
      }
  }
  // bias gradient: column sums of this workgroup's slice of gy
  if (a.bias_from) {
    constexpr int NT = 256 * SPLIT;
    const float* g = a.bias_from == 1 ? a.v : a.u;
    const int C = a.bias_from == 1 ? a.Cv : a.Cu;
    const long long Mg = a.bias_from == 1 ? Ms : (long long)a.N * a.Hu * a.Wu;
    const long long perg = (Mg + a.nwg - 1) / a.nwg;
    const long long g0 = (long long)blockIdx.x * perg;
    long long g1 = g0 + perg;
    g1 = g1 < Mg ? g1 : Mg;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int RL, c;
    if ((C & 3) == 0) {
      // 16-byte loads, four rows in flight per thread (one request at a time made this loop the longest part of the launch)
      const int CQ = C >> 2;
      RL = NT / CQ;
      const int cq = tid % CQ, rl = tid / CQ;
      c = cq;
      if (rl < RL) {
        long long row = g0 + rl;
        for (; row + 3 * RL < g1; row += 4 * RL) {
          const float4 x0 = *reinterpret_cast<const float4*>(g + row * C + 4 * cq);
          const float4 x1 = *reinterpret_cast<const float4*>(g + (row + RL) * C + 4 * cq);
          const float4 x2 = *reinterpret_cast<const float4*>(g + (row + 2 * RL) * C + 4 * cq);
          const float4 x3 = *reinterpret_cast<const float4*>(g + (row + 3 * RL) * C + 4 * cq);
          s0 += x0.x; s1 += x0.y; s2 += x0.z; s3 += x0.w;
          s0 += x1.x; s1 += x1.y; s2 += x1.z; s3 += x1.w;
          s0 += x2.x; s1 += x2.y; s2 += x2.z; s3 += x2.w;
          s0 += x3.x; s1 += x3.y; s2 += x3.z; s3 += x3.w;
        }
        for (; row < g1; row += RL) {
          const float4 x0 = *reinterpret_cast<const float4*>(g + row * C + 4 * cq);
          s0 += x0.x; s1 += x0.y; s2 += x0.z; s3 += x0.w;
        }
      }
      __syncthreads();                               // (SPLIT 2: s_x held the second half's tiles)
      if (rl < RL) {
        s_x[(rl * CQ + cq) * 4 + 0] = s0; s_x[(rl * CQ + cq) * 4 + 1] = s1;
        s_x[(rl * CQ + cq) * 4 + 2] = s2; s_x[(rl * CQ + cq) * 4 + 3] = s3;
      }
    } else {
      RL = NT / C;                                   // (C <= 64; threads beyond RL * C idle)
      c = tid % C;
      const int rl = tid / C;
      if (rl < RL)
        for (long long row = g0 + rl; row < g1; row += RL) s0 += g[row * C + c];
      __syncthreads();
      if (rl < RL) s_x[rl * C + c] = s0;
    }
    __syncthreads();
    if (tid < C) {
      float tot = 0.f;
      for (int q = 0; q < RL; ++q) tot += s_x[q * C + tid];
      part[(long long)ntile * 1024 + tid] = tot;
    }
  }
}

// The same weight gradient with its operands staged through LDS (the default where the tiles fit).  A block = RB rows of the
// coarse grid: their v records ([RB][Wv + 1][Cv], the extra column zero: the odd position of a row's last step) and, for every row,
// the k input rows of u it meets ([RB * k][Wv * s + k][Cu], zero columns for x < 0 and x >= Wu, zero rows for y outside the map):
// 16-byte requests along the channels-last records, issued a block ahead into registers and written to the other buffer behind
// the block's MFMAs (one barrier per block).  The MFMA loop then reads one dword per operand from LDS at base(tile) + step offset:
// no bounds test, no 64-bit address and no global request inside it (the direct form above spends 17 vector instructions per MFMA
// and runs at 20 % of the matrix pipe).  Two halves of four waves with their own buffers, added through LDS as above.
constexpr int WL_MAXCH = 8;
struct WLGeo { int RB, WP, nb, rows_per, svb, sub, nv4, nu4; };     // svb / sub: floats per v / u buffer; nv4 / nu4: 16-byte chunks per block

template <int NTW>
__global__ __launch_bounds__(512) void conv_train_wgrad_lds_kernel(WArgs a, WLGeo g) {
  extern __shared__ __attribute__((aligned(16))) float s_l[];        // [half][buffer]{v, u}; aliased by the combine tiles and the bias sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wq = wave & 3, half = wave >> 2, th = tid & 255;
  const int CUT = (a.Cu + 31) >> 5, CVT = (a.Cv + 31) >> 5;
  const int ntile = a.k * a.k * CUT * CVT;
  const int rc = lane & 31, kk = lane >> 5;
  const int NR = a.N * a.Hv;
  const int r0 = (blockIdx.x * 2 + half) * g.rows_per;
  const int r1 = r0 + g.rows_per < NR ? r0 + g.rows_per : NR;
  const int bufsz = g.svb + g.sub;
  constexpr int NBUF = CT_WL_PINGPONG ? 1 : 2;
  float* const sh = s_l + half * NBUF * bufsz;
  // zero both buffers of this half once: the padding columns are never written again
  for (int i = th; i < NBUF * bufsz / 4; i += 256) reinterpret_cast<float4*>(sh)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // this thread's chunks of a block: [0, nv4) v records, [nv4, nv4 + nu4) u records
  const int vq = a.Wv * a.Cv / 4, uq = a.Wu * a.Cu / 4;             // 16-byte chunks per v row / u row
  int ch_lds[WL_MAXCH], ch_src[WL_MAXCH], ch_row[WL_MAXCH], ch_ky[WL_MAXCH];   // LDS float offset in a buffer, float offset in the source row, r, ky
  bool ch_u[WL_MAXCH], ch_ok[WL_MAXCH];
#pragma unroll
  for (int c = 0; c < WL_MAXCH; ++c) {
    const int e = th + c * 256;
    ch_ok[c] = e < g.nv4 + g.nu4;
    ch_u[c] = e >= g.nv4;
    if (!ch_u[c]) {
      const int r = e / vq, q = e - r * vq;
      ch_row[c] = r; ch_ky[c] = 0; ch_src[c] = 4 * q;
      ch_lds[c] = r * (a.Wv + 1) * a.Cv + 4 * q;
    } else {
      const int e2 = e - g.nv4;
      const int rk = e2 / uq, q = e2 - rk * uq;
      ch_row[c] = rk / a.k; ch_ky[c] = rk % a.k; ch_src[c] = 4 * q;
      ch_lds[c] = g.svb + (rk * g.WP + a.pad) * a.Cu + 4 * q;
    }
  }
  int t_b[NTW], t_cv[NTW];
  bool t_ok[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int ti = wq + 4 * i;
    t_ok[i] = ti < ntile;
    const int tc = t_ok[i] ? ti : 0;
    const int cvt = tc % CVT, r2 = tc / CVT, cut = r2 % CUT, tap = r2 / CUT;
    int cu = cut * 32 + rc;
    cu = cu < a.Cu ? cu : 0;                         // (columns beyond Cu are never read back by the reduce launch)
    t_b[i] = g.svb + ((tap / a.k) * g.WP + tap % a.k) * a.Cu + cu;
    int cv = cvt * 32 + rc;
    t_cv[i] = cv < a.Cv ? cv : 0;
  }
  v16f acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float4 st[WL_MAXCH];
  auto request = [&](int b) {                        // block b of this half -> registers
    const int row0 = r0 + b * g.RB;
    int ubase[4];                                    // (wave-uniform) first input row of block row r: n * Hu + qy * s - pad
    int qys[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + r;
      const int n = row / a.Hv, qy = row - n * a.Hv;
      qys[r] = qy * a.stride - a.pad;
      ubase[r] = n * a.Hu + qys[r];
    }
#pragma unroll
    for (int c = 0; c < WL_MAXCH; ++c) {
      st[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ch_ok[c]) {
        const int r = ch_row[c];
        const int row = row0 + r;
        if (!ch_u[c]) {
          if (row < r1) st[c] = *reinterpret_cast<const float4*>(a.v + (long long)row * a.Wv * a.Cv + ch_src[c]);
        } else {
          const int ub = r == 0 ? ubase[0] : r == 1 ? ubase[1] : r == 2 ? ubase[2] : ubase[3];
          const int iy = (r == 0 ? qys[0] : r == 1 ? qys[1] : r == 2 ? qys[2] : qys[3]) + ch_ky[c];
          if (row < r1 && iy >= 0 && iy < a.Hu)
            st[c] = *reinterpret_cast<const float4*>(a.u + (long long)(ub + ch_ky[c]) * a.Wu * a.Cu + ch_src[c]);
        }
      }
    }
  };
  auto deposit = [&](int buf) {
    float* d = sh + buf * bufsz;
#pragma unroll
    for (int c = 0; c < WL_MAXCH; ++c)
      if (ch_ok[c]) *reinterpret_cast<float4*>(d + ch_lds[c]) = st[c];
  };
  const int SR = (a.Wv + 1) >> 1;
  const int sCu = a.stride * a.Cu;
  __syncthreads();                                   // (the zero fill)
  // bias gradient from the staged rows (every gy row passes through LDS exactly once: as a v row, or -- gy on the finer grid -- as
  // the input rows ky in [pad, pad + s) of its coarse row): no second pass over gy in global memory (9 us of an 87 us launch)
  const int Cb = a.bias_from == 1 ? a.Cv : a.Cu;
  const bool fuse_bias = CT_WL_FUSE_BIAS && a.bias_from != 0 && (256 % Cb) == 0 &&
                         (a.bias_from == 1 || (a.k >= a.pad + a.stride && a.Hu == a.stride * a.Hv));
  float bsum = 0.f;
  auto bias_rows = [&](const float* sv) {
    if (a.bias_from == 1) {
      const int n = g.RB * (a.Wv + 1) * a.Cv;
      for (int e = th; e < n; e += 256) bsum += sv[e];
    } else {
      const int n = g.WP * a.Cu;
      for (int r = 0; r < g.RB; ++r)
        for (int ky = a.pad; ky < a.pad + a.stride; ++ky) {
          const float* row = sv + g.svb + (r * a.k + ky) * n;
          for (int e = th; e < n; e += 256) bsum += row[e];
        }
    }
  };
  // (the operands of step s + 1 are read from LDS BEFORE the MFMAs of step s are issued, and the scheduler is told to keep it that way:
  //  left alone hipcc put every ds_read directly in front of its MFMA with a wait between them -- an LDS round trip per MFMA, the matrix
  //  pipe 31 % busy)
  auto compute = [&](const float* sv) {
    if (fuse_bias) bias_rows(sv);
    // (Cv <= 64: at most two row tiles, and tiles wq, wq + 4, ... of a wave share theirs -- ONE A operand per step)
    auto fetch = [&](int r, int s2, float& av, float (&bv)[NTW]) {
      const float* pv = sv + r * (a.Wv + 1) * a.Cv + kk * a.Cv + 2 * s2 * a.Cv;
      const float* pu = sv + r * a.k * g.WP * a.Cu + kk * sCu + 2 * s2 * sCu;
      av = pv[t_cv[0]];
#pragma unroll
      for (int i = 0; i < NTW; ++i) bv[i] = pu[t_b[i]];
    };
    // two register sets, no copies between them (a copy of a just-read value is a wait for the read in front of the MFMAs)
    float av0, bv0[NTW], av1, bv1[NTW];
    const int nstep = g.RB * SR;
    int r = 0, s2 = 0;
    auto advance = [&]() { if (++s2 == SR) { s2 = 0; ++r; } };
    auto multiply = [&](const float av, const float (&bv)[NTW]) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        if (CT_DBG & 64) acc[i][0] += av + bv[i];
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i], acc[i], 0, 0, 0);
      }
    };
    fetch(0, 0, av0, bv0);
    advance();
    // (the reads are unconditional -- one step past the block reads LDS words nobody uses: a conditional read is a control-flow join, and
    //  hipcc waits for EVERY outstanding LDS read at a join, i.e. for the prefetch it has just issued)
    for (int st = 0; st < nstep; st += 2) {
      fetch(r, s2, av1, bv1); advance();
      __builtin_amdgcn_sched_barrier(0);
      multiply(av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      fetch(r, s2, av0, bv0); advance();
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < nstep) multiply(av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (CT_WL_PINGPONG) {
    // ONE buffer per half and two barriers per block: while half 0 multiplies block b, half 1 writes its block b to LDS and
    // requests the next one; then they swap.  The two waves of a SIMD are never in their MFMA phase together (together they
    // only share the pipe: the phase takes twice as long), and a half's requests / LDS writes run beside the other half's MFMAs
    // instead of in front of its own.  (Off by default: see the knob.)
    request(0);
    if (half == 0) deposit(0);
    __syncthreads();
    for (int b = 0; b < g.nb; ++b) {
      const bool cur = r0 + b * g.RB < r1, more = r0 + (b + 1) * g.RB < r1;
      if (half == 0) {
        if (more && !(CT_DBG & 128)) request(b + 1);
        if (cur) compute(sh);
      } else if (cur) deposit(0);
      __syncthreads();
      if (half == 0) {
        if (more) deposit(0);
      } else {
        if (more && !(CT_DBG & 128)) request(b + 1);
        if (cur) compute(sh);
      }
      __syncthreads();
    }
  } else {
    request(0);
    deposit(0);
    __syncthreads();
    for (int b = 0; b < g.nb; ++b) {
      const bool more = r0 + (b + 1) * g.RB < r1;
      if (more && !(CT_DBG & 128)) request(b + 1);
      if (r0 + b * g.RB < r1) compute(sh + (b & 1) * bufsz);
      if (more) deposit((b + 1) & 1);
      __syncthreads();
    }
  }
  // second half's tiles through LDS (aliases the operand buffers: every wave is behind the last block's barrier)
  float* const s_x = s_l;
  if (half == 1) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_x[((wq * NTW + i) * 16 + r) * 64 + lane] = acc[i][r];
  }
  __syncthreads();
  float* part = a.part + (long long)blockIdx.x * a.psize;
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
      if (t_ok[i]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(long long)(wq + 4 * i) * 1024 + r * 64 + lane] = acc[i][r] + s_x[((wq * NTW + i) * 16 + r) * 64 + lane];
      }
  }
  if (fuse_bias && !(CT_DBG & 32)) {
    __syncthreads();
    s_x[tid] = bsum;                                 // (tid & 255) % Cb is the thread's channel, in either half
    __syncthreads();
    if (tid < Cb) {
      float tot = 0.f;
      for (int q = tid; q < 512; q += Cb) tot += s_x[q];
      part[(long long)ntile * 1024 + tid] = tot;
    }
  } else if (a.bias_from && !(CT_DBG & 32)) {
    __syncthreads();
    constexpr int NT = 512;
    const float* gq = a.bias_from == 1 ? a.v : a.u;
    const int C = a.bias_from == 1 ? a.Cv : a.Cu;    // (C % 4 == 0 on this path)
    const long long Mg = a.bias_from == 1 ? (long long)NR * a.Wv : (long long)a.N * a.Hu * a.Wu;
    const long long perg = (Mg + a.nwg - 1) / a.nwg;
    const long long g0 = (long long)blockIdx.x * perg;
    long long g1 = g0 + perg;
    g1 = g1 < Mg ? g1 : Mg;
    const int CQ = C >> 2, RL = NT / CQ;
    const int cq = tid % CQ, rl = tid / CQ;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (rl < RL) {
      long long row = g0 + rl;
      for (; row + 3 * RL < g1; row += 4 * RL) {
        const float4 x0 = *reinterpret_cast<const float4*>(gq + row * C + 4 * cq);
        const float4 x1 = *reinterpret_cast<const float4*>(gq + (row + RL) * C + 4 * cq);
        const float4 x2 = *reinterpret_cast<const float4*>(gq + (row + 2 * RL) * C + 4 * cq);
        const float4 x3 = *reinterpret_cast<const float4*>(gq + (row + 3 * RL) * C + 4 * cq);
        s0 += x0.x; s1 += x0.y; s2 += x0.z; s3 += x0.w;
        s0 += x1.x; s1 += x1.y; s2 += x1.z; s3 += x1.w;
        s0 += x2.x; s1 += x2.y; s2 += x2.z; s3 += x2.w;
        s0 += x3.x; s1 += x3.y; s2 += x3.z; s3 += x3.w;
      }
      for (; row < g1; row += RL) {
        const float4 x0 = *reinterpret_cast<const float4*>(gq + row * C + 4 * cq);
        s0 += x0.x; s1 += x0.y; s2 += x0.z; s3 += x0.w;
      }
      s_x[(rl * CQ + cq) * 4 + 0] = s0; s_x[(rl * CQ + cq) * 4 + 1] = s1;
      s_x[(rl * CQ + cq) * 4 + 2] = s2; s_x[(rl * CQ + cq) * 4 + 3] = s3;
    }
    __syncthreads();
    if (tid < C) {
      float tot = 0.f;
      for (int q = 0; q < RL; ++q) tot += s_x[q * C + tid];
      part[(long long)ntile * 1024 + tid] = tot;
    }
  }
}

// adds the partials of all workgroups and scatters: tiles (MODE 0) or plain [tap][c] rows (MODE 1).  Eight threads per output
// element take every eighth workgroup's partial and are added in thread order: a fixed order, and loads that do not wait for
// one another.
template <int MODE>
__global__ __launch_bounds__(256) void conv_train_wgrad_reduce_kernel(WArgs a) {
  constexpr int TPE = MODE == 0 ? 8 : 32, EPB = 256 / TPE;          // threads per element, elements per block
  __shared__ float s_q[256];
  const int CUT = (a.Cu + 31) >> 5, CVT = (a.Cv + 31) >> 5;
  const int nt = a.k * a.k;
  const int E = nt * a.Cu * a.Cv;
  const int CB = a.bias_from == 0 ? 0 : (a.bias_from == 1 ? a.Cv : a.Cu);
  const int el = threadIdx.x % EPB, q = threadIdx.x / EPB;
  const int e = blockIdx.x * EPB + el;
  long long src = 0;
  float* dst = nullptr;
  if (e < E) {
    if (MODE == 0) {
      const int cu = e % a.Cu, r2 = e / a.Cu, cv = r2 % a.Cv, tap = r2 / a.Cv;
      const int ti = (tap * CUT + (cu >> 5)) * CVT + (cv >> 5);
      const int row = cv & 31, col = cu & 31;
      src = (long long)ti * 1024 + ((row >> 3) * 4 + (row & 3)) * 64 + col + 32 * ((row >> 2) & 1);
      dst = a.gw + tap * a.g_tap + cu * a.g_u + cv * a.g_v;
    } else {
      const int cv = e % a.Cv, r2 = e / a.Cv, cu = r2 % a.Cu, tap = r2 / a.Cu;
      src = e;
      dst = a.gw + tap * a.g_tap + cu * a.g_u + cv * a.g_v;
    }
  } else if (e < E + CB) {
    src = (MODE == 0 ? (long long)nt * CUT * CVT * 1024 : (long long)E) + (e - E);
    dst = a.gb + (e - E);
  }
  float s = 0.f;
  if (dst) {
    int w = q;
    for (; w + 3 * TPE < a.nwg; w += 4 * TPE) {                      // four requests in flight, added in index order
      const float x0 = a.part[(long long)w * a.psize + src], x1 = a.part[(long long)(w + TPE) * a.psize + src];
      const float x2 = a.part[(long long)(w + 2 * TPE) * a.psize + src], x3 = a.part[(long long)(w + 3 * TPE) * a.psize + src];
      s += x0; s += x1; s += x2; s += x3;
    }
    for (; w < a.nwg; w += TPE) s += a.part[(long long)w * a.psize + src];
  }
  s_q[threadIdx.x] = s;
  __syncthreads();
  if (q == 0 && dst) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < TPE; ++k) tot += s_q[k * EPB + el];
    *dst = tot;
  }
}

// one to four channels in U (the first layer's input, the read-out layer's gy): D[tap][cu][cv] = sum over (n, q) of
// u[n, q * s - pad + k][cu] * V[n, q][cv].  NTM = the largest tap count the instantiation holds accumulators for.
template <int CU, int NTM>
__global__ __launch_bounds__(256) void conv_train_c1_wgrad_kernel(WArgs a) {
  extern __shared__ float s_acc[];                  // [4 waves][(nt * CU + 1) * Cv]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = a.Cv, CQ = C >> 2, RL = 256 / CQ;
  const int nt = a.k * a.k;
  const int cq = tid % CQ, rl = tid / CQ;           // (CQ is a power of two <= 16: lanes l, l + CQ, ... of a wave share a channel quad)
  // (positions, not rows, are dealt to the threads: a row-wise walk leaves 256 / (Cv / 4) - Wv thread rows idle -- measured 60 against
  //  41 us per launch on the two layers that use this kernel)
  const long long Ms = (long long)a.N * a.Hv * a.Wv;
  const long long per = (Ms + a.nwg - 1) / a.nwg;
  const long long p0 = (long long)blockIdx.x * per;
  long long p1 = p0 + per;
  p1 = p1 < Ms ? p1 : Ms;
  float4 acc[NTM * CU];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float usum[CU];
#pragma unroll
  for (int i = 0; i < CU; ++i) usum[i] = 0.f;
#pragma unroll
  for (int t = 0; t < NTM * CU; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the position's (n, qy, qx) is decoded once and then stepped: two 64-bit divisions per trip were a third of this loop)
  int qx, qy, n;
  {
    const long long pf = p0 + rl;
    qx = (int)(pf % a.Wv);
    const long long t2 = pf / a.Wv;
    qy = (int)(t2 % a.Hv); n = (int)(t2 / a.Hv);
  }
  for (long long p = p0 + rl; p < p1; p += RL) {
    const float4 g = *reinterpret_cast<const float4*>(a.v + p * C + 4 * cq);
    bsum.x += g.x; bsum.y += g.y; bsum.z += g.z; bsum.w += g.w;
    const float* un = a.u + (long long)n * a.Hu * a.Wu * CU;
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
      if (t < nt) {
        const int iy = qy * a.stride - a.pad + t / a.k, ix = qx * a.stride - a.pad + t % a.k;
        const bool ok = iy >= 0 && iy < a.Hu && ix >= 0 && ix < a.Wu;
        const float* up = un + ((ok ? iy : 0) * a.Wu + (ok ? ix : 0)) * CU;
#pragma unroll
        for (int cu = 0; cu < CU; ++cu) {
          const float ld = up[cu];
          const float x = ok ? ld : 0.f;
          float4& d = acc[t * CU + cu];
          d.x = fmaf(x, g.x, d.x); d.y = fmaf(x, g.y, d.y); d.z = fmaf(x, g.z, d.z); d.w = fmaf(x, g.w, d.w);
        }
      }
    }
    qx += RL;
    while (qx >= a.Wv) {
      qx -= a.Wv;
      if (++qy >= a.Hv) { qy = 0; ++n; }
    }
  }
  if (a.bias_from == 2) {                           // column sums of the few-channel tensor: its own slice
    const long long Mu = (long long)a.N * a.Hu * a.Wu;
    const long long peru = (Mu + a.nwg - 1) / a.nwg;
    const long long u0 = (long long)blockIdx.x * peru;
    long long u1 = u0 + peru;
    u1 = u1 < Mu ? u1 : Mu;
    for (long long q = u0 + tid; q < u1; q += 256) {
#pragma unroll
      for (int cu = 0; cu < CU; ++cu) usum[cu] += a.u[q * CU + cu];
    }
  }
  // workgroup reduction in a fixed order: lanes of a wave that share a channel quad by exchanges, the four waves through LDS
  const int W = (nt * CU + 1) * C;
  auto wave_sum4 = [&](float4& v) {
    for (int m = CQ; m < 64; m <<= 1) {
      v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
    }
  };
#pragma unroll
  for (int t = 0; t < NTM * CU; ++t)
    if (t < nt * CU) {
      wave_sum4(acc[t]);
      if (lane < CQ) *reinterpret_cast<float4*>(s_acc + wave * W + t * C + 4 * cq) = acc[t];
    }
  wave_sum4(bsum);
  if (lane < CQ) *reinterpret_cast<float4*>(s_acc + wave * W + nt * CU * C + 4 * cq) = bsum;
  __syncthreads();
  float* part = a.part + (long long)blockIdx.x * a.psize;
  for (int e = tid; e < W; e += 256) {
    const float s = ((s_acc[e] + s_acc[W + e]) + s_acc[2 * W + e]) + s_acc[3 * W + e];
    if (e < nt * CU * C) part[e] = s;
    else if (a.bias_from == 1) part[e] = s;
  }
  if (a.bias_from == 2) {
    __syncthreads();
#pragma unroll
    for (int cu = 0; cu < CU; ++cu) s_acc[cu * 256 + tid] = usum[cu];
    __syncthreads();
    if (tid < CU) {
      float s = 0.f;
      for (int q = 0; q < 256; ++q) s += s_acc[tid * 256 + q];
      part[nt * CU * C + tid] = s;
    }
  }
}

int gather_kind(int Cred, int Cout, int k, int stride, int form) {
  if (k < 1 || k * k > MAX_TAPS || stride < 1 || Cred < 1 || Cout < 1) return 0;
  if (Cred <= 4) return (!form && (Cout % 4) == 0 && k * k * Cred * Cout * 4 <= 64 * 1024) ? 3 : 0;
  if (Cout <= 4) {
    if (form && stride != 1) return 0;
    return (Cred == 8 || Cred == 16 || Cred == 32 || Cred == 64) ? 2 : 0;
  }
  if (Cred % 8 != 0 || Cred > 64 || Cout > 64) return 0;
  if (stride > 4 || (form && stride > 2) || (long long)k * k * Cred * 32 * 4 + (CT_STAGE_A ? 8 * 32 * Cred * 4 : 0) > GATHER_LDS_MAX) return 0;    // (one column tile per workgroup always fits then)
  return 1;
}

template <int J, int CN, int RM>
void launch_gather(const GArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  // (the opt-in to > 64 KB of dynamic LDS: once per instantiation -- thread-safe -- and outside any stream capture: the warm-up
  //  iterations in front of a capture make the first call)
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_train_gather_kernel<J, CN, RM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              152 * 1024);
  });
  ClassTab ct;
  build_class_tab(ct, a, RM * 32);
  hipLaunchKernelGGL((conv_train_gather_kernel<J, CN, RM>), grid, dim3(512), lds, s, a, ct);
}
template <int J>
void launch_gather_j(const GArgs& a, int CN, int RM, dim3 grid, size_t lds, hipStream_t s) {
  if (CN == 2) launch_gather<J, 2, 1>(a, grid, lds, s);
  else launch_gather<J, 1, 1>(a, grid, lds, s);
}

}  // namespace

extern "C" int spk_conv_train_gather_supported(int Cred, int Cout, int k, int stride, int form) {
  return gather_kind(Cred, Cout, k, stride, form) != 0;
}

extern "C" int spk_conv_train_gather(const float* in_cl, const float* w, const float* bias_or_null, float* out_cl, int N, int Hi,
                                     int Wi, int Cred, int Ho, int Wo, int Cout, int k, int stride, int pad, int form,
                                     long long w_tap, long long w_red, long long w_out, spk_stream_t stream) {
  if (!in_cl || !w || !out_cl || N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || pad < 0) return SPK_ERR_ARG;
  const int kind = gather_kind(Cred, Cout, k, stride, form);
  if (!kind) return SPK_ERR_UNSUPPORTED;
  if ((long long)N * Ho * Wo >= (1ll << 31) / 64 * 64 || (long long)N * Hi * Wi * Cred >= (1ll << 31)) return SPK_ERR_UNSUPPORTED;
  GArgs a{in_cl, w, bias_or_null, out_cl, N, Hi, Wi, Cred, Ho, Wo, Cout, k, stride, pad, form, w_tap, w_red, w_out};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (kind == 3) {
    const int rows = N * Ho;
    const dim3 gr(rows < CT_C1_ROWS_CAP ? rows : CT_C1_ROWS_CAP);
    const size_t ldsw = (size_t)k * k * Cred * Cout * 4;
    if (Cred == 1) hipLaunchKernelGGL(conv_train_c1in_kernel<1>, gr, dim3(256), ldsw, s, a);
    else if (Cred == 2) hipLaunchKernelGGL(conv_train_c1in_kernel<2>, gr, dim3(256), ldsw, s, a);
    else if (Cred == 3) hipLaunchKernelGGL(conv_train_c1in_kernel<3>, gr, dim3(256), ldsw, s, a);
    else hipLaunchKernelGGL(conv_train_c1in_kernel<4>, gr, dim3(256), ldsw, s, a);
  } else if (kind == 2) {
    const int LP = Cred / 4;
    const int blocks = N * Ho < CT_C1_ROWS_CAP ? N * Ho : CT_C1_ROWS_CAP;
#define SPK_C1OUT(LP_)                                                                                                 \
  do {                                                                                                                 \
    if (Cout == 1) hipLaunchKernelGGL((conv_train_c1out_kernel<LP_, 1>), dim3(blocks), dim3(256), 0, s, a);            \
    else if (Cout == 2) hipLaunchKernelGGL((conv_train_c1out_kernel<LP_, 2>), dim3(blocks), dim3(256), 0, s, a);       \
    else if (Cout == 3) hipLaunchKernelGGL((conv_train_c1out_kernel<LP_, 3>), dim3(blocks), dim3(256), 0, s, a);       \
    else hipLaunchKernelGGL((conv_train_c1out_kernel<LP_, 4>), dim3(blocks), dim3(256), 0, s, a);                      \
  } while (0)
    if (LP == 2) SPK_C1OUT(2); else if (LP == 4) SPK_C1OUT(4); else if (LP == 8) SPK_C1OUT(8); else SPK_C1OUT(16);
#undef SPK_C1OUT
  } else {
    const int cs = form ? stride : 1;
    if (cs * cs > MAX_CLASSES) return SPK_ERR_UNSUPPORTED;
    const int CNT = (Cout + 31) / 32, J = Cred / 8;
    if (CT_SUBPIXEL && form && stride == 2 && k <= 3 && (J == 1 || J == 2 || J == 4 || J == 8)) {
      // all four sub-pixel classes of 32 coarse positions per item
      const long long items = ((long long)N * ((Ho + 1) / 2) * ((Wo + 1) / 2) + 31) / 32;
      const bool big = CNT == 1 || (items >= CT_BIG_ITEMS / 4 && (size_t)k * k * Cred * CNT * 32 * 4 <= 150 * 1024);
      const int CN = big ? CNT : 1, gy = big ? 1 : CNT;
      const size_t lds = (size_t)k * k * Cred * CN * 32 * 4;
      int gx = (int)((items + 7) / 8);
      const int per_cu = lds > 76 * 1024 ? 1 : 2;
      const int cap = 256 * per_cu / gy > 0 ? 256 * per_cu / gy : 1;
      gx = gx < cap ? gx : cap;
      SubTab tb;
      build_sub_tab(tb, a);
#define SPK_SUB_LAUNCH(J_, CN_)                                                                                        \
  do {                                                                                                                 \
    static std::once_flag once_;                                                                                       \
    std::call_once(once_, [] {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_train_gather_sub_kernel<J_, CN_>),                                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);                               \
    });                                                                                                                \
    hipLaunchKernelGGL((conv_train_gather_sub_kernel<J_, CN_>), dim3(gx, gy), dim3(512), lds, s, a, tb);               \
  } while (0)
      if (CN == 2) {
        if (J == 1) SPK_SUB_LAUNCH(1, 2); else if (J == 2) SPK_SUB_LAUNCH(2, 2); else if (J == 4) SPK_SUB_LAUNCH(4, 2); else SPK_SUB_LAUNCH(8, 2);
      } else {
        if (J == 1) SPK_SUB_LAUNCH(1, 1); else if (J == 2) SPK_SUB_LAUNCH(2, 1); else if (J == 4) SPK_SUB_LAUNCH(4, 1); else SPK_SUB_LAUNCH(8, 1);
      }
#undef SPK_SUB_LAUNCH
      SPK_LAUNCH_CHECK();
      return SPK_OK;
    }
    // items of 32 rows x all column tiles when there are at least four per SIMD; otherwise 32 rows x one column tile
    long long items = 0;
    for (int c = 0; c < cs * cs; ++c)
      items += ((long long)N * ((Ho - c / cs + cs - 1) / cs) * ((Wo - c % cs + cs - 1) / cs) + 31) / 32;
    const bool staged = CT_STAGE_A && (J == 1 || J == 2 || J == 4 || J == 8);
    const size_t lds_a = staged ? (size_t)8 * 32 * Cred * 4 : 0;
    const bool big = CNT == 1 || (items >= CT_BIG_ITEMS && (size_t)k * k * Cred * CNT * 32 * 4 + lds_a <= 150 * 1024);
    const int CN = big ? CNT : 1, RM = 1;
    const int gy = big ? 1 : CNT;
    const size_t lds = (size_t)k * k * Cred * CN * 32 * 4 + lds_a;
    int gx = (int)((items + 7) / 8);
    const int per_cu = lds > 76 * 1024 ? 1 : 2;                      // workgroups per CU that fit the LDS
    const int cap = 256 * per_cu / gy > 0 ? 256 * per_cu / gy : 1;
    gx = gx < cap ? gx : cap;
    dim3 grid(gx, gy);
    switch (J) {
      case 1: launch_gather_j<1>(a, CN, RM, grid, lds, s); break;
      case 2: launch_gather_j<2>(a, CN, RM, grid, lds, s); break;
      case 3: launch_gather_j<3>(a, CN, RM, grid, lds, s); break;
      case 4: launch_gather_j<4>(a, CN, RM, grid, lds, s); break;
      case 5: launch_gather_j<5>(a, CN, RM, grid, lds, s); break;
      case 6: launch_gather_j<6>(a, CN, RM, grid, lds, s); break;
      case 7: launch_gather_j<7>(a, CN, RM, grid, lds, s); break;
      default: launch_gather_j<8>(a, CN, RM, grid, lds, s); break;
    }
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

namespace {
int wgrad_kind(int Cu, int Cv, int k) {
  if (k < 1 || k * k > MAX_TAPS || Cu < 1 || Cv < 1) return 0;
  if (Cu <= 4) {
    const int CQ = Cv / 4;
    const bool cq_ok = Cv % 4 == 0 && (CQ == 1 || CQ == 2 || CQ == 4 || CQ == 8 || CQ == 16);
    return (cq_ok && (Cu == 1 || k <= 3) && 4 * (k * k * Cu + 1) * Cv * 4 <= 64 * 1024) ? 2 : 0;
  }
  if (Cu > 64 || Cv > 64) return 0;
  const int ntile = k * k * ((Cu + 31) / 32) * ((Cv + 31) / 32);
  return ntile <= 36 ? 1 : 0;
}
int wgrad_nwg(int kind, long long Ms) {
  long long n = kind == 1 ? (Ms + 63) / 64 : (Ms + 255) / 256;      // at least 64 / 256 positions per workgroup
  const int cap = kind == 1 ? 256 : CT_C1W_CAP;                     // (vector kernel: tiny partials, latency-bound position loop)
  return (int)(n < 1 ? 1 : (n < cap ? n : cap));
}
int wgrad_psize(int kind, int Cu, int Cv, int k) {
  if (kind == 1) return k * k * ((Cu + 31) / 32) * ((Cv + 31) / 32) * 1024 + 64;
  return (k * k * Cu + 1) * Cv + 8;
}
}  // namespace

extern "C" long long spk_conv_train_wgrad_ws_bytes(int N, int Hv, int Wv, int Cu, int Cv, int k) {
  const int kind = wgrad_kind(Cu, Cv, k);
  if (!kind || N <= 0 || Hv <= 0 || Wv <= 0) return -1;
  const long long Ms = (long long)N * Hv * Wv;
  return (long long)wgrad_nwg(kind, Ms) * wgrad_psize(kind, Cu, Cv, k) * 4;
}

extern "C" int spk_conv_train_wgrad(const float* u_cl, const float* v_cl, float* ws, long long ws_bytes, float* gw_out,
                                    float* gb_out_or_null, int N, int Hu, int Wu, int Cu, int Hv, int Wv, int Cv, int k, int stride,
                                    int pad, long long g_tap, long long g_u, long long g_v, int bias_from, spk_stream_t stream) {
  if (!u_cl || !v_cl || !ws || !gw_out || N <= 0 || Hu <= 0 || Wu <= 0 || Hv <= 0 || Wv <= 0 || stride < 1 || pad < 0 ||
      bias_from < 0 || bias_from > 2)
    return SPK_ERR_ARG;
  const int kind = wgrad_kind(Cu, Cv, k);
  if (!kind) return SPK_ERR_UNSUPPORTED;
  if ((long long)N * Hu * Wu >= (1ll << 31) || (long long)N * Hv * Wv >= (1ll << 31)) return SPK_ERR_UNSUPPORTED;
  if (ws_bytes < spk_conv_train_wgrad_ws_bytes(N, Hv, Wv, Cu, Cv, k)) return SPK_ERR_ARG;
  if (!gb_out_or_null) bias_from = 0;
  const long long Ms = (long long)N * Hv * Wv;
  WArgs a{u_cl, v_cl, ws, N, Hu, Wu, Cu, Hv, Wv, Cv, k, stride, pad, bias_from, wgrad_nwg(kind, Ms), wgrad_psize(kind, Cu, Cv, k),
          gw_out, gb_out_or_null, g_tap, g_u, g_v};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (kind == 1) {
    const int ntile = k * k * ((Cu + 31) / 32) * ((Cv + 31) / 32);
    const int ntw = (ntile + 3) / 4;
    // LDS-staged form where its tiles fit (Cu, Cv multiples of 4, 16-byte aligned tensors, at most WL_MAXCH chunks per thread and block)
    bool lds_form = false;
    if (CT_WGRAD_LDS && ntw <= 5 && (Cu & 3) == 0 && (Cv & 3) == 0 && ((reinterpret_cast<uintptr_t>(u_cl) | reinterpret_cast<uintptr_t>(v_cl)) & 15) == 0) {
      WLGeo g{};
      g.WP = Wv * stride + k;
      if (g.WP >= Wu + pad) {                        // (the row image holds every input column)
        const int NR = N * Hv;
        g.rows_per = (NR + a.nwg * 2 - 1) / (a.nwg * 2);
        const size_t comb = (size_t)4 * ntw * 16 * 64 * 4;
        for (int RB = 4; RB >= 1 && !lds_form; --RB) {
          if (RB > 1 && (RB - 1) * Wv >= 24) continue;               // no more rows than ~24 positions need
          g.RB = RB;
          g.svb = RB * (Wv + 1) * Cv;
          g.sub = RB * k * g.WP * Cu;
          g.nv4 = RB * Wv * Cv / 4;
          g.nu4 = RB * k * Wu * Cu / 4;
          const size_t need = (size_t)(CT_WL_PINGPONG ? 2 : 4) * (g.svb + g.sub) * 4;
          if (need <= 150 * 1024 && g.nv4 + g.nu4 <= 256 * WL_MAXCH) {
            g.nb = (g.rows_per + RB - 1) / RB;
            const size_t lds_ = need > comb ? need : comb;
#define SPK_WL_LAUNCH(N_)                                                                                              \
  do {                                                                                                                 \
    static std::once_flag once_;                                                                                       \
    std::call_once(once_, [] {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_train_wgrad_lds_kernel<N_>),                                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);                               \
    });                                                                                                                \
    hipLaunchKernelGGL((conv_train_wgrad_lds_kernel<N_>), dim3(a.nwg), dim3(512), lds_, s, a, g);                      \
  } while (0)
            switch (ntw) {
              case 1: SPK_WL_LAUNCH(1); break;
              case 2: SPK_WL_LAUNCH(2); break;
              case 3: SPK_WL_LAUNCH(3); break;
              case 4: SPK_WL_LAUNCH(4); break;
              default: SPK_WL_LAUNCH(5); break;
            }
#undef SPK_WL_LAUNCH
            lds_form = true;
          }
        }
      }
    }
    if (!lds_form) {
#define SPK_WG_LAUNCH(N_, S_)                                                                                          \
  do {                                                                                                                 \
    const size_t lds_ = (S_) == 2 ? (size_t)4 * (N_) * 16 * 64 * 4 : 8192;                                             \
    static std::once_flag once_;                                                                                       \
    std::call_once(once_, [] {                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_train_wgrad_kernel<N_, S_>),                                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);                               \
    });                                                                                                                \
    hipLaunchKernelGGL((conv_train_wgrad_kernel<N_, S_>), dim3(a.nwg), dim3(256 * (S_)), lds_, s, a);                  \
  } while (0)
    switch (ntw) {
      case 1: SPK_WG_LAUNCH(1, 2); break;
      case 2: SPK_WG_LAUNCH(2, 2); break;
      case 3: SPK_WG_LAUNCH(3, 2); break;
      case 4: SPK_WG_LAUNCH(4, 2); break;
      case 5: SPK_WG_LAUNCH(5, 2); break;
      case 6: SPK_WG_LAUNCH(6, 1); break;
      case 7: SPK_WG_LAUNCH(7, 1); break;
      case 8: SPK_WG_LAUNCH(8, 1); break;
      default: SPK_WG_LAUNCH(9, 1); break;
    }
#undef SPK_WG_LAUNCH
    }
    SPK_LAUNCH_CHECK();
    const int CB = bias_from == 0 ? 0 : (bias_from == 1 ? Cv : Cu);
    const int E = k * k * Cu * Cv + CB;
    hipLaunchKernelGGL(conv_train_wgrad_reduce_kernel<0>, dim3((E + 31) / 32), dim3(256), 0, s, a);
  } else {
    size_t lds = (size_t)4 * (k * k * Cu + 1) * Cv * 4;
    if (lds > 64 * 1024) return SPK_ERR_UNSUPPORTED;
    lds = lds < (size_t)Cu * 1024 ? (size_t)Cu * 1024 : lds;
    if (Cu == 1) hipLaunchKernelGGL((conv_train_c1_wgrad_kernel<1, MAX_TAPS>), dim3(a.nwg), dim3(256), lds, s, a);
    else if (Cu == 2) hipLaunchKernelGGL((conv_train_c1_wgrad_kernel<2, 9>), dim3(a.nwg), dim3(256), lds, s, a);
    else if (Cu == 3) hipLaunchKernelGGL((conv_train_c1_wgrad_kernel<3, 9>), dim3(a.nwg), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_train_c1_wgrad_kernel<4, 9>), dim3(a.nwg), dim3(256), lds, s, a);
    SPK_LAUNCH_CHECK();
    const int CB = bias_from == 0 ? 0 : (bias_from == 1 ? Cv : Cu);
    const int E = k * k * Cu * Cv + CB;
    hipLaunchKernelGGL(conv_train_wgrad_reduce_kernel<1>, dim3((E + 7) / 8), dim3(256), 0, s, a);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
