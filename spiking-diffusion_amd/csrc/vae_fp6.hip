// The spike-input layers of the spiking VQ-VAE on the block-scaled fp6 x fp4 MFMA (R/snn_model/vae_model.py:115-118 Encoder
// conv2, :139-150 Decoder convT1 / convT2), for calls that start from the reset state.  Same arithmetic
// contract as den_mfma_fp6v2.hip: 29-bit per-channel fixed-point weights as radix-32 digits, adjacent digits sharing an fp32
// accumulator through the per-block scales, every spike decision certified against an error bound, the few neurons that come
// within the bound of the threshold recomputed exactly (all six digits, 64-bit sums, fp64 recombination, one rounding) by a
// tail launch.  Unflagged neurons provably emit the exact path's spikes; the result is the one spk_conv_mfma_fused_fwd (int8
// gather kernel) produces for the same layer.  Round 4: FOUR digits on the matrix cores (SPK_VT_D4: two accumulators per tile,
// 4 / 2 instead of 5 / 3 MFMAs and weight-tile reads per tap and chunk) behind the COUNTED bound of den_mfma_fp6v2.hip -- the
// active inputs of every input record are popcounted once per item, then per class and output position the maximum over the
// steps of the sum over the class's taps multiplies the per-input bound of the two dropped digits.  Measured at B = 1024 (same
// box, alternating): convT2 0.246-0.249 -> 0.222-0.228 ms, conv2 (nine taps; no longer spills) 0.118-0.127 -> 0.075-0.084 ms,
// encode -> decode 1.41-1.43 -> 1.57-1.58 M images/s; 0 mismatches against the int8 gather kernel over 7.2e8 neuron-steps.
//
// What differs from the denoiser kernel is the geometry: K is small (16 .. 64 input channels = 1 or 2 chunks of 32) and the
// spatial extent large, so ALL weight tiles of a 32-channel output group stay in LDS for the whole launch (9 taps x [pair 01,
// pair 23] per chunk: 27 / 54 KB; the fifth-digit tiles of the packed format stay in memory) next to the input rows of one item.
//   GEO 0  ConvTranspose2d(k3, s2, p1, op1): four sub-pixel classes (oy % 2, ox % 2) with 1 / 2 / 2 / 4 contributing taps; the
//          rows of a 32-row MFMA tile are two consecutive positions of ONE class x 16 time steps, so a tile's tap list is
//          compile-time.  Item = an image (or its upper / lower half) : class rows + one more input row, zero column on the right.
//   GEO 1  Conv2d(k3, s2, p1): rows = two consecutive OUTPUT positions; item = an image with a zero row above and a zero column left.
// The eight waves of a workgroup (two per SIMD) take passes of two tiles -- a weight tile read from LDS serves both -- and the
// pass list of an item (class-major) is dealt round-robin, so every wave gets a mix of cheap and expensive classes.  These
// layers are bound by the vector work of the LIF scan (16 steps x ~11 instructions per neuron), not by the matrix pipe or memory.
//
// Spikes in: "S32" [B][Cin/32 (rounded up)][H*W][16][16 B] (fp4 nibbles; channels beyond Cin must be zero).  Out: S32, plain u8
// PTC [B][Ho*Wo][16][Cout], or -- for the layer in front of the linear read-out -- the time-collapsed tensor
// m = sum_t coef[t] * s_t (fp32 [B][Ho*Wo][Cout], spk_readout_collapsed_fwd): the spike frames are then never stored.
#include "den_common.h"
#include "../../include/spkdiff.h"
#include <math.h>
#include <type_traits>
#include <utility>

namespace {

typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int T16 = 16;
constexpr int POSB = 256;                    // bytes per position and 32-channel chunk: 16 steps x 16 B
constexpr int WT = 1536;                     // one B tile: 64 lanes x 32 six-bit codes
constexpr unsigned FLAG_CAP = 1u << 20;
constexpr int OUT_COLLAPSED = 0, OUT_S32 = 1, OUT_PTC = 2;

__host__ __device__ constexpr int tiles_per_tap(int nch) { return nch == 2 ? 5 : 3; }
// tile (tap, j): NCH = 2: j = 0..3: chunk j / 2, digit pair j % 2; j = 4: fifth digit, K half 0 = chunk 0, half 1 = chunk 1.
//                NCH = 1: j = 0, 1: digit pair j; j = 2: fifth digit in K half 0 (half 1 zero).

struct TArgs {
  const uint8_t* in;                         // S32 [B][NCH][H*W][16][16 B]
  const uint8_t* wq; const double* scale; const double* bias; const float* bn_a; const float* bn_b; const float* coef;
  void* out;
  // flags[0]: live count of flagged neurons (zero between calls), flags[1]: the count published by the main launch's last
  // workgroup for the repair launch, flags[2 .. 2 + cap): their ids, then the overflow bitmap, then the ticket of that hand-over
  unsigned* flags; unsigned flag_cap; long long ticket_idx; const int* qtab;       // qtab int32 [Cout][9][Cin]
  int B, Cout, Cin;
};

template <typename F, int... S>
__device__ __forceinline__ void tfor_impl(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void tfor(F&& f) {
  tfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr float CERT_4EPS = 4.0f * 2.38418579e-07f;
#ifndef SPK_VT_EXEC_SCAN
#define SPK_VT_EXEC_SCAN 1      // 0: the collapsed-output scan handles a spike with compare + selects (the compiler's form)
#endif
#ifndef SPK_VT_NWV
#define SPK_VT_NWV 16           // waves per workgroup.  Round 4: SIXTEEN (four per SIMD) with one tile per pass: the four-digit form needs ~100
                                // registers at SPK_VT_TPP = 1, four waves issue vector instructions at 2.0 instead of 3.1 cycles each
                                // (tools/coexec_probe.hip) and overlap one another's multiply phases: encode -> decode at B = 1024
                                // 1.577 -> 1.632-1.640 M images/s, convT2 0.225 -> 0.213 ms (same box; 12 waves: 1.61 M).  Round 3 (five
                                // digits): 16 waves x 1 tile 253-255 us against 259-267 for convT2 and spills in the nine-tap layer
#endif
#ifndef SPK_VT_PIPE
#define SPK_VT_PIPE 0           // 1: software-pipelined LDS reads in the multiply phase (weight tile two steps ahead, next tap's spike
                                // fragments one tap ahead, schedule pinned); measured equal (264-267 against 267 us): hipcc already defers the
                                // second tile's MFMAs to after the first tile's scan and runs them back to back from registers
#endif
#ifndef SPK_VT_PFB
#define SPK_VT_PFB 2            // (SPK_VT_PIPE) how many steps ahead a weight tile is requested
#endif
#ifndef SPK_VT_STAGGER
#define SPK_VT_STAGGER 0        // s_sleep argument (x 64 cycles) of waves 4..7 after every item barrier.  Measured (convT2, B = 1024): 8 / 16 / 24 /
                                // 32 / 48 -> 273 / 270-274 / 273 / 273 / 275 us against 266-267: a wave that scans while its partner multiplies
                                // issues its vector instructions at the single-wave rate (6.3 instead of 3.1 cycles): nothing is gained
#endif
#ifndef SPK_VT_D4
#define SPK_VT_D4 1             // round 4: FOUR digits on the matrix cores (two accumulators per tile, 4 / 2 instead of 5 / 3 MFMAs and
                                // weight-tile reads per tap) behind the COUNTED certification bound of den_mfma_fp6v2.hip: the two dropped
                                // digits move a pre-activation by at most 528 units of 2^-s per ACTIVE input, and the active inputs of
                                // every output row are counted once per item (popcounts of the input records, then per class and
                                // position the maximum over the steps of the sum over the class's taps).  0: five digits, static bound
#endif
#ifndef SPK_VT_TPP
#define SPK_VT_TPP (SPK_VT_NWV >= 12 ? 1 : 2)   // row tiles per pass (a weight tile read from LDS serves all of them)
#endif
#ifndef SPK_VT_TPP_T
#define SPK_VT_TPP_T SPK_VT_TPP // ... of the transposed (1 / 2 / 2 / 4-tap) layers.  Round 4 (four digits): three tiles per pass fit (256 registers,
                                // 16 B of scratch) and measure the same as two (dec2 0.2275 against 0.2219 / 0.2249 ms): two
#endif
#ifndef SPK_VT_SIGNBITS
#define SPK_VT_SIGNBITS 1       // spike-bit outputs: the sixteen bits of a lane shifted in from the sign of h - 1 (den_mfma_fp6v2.hip, SPK_V2_SIGNBITS):
                                // same speed here (encode -> decode 1.667-1.685 against 1.677-1.678 M images/s), ten registers fewer in the
                                // spike-bit layers and no scratch left in the CIFAR-shaped 8x8 -> 16x16 layer.  0: the select form
#endif
#ifndef SPK_VT_HOIST
#define SPK_VT_HOIST 1          // the weight fragments of a class's FIRST tap stay in registers over the wave's passes of that class within an
                                // item (24 registers with two chunks, 12 with one): the multiply phase of these layers is bound by LDS reads,
                                // 1.5 KB of weight fragments per MFMA (profiles/r4_ab_kernel_variants.txt (11)); 0: every pass reads them
#endif
#ifndef SPK_VT_NHOLD
#define SPK_VT_NHOLD 1          // how many of a class's leading taps are held that way (SPK_VT_HOIST).  All four taps of the four-tap class need
                                // 96 registers with two chunks: twelve waves (168 registers each) hold three, sixteen (128) hold one
#endif
#ifndef SPK_VT_DBG
#define SPK_VT_DBG 0            // timing experiments only (results are wrong): 1 = no MFMAs, 2 = no LIF scan, 4 = no weight-tile reads from LDS
#endif

__device__ __forceinline__ unsigned spread8_v(unsigned x) {        // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
  x = (x | (x << 12)) & 0x000f000fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}

template <int GEO, int H, int W>
struct Geo {
  static constexpr int Ho = GEO == 0 ? 2 * H : H / 2, Wo = GEO == 0 ? 2 * W : W / 2;
  static constexpr int KMAX = GEO == 0 ? 4 : 9;                       // taps that can reach one output
};

// the taps that reach an output of sub-pixel class CLS (transposed convolution: 1 / 2 / 2 / 4 of nine; plain stride 2: all)
template <int GEO, int CLS>
__host__ __device__ constexpr bool tap_is_on(int tap) {
  const int ky = tap / 3, kx = tap % 3, py = CLS >> 1, px = CLS & 1;
  return GEO == 1 || ((py == 0 ? ky == 1 : ky != 1) && (px == 0 ? kx == 1 : kx != 1));
}
template <int GEO, int CLS>
__host__ __device__ constexpr int n_on_taps() {
  int n = 0;
  for (int t = 0; t < 9; ++t) n += tap_is_on<GEO, CLS>(t) ? 1 : 0;
  return n;
}
template <int GEO, int CLS>
__host__ __device__ constexpr int on_tap(int k) {
  for (int t = 0; t < 9; ++t)
    if (tap_is_on<GEO, CLS>(t) && k-- == 0) return t;
  return 0;
}

#ifndef SPK_VT_DYN
#define SPK_VT_DYN 1            // round 5: the passes of an item are HANDED OUT (an LDS counter, class-major order kept) instead of dealt
                                // round-robin.  tools/vae_phase.py: of the four waves of a SIMD the oldest issues its scan's vector
                                // instructions first -- waves 0-3 scan a pass in 1 280 cycles, waves 12-15 in 2 410 -- so with equal
                                // shares the first four waited 31 % of the launch at the item's end barrier (mean over the waves: 19 %)
#endif
#ifndef SPK_VT_PREFETCH
#define SPK_VT_PREFETCH 0       // 1 (single-slab layers): one dword of every 128-byte line of the NEXT item's input rows is requested at the start of
                                // this item's passes, so that the copy behind the end barrier finds them in the L2
#endif
#ifndef SPK_VT_PRIO
#define SPK_VT_PRIO 0           // s_setprio of a wave in its multiply phase (0 in the scan): the MFMAs and their LDS reads go first
#endif
#ifndef SPK_VT_STAMP
#define SPK_VT_STAMP 0          // 1 (timing builds only): workgroup 0 accumulates shader-clock cycles per wave and phase into g_vt_stamp
                                // (spk_vt_stamps reads and clears it; tools/vae_phase.py)
#endif
#if SPK_VT_STAMP
__device__ unsigned long long g_vt_stamp[SPK_VT_NWV][8];
#define VT_CLK() ((long long)__builtin_readcyclecounter())
#define VT_ACC(slot, t0) do { const long long t1_ = VT_CLK(); st_acc[slot] += t1_ - (t0); (t0) = t1_; } while (0)
#else
#define VT_ACC(slot, t0) do { } while (0)
#endif

#ifndef SPK_VT_INWAVE_FIX
#define SPK_VT_INWAVE_FIX 0     // 1 (round 6 experiment, measured SLOWER): a wave recomputes the neurons IT flags, exactly, right behind the tile's
                                // stores (an out-of-line call on a path 0.6 % of the tiles take) -- no id list, no repair launch.  Bit-equal (28
                                // vae_fp6 tests), and the three repair launches (8 + 8 + 6 us) go away, but the main launches grow by more:
                                // convT2 198.8 -> 209.8, convT1 105.9 -> 110.5, conv2 65.3 -> 79.7 us; encode -> decode 0.555 -> 0.575 ms, same
                                // box (profiles/r6_ab_kernel_variants.txt (3)).  0: id list + overflow bitmap + vae_fp6_fixup_kernel
#endif
// (the out-of-line form of vae_fix_neuron: the hot kernel then pays for the call only on the rare path)
template <int GEO, int H, int W, int NCH, int OUT>
__device__ __attribute__((noinline)) void vae_fix_neuron_call(const uint8_t* in, const int* qtab, const double* scale, const double* bias,
                                                              const float* bn_a, const float* bn_b, const float* coef, void* out,
                                                              int Cin, int Cout, long long nid, int lane);

template <int GEO, int H, int W, int NCH, int OUT, int SPLIT, bool DB>
__global__ __launch_bounds__(SPK_VT_NWV * 64, 1) void vae_fp6_kernel(TArgs a) {
#if SPK_VT_STAMP
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // 0 stage+wait+barrier, 1 popcount phases, 2 multiply, 3 scan+store, 4 end barrier, 5 hold reads, 6 passes
  long long st_t = VT_CLK();
  const long long st_begin = st_t;
#endif
  constexpr int TPT = tiles_per_tap(NCH), W_BYTES = 9 * TPT * WT;
  constexpr int TPL = SPK_VT_D4 ? TPT - 1 : TPT;            // tiles per tap kept in LDS (the fifth-digit tile stays in memory)
  constexpr int WL_BYTES = 9 * TPL * WT;
  constexpr int RQ = GEO == 0 ? H / SPLIT : H / 2;          // rows of positions per item (class rows / output rows)
  constexpr int CW = GEO == 0 ? W : W / 2;                  // positions per row
  constexpr int NPOS = RQ * CW, NTC = (NPOS + 1) / 2, TPP = GEO == 0 ? SPK_VT_TPP_T : SPK_VT_TPP, NPASS = (NTC + TPP - 1) / TPP,
                NCLS = GEO == 0 ? 4 : 1;
  constexpr int SROWS = GEO == 0 ? RQ + 1 : H + 1, SCOLS = W + 1;
  constexpr int A_CH = SROWS * SCOLS * POSB, A_BYTES = NCH * A_CH, NBUF = DB ? 2 : 1;
  constexpr int PPR = (W + 3) / 4;                          // 1 KiB DMA pieces per image row
  constexpr int DROWS = GEO == 0 ? SROWS : H;               // image rows copied per item
  constexpr int Ho = Geo<GEO, H, W>::Ho, Wo = Geo<GEO, H, W>::Wo;
  static_assert(GEO == 0 ? (H % SPLIT) == 0 : ((H % 2) == 0 && (W % 2) == 0 && SPLIT == 1), "item geometry");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const sA = lds;
  uint8_t* const sW = lds + NBUF * A_BYTES;
  // four-digit form: active inputs per input cell and step (u8, borders zero) and, per class and position, their maximum over
  // the steps of the sum over the class's taps
  constexpr int NCELL = SROWS * SCOLS;
  uint8_t* const s_cin = lds + NBUF * A_BYTES + WL_BYTES;                       // [NCELL][16]
  int* const s_nmax = reinterpret_cast<int*>(s_cin + ((NCELL * 16 + 15) & ~15));   // [NCLS][NPOS]
  constexpr bool DYN = SPK_VT_DYN && GEO == 0;              // (the 25 passes of the plain stride-2 layer's item: dealt; handed out it measured 3 us slower)
  int* const s_ctr = SPK_VT_D4 ? s_nmax + NCLS * NPOS : reinterpret_cast<int*>(s_cin);   // the item's next pass (SPK_VT_DYN)
  const unsigned sA_addr = spk_lds_addr(sA);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.Cout >> 5;
  const int g = blockIdx.x % G, il = blockIdx.x / G, lanes = gridDim.x / G;

  {  // the group's weight tiles and zeroed input slabs (their borders stay zero for the whole launch)
    const uint4* src = reinterpret_cast<const uint4*>(a.wq + (long long)g * W_BYTES);
    for (int i = tid; i < WL_BYTES / 16; i += SPK_VT_NWV * 64) {
      const int tile = i / (WT / 16), o = i - tile * (WT / 16);
      reinterpret_cast<uint4*>(sW)[i] = src[((tile / TPL) * TPT + tile % TPL) * (WT / 16) + o];
    }
    for (int i = tid; i < NBUF * A_BYTES / 16; i += SPK_VT_NWV * 64) reinterpret_cast<uint4*>(sA)[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  const int co = g * 32 + (lane & 31);
  const float scale_f = (float)a.scale[co], bias_f = (float)a.bias[co];
  const float bna = a.bn_a[co], bnb = a.bn_b[co];
  const float Ac = 32.0f * scale_f * bna;                   // z = Q5 * Ac + Bc,  Q5 = P01 * 2^15 + P23 * 2^5 + P4
  const float Ac0 = Ac * 32768.0f, Ac1 = Ac * 32.0f;        // (powers of two: exact)
  const float Bc = fmaf(bias_f, bna, bnb);
  // Certification (den_mfma_fp6v2.hip): the approximate and the exact pre-activation differ by at most cE + 2 eps |z|:
  // the dropped sixth digit moves a pre-activation by at most 16 units of 2^-s per active input (at most KMAX taps x Cin
  // inputs reach an output), and the three-term fp32 recombination z = P01 * Ac0 + (P23 * Ac1 + (P4 * Ac + Bc)) rounds partial
  // sums bounded by the middle / low digit groups (|32 d2 + d3| <= 528, |d4| <= 16 per input) rather than by |z|.
  const float kin = (float)(Geo<GEO, H, W>::KMAX * a.Cin);
  const float E5 = 16.0f * kin * scale_f;
  const float part_max = (528.0f * fabsf(Ac1) + 16.0f * fabsf(Ac)) * kin + fabsf(Bc);
  const float cE = SPK_VT_D4 ? 2.0f * 2.38418579e-07f * (fabsf(bnb) + fabsf(Bc)) + 1e-30f
                             : fabsf(bna) * E5 + 2.0f * 2.38418579e-07f * (fabsf(bnb) + fabsf(Bc) + 2.0f * part_max) + 1e-30f;
  // four digits (den_mfma_fp6v2.hip, "Certification"): z = Q4 * Ac4 + Bc, Q4 = P01 * 2^10 + P23; the dropped digits move z_t by at
  // most cT * n_t (|32 d4 + d5| <= 528 units of 2^-s per active input), every fp32 rounding is inside the eps terms
  const float cT = 528.0f * scale_f * fabsf(bna) * 1.000001f;
  const float Ac4 = 1024.0f * scale_f * bna;
  float coef[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) coef[r] = OUT == OUT_COLLAPSED ? a.coef[r] : 0.f;

  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  const int sc_a = 0x7f7f7f7f;
  const int sc_p = half ? (int)0x82828282u : (int)0x87878787u;     // even digit (K half 0) x 2^8, odd digit x 2^3
  const int sc_4 = (int)0x82828282u;                                // fifth digit x 2^3
  const unsigned lane16 = (unsigned)lane * 16u;
  const int nitems = SPLIT * a.B;

  // copy the input rows of item `itm` into slab `buf` (asynchronous; spk_dma_wait_all + barrier before use)
  auto stage = [&](int itm, int buf) {
    const int b = itm / SPLIT, part = itm - b * SPLIT;
    const unsigned dst0 = sA_addr + buf * A_BYTES;
    for (int id = wave; id < NCH * DROWS * PPR; id += SPK_VT_NWV) {
      const int c = id / (DROWS * PPR), rr = (id / PPR) % DROWS, px4 = id % PPR;
      const int iy = GEO == 0 ? part * RQ + rr : rr;
      if (iy < H) {
        const int np = (W - 4 * px4) < 4 ? (W - 4 * px4) : 4;
        const unsigned long long mask = np == 4 ? ~0ull : ((1ull << (16 * np)) - 1ull);
        const uint8_t* src = a.in + (((long long)b * NCH + c) * H * W + iy * W + 4 * px4) * POSB;
        const int cell = GEO == 0 ? rr * SCOLS + 4 * px4 : (rr + 1) * SCOLS + 1 + 4 * px4;
        spk_dma16s_masked(src, lane16, dst0 + c * A_CH + cell * POSB, mask);
      }
    }
    if (GEO == 0 && part * RQ + SROWS - 1 >= H) {           // the row below the image: zeros
      for (int i = tid; i < NCH * W * POSB / 16; i += SPK_VT_NWV * 64) {
        const int c = i / (W * POSB / 16), o = i % (W * POSB / 16);
        reinterpret_cast<uint4*>(sA + buf * A_BYTES + c * A_CH + (SROWS - 1) * SCOLS * POSB)[o] = make_uint4(0, 0, 0, 0);
      }
    }
  };

  if (DB && il < nitems) stage(il, 0);
  int n = 0;
  for (int itm = il; itm < nitems; itm += lanes, ++n) {
    const int b = itm / SPLIT, part = itm - b * SPLIT;
    const int buf = DB ? (n & 1) : 0;
    if (!DB) stage(itm, 0);
    spk_dma_wait_all();
    __syncthreads();
    if (DB && itm + lanes < nitems) stage(itm + lanes, buf ^ 1);
    if constexpr (DYN) {
      // (every wave took its last pass number of the previous item before the barrier above; the first one of this item is taken
      //  behind the barriers of the counting phases below)
      if (tid == 0) *s_ctr = SPK_VT_NWV;
      if constexpr (!SPK_VT_D4) __syncthreads();
    }
    VT_ACC(0, st_t);
#if SPK_VT_STAGGER > 0
    // The two waves of a SIMD run the same pass list from the same barrier, so both are in their multiply phase together (each
    // then sees the matrix pipe at half rate: stamped 65-70 cycles per MFMA) and in their scans together.  The second wave
    // starts every item this many x 64 cycles late: its multiply phases then fall into the first one's scans.
    if (wave >= SPK_VT_NWV / 2) __builtin_amdgcn_s_sleep(SPK_VT_STAGGER);
#endif
    const uint8_t* const A0 = sA + buf * A_BYTES;

    if constexpr (SPK_VT_D4) {
      // (A) active inputs of every input record (cell, step), summed over the chunks: a spike is the nibble 0x2 = one set bit
      for (int r = tid; r < NCELL * 16; r += SPK_VT_NWV * 64) {
        int n = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const v4i q = *reinterpret_cast<const v4i*>(A0 + c * A_CH + r * 16);
          n += __builtin_popcount((unsigned)q[0]) + __builtin_popcount((unsigned)q[1]) + __builtin_popcount((unsigned)q[2]) +
               __builtin_popcount((unsigned)q[3]);
        }
        s_cin[r] = (uint8_t)n;                                 // <= 64
      }
      __syncthreads();
      // (B) per class and output position: max over the steps of the sum over the class's taps (what the bound multiplies cT by)
      for (int e = tid; e < NCLS * NPOS; e += SPK_VT_NWV * 64) {
        const int cls = e / NPOS, p = e - cls * NPOS, ry = p / CW, rx = p - ry * CW;
        const int cell0 = GEO == 0 ? ry * SCOLS + rx : 2 * ry * SCOLS + 2 * rx;
        const int py = cls >> 1, px = cls & 1;
        unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};   // sixteen 16-bit sums (even / odd bytes of the four words)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int ky = tap / 3, kx = tap % 3;
          const bool on = GEO == 1 || ((py == 0 ? ky == 1 : ky != 1) && (px == 0 ? kx == 1 : kx != 1));
          const int dy = GEO == 1 ? ky : ((py == 1 && ky == 0) ? 1 : 0), dx = GEO == 1 ? kx : ((px == 1 && kx == 0) ? 1 : 0);
          if (on) {
            const v4i q = *reinterpret_cast<const v4i*>(s_cin + (cell0 + dy * SCOLS + dx) * 16);
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
              lo[w4] += (unsigned)q[w4] & 0x00ff00ffu;
              hi[w4] += ((unsigned)q[w4] >> 8) & 0x00ff00ffu;
            }
          }
        }
        unsigned mx = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          mx = max(mx, max(lo[w4] & 0xffffu, lo[w4] >> 16));
          mx = max(mx, max(hi[w4] & 0xffffu, hi[w4] >> 16));
        }
        s_nmax[e] = (int)mx;
      }
      __syncthreads();
    }
    VT_ACC(1, st_t);
    if constexpr (SPK_VT_PREFETCH && !DB) {
      if (itm + lanes < nitems) {
        const int nb_ = (itm + lanes) / SPLIT, np_ = (itm + lanes) - nb_ * SPLIT;
        const int iy0 = GEO == 0 ? np_ * RQ : 0;
        const int nrows = (iy0 + DROWS <= H ? DROWS : H - iy0), lpc = nrows * (W * POSB / 128);   // rows are contiguous within a chunk
        for (int l = tid; l < NCH * lpc; l += SPK_VT_NWV * 64) {
          const int c = l / lpc, o = l - c * lpc;
          (void)*reinterpret_cast<const volatile unsigned*>(a.in + (((long long)nb_ * NCH + c) * H * W + iy0 * W) * POSB + (long long)o * 128);
        }
      }
    }

    // hb: the weight fragments of the class's first tap, read once per class by the caller (SPK_VT_HOIST)
    auto run_pass = [&](auto cls_tag, int k, const v6i (&hb)[SPK_VT_NHOLD][NCH * 2]) __attribute__((always_inline)) {
      constexpr int CLS = decltype(cls_tag)::value, PY = CLS >> 1, PX = CLS & 1;
      constexpr int NHELD = n_on_taps<GEO, CLS>() < SPK_VT_NHOLD ? n_on_taps<GEO, CLS>() : SPK_VT_NHOLD;
      auto held_slot = [](int tap) constexpr {                // which held set carries this tap's fragments (-1: none)
        for (int h = 0; h < NHELD; ++h)
          if (on_tap<GEO, CLS>(h) == tap) return h;
        return -1;
      };
      int tl[TPP];                                         // tiles of the pass; one past the end repeats the first (computed, dropped)
      bool tv[TPP];
#pragma unroll
      for (int i = 0; i < TPP; ++i) {
        tv[i] = TPP * k + i < NTC;
        tl[i] = tv[i] ? TPP * k + i : TPP * k;
      }
      int base[TPP];
#pragma unroll
      for (int i = 0; i < TPP; ++i) {
        int p = 2 * tl[i] + hsel;
        p = p < NPOS ? p : NPOS - 1;
        const int ry = p / CW, rx = p - ry * CW;
        base[i] = (GEO == 0 ? ry * SCOLS + rx : 2 * ry * SCOLS + 2 * rx) * POSB + tt * 16;
      }
      if (SPK_VT_PRIO) __builtin_amdgcn_s_setprio(SPK_VT_PRIO);
      constexpr int NACC = SPK_VT_D4 ? 2 : 3;
      v16f acc[TPP][NACC];
#pragma unroll
      for (int i = 0; i < TPP; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      auto ldb = [&](int tile) -> v6i {
        if (SPK_VT_DBG & 4) return v6i{lane + tile, lane, tile, 0x11111111, lane * 3, 0x01010101};   // (timing only: no weight reads)
        const uint8_t* p = sW + tile * WT;
        const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
        const v2i y = *reinterpret_cast<const v2i*>(p + 1024 + lane * 8);
        return v6i{x[0], x[1], x[2], x[3], y[0], y[1]};
      };
      // (the builtin, not inline assembly: hipcc then places the hazard wait states between an MFMA and the reads of its
      //  accumulator itself -- an asm form measured slower and raced)
      auto mm = [&](v16f& d, const v4i& av, const v6i& bv, int sb) {
        const v8i a8 = {av[0], av[1], av[2], av[3], 0, 0, 0, 0};
        const v8i b8 = {bv[0], bv[1], bv[2], bv[3], bv[4], bv[5], 0, 0};
        d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d, 4, 2, 0, sc_a, 0, sb);
      };
#if SPK_VT_PIPE
      // The pass's products as ONE compile-time list of steps (an active tap of the class x one weight tile), software pipelined:
      // the weight tile of step s + PFB and the spike fragments of the NEXT tap are requested from LDS while the MFMAs of step s run
      // (tap by tap -- read, wait, multiply -- a pass spends 1 000 - 1 500 cycles on 288 cycles of matrix time: tools/vae_phase.py).
      // Steps of a held tap (SPK_VT_HOIST) take their tile from hb and request nothing.
      constexpr int NON = n_on_taps<GEO, CLS>(), TS = TPL, NSTEP = NON * TS, PFB = SPK_VT_PFB, RB = PFB + 1;
      auto toff = [&](auto k_tag) {
        constexpr int TAP = on_tap<GEO, CLS>(decltype(k_tag)::value), KY = TAP / 3, KX = TAP % 3;
        constexpr int DY = GEO == 1 ? KY : ((PY == 1 && KY == 0) ? 1 : 0), DX = GEO == 1 ? KX : ((PX == 1 && KX == 0) ? 1 : 0);
        return std::integral_constant<int, (DY * SCOLS + DX) * POSB>{};
      };
      auto step_held = [](int st) constexpr {                 // the held set of step st's tap (-1: its tile comes from LDS)
        if (!(SPK_VT_HOIST && SPK_VT_D4)) return -1;
        for (int h = 0; h < NHELD; ++h)
          if (h == st / TS) return h;                          // (the held taps are the class's leading ones)
        return -1;
      };
      v4i av[2][TPP][NCH];
      v6i bq[RB];
      auto lda = [&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value, TOFF = decltype(toff(k_tag))::value;
#pragma unroll
        for (int i = 0; i < TPP; ++i)
#pragma unroll
          for (int c = 0; c < NCH; ++c) av[k & 1][i][c] = *reinterpret_cast<const v4i*>(A0 + c * A_CH + base[i] + TOFF);
      };
      lda(std::integral_constant<int, 0>{});
      tfor<(PFB < NSTEP ? PFB : NSTEP)>([&](auto s_tag) {
        constexpr int st = decltype(s_tag)::value;
        if constexpr (step_held(st) < 0) bq[st % RB] = ldb(on_tap<GEO, CLS>(st / TS) * TPL + st % TS);
      });
      tfor<NSTEP>([&](auto s_tag) {
        constexpr int st = decltype(s_tag)::value, k = st / TS, j = st % TS;
        if constexpr (st + PFB < NSTEP && step_held(st + PFB) < 0)
          bq[(st + PFB) % RB] = ldb(on_tap<GEO, CLS>((st + PFB) / TS) * TPL + (st + PFB) % TS);
        if constexpr (j == 0 && k + 1 < NON) lda(std::integral_constant<int, k + 1>{});
        asm volatile("" ::: "memory");                       // (every read stays where it is written)
        constexpr int HS = step_held(st);
        const v6i bv = HS >= 0 ? hb[HS >= 0 ? HS : 0][j] : bq[st % RB];
#pragma unroll
        for (int i = 0; i < TPP; ++i) {
          if (SPK_VT_DBG & 1) continue;
          if constexpr (NCH == 2) {
            if constexpr (j < 4) mm(acc[i][j & 1], av[k & 1][i][j >> 1], bv, sc_p);
            else mm(acc[i][NACC - 1], half ? av[k & 1][i][1] : av[k & 1][i][0], bv, sc_4);
          } else {
            mm(acc[i][j < 2 ? j : NACC - 1], av[k & 1][i][0], bv, j < 2 ? sc_p : sc_4);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#else
      tfor<9>([&](auto tap_tag) {
        constexpr int TAP = decltype(tap_tag)::value, KY = TAP / 3, KX = TAP % 3;
        // GEO 0: oy = 2 iy - 1 + ky: class parity PY takes ky = 1 (iy = qy) when even, ky = 0 (iy = qy + 1) and ky = 2 (iy = qy) when odd
        constexpr bool ON = GEO == 1 || ((PY == 0 ? KY == 1 : KY != 1) && (PX == 0 ? KX == 1 : KX != 1));
        if constexpr (ON) {
          constexpr int DY = GEO == 1 ? KY : ((PY == 1 && KY == 0) ? 1 : 0), DX = GEO == 1 ? KX : ((PX == 1 && KX == 0) ? 1 : 0);
          constexpr int TOFF = (DY * SCOLS + DX) * POSB;
          if constexpr (NCH == 2) {
            v4i av[TPP][2];
#pragma unroll
            for (int i = 0; i < TPP; ++i) {
              av[i][0] = *reinterpret_cast<const v4i*>(A0 + base[i] + TOFF);
              av[i][1] = *reinterpret_cast<const v4i*>(A0 + A_CH + base[i] + TOFF);
            }
            constexpr int HS = held_slot(TAP);
            constexpr bool HB = SPK_VT_HOIST && SPK_VT_D4 && HS >= 0;
            constexpr int HQ = HB ? HS : 0;
            const v6i b0 = HB ? hb[HQ][0] : ldb(TAP * TPL + 0), b1 = HB ? hb[HQ][1] : ldb(TAP * TPL + 1),
                      b2 = HB ? hb[HQ][2] : ldb(TAP * TPL + 2), b3 = HB ? hb[HQ][3] : ldb(TAP * TPL + 3);
            v6i b4 = b0;
            if constexpr (!SPK_VT_D4) b4 = ldb(TAP * TPL + 4);
#pragma unroll
            for (int i = 0; i < TPP; ++i) {
              if (SPK_VT_DBG & 1) continue;
              mm(acc[i][0], av[i][0], b0, sc_p);
              mm(acc[i][1], av[i][0], b1, sc_p);
              mm(acc[i][0], av[i][1], b2, sc_p);
              mm(acc[i][1], av[i][1], b3, sc_p);
              if constexpr (!SPK_VT_D4) {
                const v4i a4 = half ? av[i][1] : av[i][0];
                mm(acc[i][NACC - 1], a4, b4, sc_4);
              }
            }
          } else {
            v4i av[TPP];
#pragma unroll
            for (int i = 0; i < TPP; ++i) av[i] = *reinterpret_cast<const v4i*>(A0 + base[i] + TOFF);
            constexpr int HS = held_slot(TAP);
            constexpr bool HB = SPK_VT_HOIST && SPK_VT_D4 && HS >= 0;
            constexpr int HQ = HB ? HS : 0;
            const v6i b0 = HB ? hb[HQ][0] : ldb(TAP * TPL + 0), b1 = HB ? hb[HQ][1] : ldb(TAP * TPL + 1);
            v6i b4 = b0;
            if constexpr (!SPK_VT_D4) b4 = ldb(TAP * TPL + 2);
#pragma unroll
            for (int i = 0; i < TPP; ++i) {
              if (SPK_VT_DBG & 1) continue;
              mm(acc[i][0], av[i], b0, sc_p);
              mm(acc[i][1], av[i], b1, sc_p);
              if constexpr (!SPK_VT_D4) mm(acc[i][NACC - 1], av[i], b4, sc_4);
            }
          }
        }
      });
#endif
      if (SPK_VT_PRIO) __builtin_amdgcn_s_setprio(0);
#if SPK_VT_STAMP
      {
        float sink = acc[0][0][15] + acc[0][NACC - 1][15];
        asm volatile("" : "+v"(sink));
        VT_ACC(2, st_t);
        st_acc[6] += 1;
      }
#endif
      // ---- epilogue: fp32 recombination, BN, LIF scan with certification, output
#pragma unroll
      for (int i = 0; i < TPP; ++i) {
        // D_t = D_{t-1} / 2 + cE + 4 eps (|z_t| + |v_{t-1}|) and |v| <= max |z|, so D_t <= 2 cE + 16 eps max_t |z_t| for every t:
        // track max |z| and min |h - 1| (two instructions per step instead of five) and compare once
        float v = 0.f, m = 0.f, zmax = 0.f, dmin = 3.0e38f;
        unsigned mybits = 0;
        if (SPK_VT_DBG & 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) m += acc[i][0][r] + acc[i][1][r] + acc[i][NACC - 1][r];
        } else
#pragma unroll
        for (int r2 = 0; r2 < 16; r2 += 2) {
          // the recombination of two steps at a time on the packed fp32 pipe (adjacent accumulator registers)
          const v2f p0 = {acc[i][0][r2], acc[i][0][r2 + 1]}, p1 = {acc[i][1][r2], acc[i][1][r2 + 1]},
                    p2 = {acc[i][NACC - 1][r2], acc[i][NACC - 1][r2 + 1]};
          v2f z2;
          if constexpr (SPK_VT_D4) {
            const v2f q4 = __builtin_elementwise_fma(p0, (v2f){1024.0f, 1024.0f}, p1);
            z2 = __builtin_elementwise_fma(q4, (v2f){Ac4, Ac4}, (v2f){Bc, Bc});
          } else {
            z2 = __builtin_elementwise_fma(p0, (v2f){Ac0, Ac0},
                 __builtin_elementwise_fma(p1, (v2f){Ac1, Ac1}, __builtin_elementwise_fma(p2, (v2f){Ac, Ac}, (v2f){Bc, Bc})));
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = r2 + e;
            const float z = z2[e];
            zmax = fmaxf(zmax, fabsf(z));
            const float h = fmaf(z - v, 0.5f, v);            // == v + (z - v) * 0.5f: the product is exact
            const float hm = h - 1.0f;
            dmin = fminf(dmin, fabsf(hm));
            if constexpr (OUT == OUT_COLLAPSED && SPK_VT_EXEC_SCAN) {
              // spike = h >= 1: reset v and add the step's coefficient UNDER THE SPIKE MASK (v_cmpx narrows exec, two plain
              // instructions, exec restored): three vector instructions instead of compare + two selects + add (convT2 -5 %;
              // the spike-bit outputs measured slower in this form -- hipcc keeps their sixteen masks in SGPRs -- and keep the C form)
              v = h;
              unsigned long long exec_sv;
              asm volatile("s_mov_b64 %[sv], exec\n\tv_cmpx_le_f32_e32 1.0, %[v]\n\tv_mov_b32_e32 %[v], 0\n\t"
                           "v_add_f32_e32 %[m], %[c], %[m]\n\ts_mov_b64 exec, %[sv]"
                           : [v] "+v"(v), [m] "+v"(m), [sv] "=&s"(exec_sv) : [c] "s"(coef[r]) : "vcc");
            } else {
              const bool s = h >= 1.0f;
              v = s ? 0.0f : h;
              if (OUT == OUT_COLLAPSED) m = m + (s ? coef[r] : 0.f);
              else if constexpr (SPK_VT_SIGNBITS) mybits = __builtin_amdgcn_alignbit(mybits, __float_as_uint(hm), 31);   // (mybits << 1) | sign(h - 1)
              else mybits |= s ? (1u << r) : 0u;
            }
          }
        }
        if constexpr (SPK_VT_SIGNBITS && OUT != OUT_COLLAPSED) {
          if (!(SPK_VT_DBG & 2)) mybits = ~(__builtin_bitreverse32(mybits) >> 16) & 0xffffu;      // bit r = NOT sign(h_r - 1)
        }
        const int p = 2 * tl[i] + half;                       // accumulator lane half == position within the tile
        const bool ok = tv[i] && p < NPOS;
        const int pc = p < NPOS ? p : NPOS - 1;
        // dh <= c + 8 eps max |z| (10 eps: a little to spare); c = cE (five digits, every input active) or cE + cT max_t n_t
        const float cb = SPK_VT_D4 ? fmaf((float)s_nmax[CLS * NPOS + pc], cT, cE) : cE;
        const bool flg = dmin <= fmaf(zmax, 2.5f * CERT_4EPS, cb);
        const int ry = pc / CW, rx = pc - ry * CW;
        const int oy = GEO == 0 ? 2 * (part * RQ + ry) + PY : ry, ox = GEO == 0 ? 2 * rx + PX : rx;
        const long long pos = ((long long)b * Ho + oy) * Wo + ox;
        const long long nid_f = pos * a.Cout + co;
        if (flg && ok) {
          const unsigned idx = atomicAdd(a.flags, 1u);            // (the count of flagged neurons: statistics, tests)
          if constexpr (!SPK_VT_INWAVE_FIX) {
            if (idx < a.flag_cap) a.flags[2 + idx] = (unsigned)nid_f;
            else atomicOr(a.flags + 2 + FLAG_CAP + (nid_f >> 5), 1u << (nid_f & 31));
          }
        }
        if (OUT == OUT_COLLAPSED) {
          if (ok) reinterpret_cast<float*>(a.out)[pos * a.Cout + co] = m;
        } else {
          // a 16x16 bit transpose per 16-lane row gives lane t the 16 channel bits of step t (den_mfma_fp6v2.hip)
          const unsigned bitsv = spk_transpose16_rows(mybits, lane);
          const int t = lane & 15, hi = (lane >> 4) & 1;
          if (ok && OUT == OUT_S32) {
            uint2 o;
            o.x = spread8_v(bitsv & 0xffu);
            o.y = spread8_v((bitsv >> 8) & 0xffu);
            uint8_t* rec = reinterpret_cast<uint8_t*>(a.out) + ((((long long)b * G + g) * Ho * Wo + oy * Wo + ox) * T16 + t) * 16;
            *reinterpret_cast<uint2*>(rec + 8 * hi) = o;
          }
          if (ok && OUT == OUT_PTC) {
            uint4 o;
            o.x = ((bitsv & 0xfu) * 0x00204081u) & 0x01010101u;
            o.y = (((bitsv >> 4) & 0xfu) * 0x00204081u) & 0x01010101u;
            o.z = (((bitsv >> 8) & 0xfu) * 0x00204081u) & 0x01010101u;
            o.w = (((bitsv >> 12) & 0xfu) * 0x00204081u) & 0x01010101u;
            *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(a.out) + (pos * T16 + t) * a.Cout + g * 32 + 16 * hi) = o;
          }
        }
        if constexpr (SPK_VT_INWAVE_FIX) {
          unsigned long long fm = __builtin_amdgcn_ballot_w64(flg && ok);
          if (fm) {                                             // (wave-uniform; 0.6 % of the tiles)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tile's own stores first: the exact values overwrite them
            while (fm) {
              const int l = __builtin_ctzll(fm);
              fm &= fm - 1;
              const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nid_f & 0xffffffffll), l);
              const unsigned hi32 = (unsigned)__builtin_amdgcn_readlane((int)(nid_f >> 32), l);
              vae_fix_neuron_call<GEO, H, W, NCH, OUT>(a.in, a.qtab, a.scale, a.bias, a.bn_a, a.bn_b, a.coef, a.out, a.Cin, a.Cout,
                                                       (long long)(((unsigned long long)hi32 << 32) | lo), lane);
            }
          }
        }
      }
      VT_ACC(3, st_t);
    };

    // the item's passes, class-major, dealt round-robin over the waves (every wave gets a mix of cheap and expensive classes); a
    // wave's passes of one class follow one another, so the class's first-tap weight fragments are read once for all of them
    [[maybe_unused]] int Pdyn = wave;                          // (SPK_VT_DYN) the wave's pass: the first NWV are dealt, the rest handed out
    tfor<NCLS>([&](auto cls_tag) {
      constexpr int CLS = decltype(cls_tag)::value;
      constexpr int NHELD = n_on_taps<GEO, CLS>() < SPK_VT_NHOLD ? n_on_taps<GEO, CLS>() : SPK_VT_NHOLD;
      int P = DYN ? Pdyn : CLS * NPASS + ((wave - CLS * NPASS) % SPK_VT_NWV + SPK_VT_NWV) % SPK_VT_NWV;   // static: first P >= CLS * NPASS with P = wave (mod NWV)
      if (P < (CLS + 1) * NPASS) {
        v6i hb[SPK_VT_NHOLD][NCH * 2];
#pragma unroll
        for (int h = 0; h < SPK_VT_NHOLD; ++h)
#pragma unroll
          for (int j = 0; j < NCH * 2; ++j) hb[h][j] = v6i{0, 0, 0, 0, 0, 0};
        if constexpr (SPK_VT_HOIST && SPK_VT_D4) {
          tfor<NHELD>([&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value, TAPH = on_tap<GEO, CLS>(h);
#pragma unroll
            for (int j = 0; j < NCH * 2; ++j) {
              const uint8_t* p = sW + (TAPH * TPL + j) * WT;
              const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
              const v2i y = *reinterpret_cast<const v2i*>(p + 1024 + lane * 8);
              hb[h][j] = v6i{x[0], x[1], x[2], x[3], y[0], y[1]};
            }
          });
        }
#if SPK_VT_STAMP
        {
          int sink = hb[0][0][0];
          asm volatile("" : "+v"(sink));
          VT_ACC(5, st_t);
        }
#endif
        if constexpr (DYN) {
          while (P < (CLS + 1) * NPASS) {
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(s_ctr, 1);          // (requested before the pass, read behind it)
            run_pass(cls_tag, P - CLS * NPASS, hb);
            P = __builtin_amdgcn_readfirstlane(nxt);
          }
          Pdyn = P;
        } else {
          for (; P < (CLS + 1) * NPASS; P += SPK_VT_NWV) run_pass(cls_tag, P - CLS * NPASS, hb);
        }
      }
    });
    if (!DB) __syncthreads();                                  // everyone is done with the slab before the next copy lands
    VT_ACC(4, st_t);
  }
  spk_dma_wait_all();
#if SPK_VT_STAMP
  if (blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 7; ++k) atomicAdd(&g_vt_stamp[wave][k], (unsigned long long)st_acc[k]);
    atomicAdd(&g_vt_stamp[wave][7], (unsigned long long)(VT_CLK() - st_begin));
  }
#endif
  // hand-over to the repair launch: the workgroup that finishes last publishes the count and re-arms the live counter (the
  // workgroups finish at different times, so this ticket costs nothing; a ticket in the repair launch, whose workgroups all
  // arrive at once, cost 16 us, a separate reset launch 5)
  // (no fence needed: every atomicAdd on the counter has returned before its wave reaches the barrier, and the count is read
  //  back with an atomic)
  __syncthreads();
  if (tid == 0) {
    if (atomicAdd(a.flags + a.ticket_idx, 1u) == gridDim.x - 1) {
      a.flags[1] = atomicAdd(a.flags, 0u);
      a.flags[0] = 0u;
      a.flags[a.ticket_idx] = 0u;
    }
  }
}

// One flagged neuron, exactly, by one wave: lane = (time step t, quarter of a 32-channel chunk); every contributing tap and
// chunk costs one 4-byte read of the spike record and eight quantised weights; 64-bit sums meet through two shuffles, then
// the reference's BN and LIF steps run on the correctly rounded pre-activations and the neuron's output is rewritten.
template <int GEO, int H, int W, int NCH, int OUT>
__device__ __forceinline__ void vae_fix_neuron(const TArgs& a, long long nid, int lane) {
  constexpr int Ho = Geo<GEO, H, W>::Ho, Wo = Geo<GEO, H, W>::Wo;
  const int co = (int)(nid % a.Cout);
  const long long pos = nid / a.Cout;
  const int ox = (int)(pos % Wo), oy = (int)((pos / Wo) % Ho), b = (int)(pos / ((long long)Wo * Ho));
  const int t = lane & 15, q = lane >> 4;
  long long part = 0;
  for (int ky = 0; ky < 3; ++ky) {
    int iy;
    if (GEO == 0) {
      const int ty = oy + 1 - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= H) continue;
      iy = ty >> 1;
    } else {
      iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
    }
    for (int kx = 0; kx < 3; ++kx) {
      int ix;
      if (GEO == 0) {
        const int tx = ox + 1 - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= W) continue;
        ix = tx >> 1;
      } else {
        ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
      }
      const int tap = ky * 3 + kx;
      for (int c = 0; c < NCH; ++c) {
        if (c * 32 + 8 * q >= a.Cin) continue;                // (padding channels of the last chunk carry no weights)
        const unsigned nib = *reinterpret_cast<const unsigned*>(a.in + ((((long long)b * NCH + c) * H * W + iy * W + ix) * T16 + t) * 16 + 4 * q);
        const int4* qp = reinterpret_cast<const int4*>(a.qtab + ((long long)co * 9 + tap) * a.Cin + c * 32 + 8 * q);
        const int4 q0 = qp[0], q1 = qp[1];
        part += (nib & 0x0000000fu) ? (long long)q0.x : 0ll;
        part += (nib & 0x000000f0u) ? (long long)q0.y : 0ll;
        part += (nib & 0x00000f00u) ? (long long)q0.z : 0ll;
        part += (nib & 0x0000f000u) ? (long long)q0.w : 0ll;
        part += (nib & 0x000f0000u) ? (long long)q1.x : 0ll;
        part += (nib & 0x00f00000u) ? (long long)q1.y : 0ll;
        part += (nib & 0x0f000000u) ? (long long)q1.z : 0ll;
        part += (nib & 0xf0000000u) ? (long long)q1.w : 0ll;
      }
    }
  }
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const int lo = __shfl_xor((int)(unsigned)(part & 0xffffffffll), off);
    const int hi = __shfl_xor((int)(part >> 32), off);
    part += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  }
  const double sc = a.scale[co], bi = a.bias[co];
  const float bna = a.bn_a[co], bnb = a.bn_b[co];
  float v = 0.f, m = 0.f;
  unsigned bits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int lo = __shfl((int)(unsigned)(part & 0xffffffffll), r);
    const int hi = __shfl((int)(part >> 32), r);
    const long long S = (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    const float y = (float)fma((double)S, sc, bi);
    const bool s = spk_lif_step_default(v, fmaf(y, bna, bnb));
    if (OUT == OUT_COLLAPSED) m = m + (s ? a.coef[r] : 0.f);
    bits |= s ? (1u << r) : 0u;
  }
  if (OUT == OUT_COLLAPSED) {
    if (lane == 0) reinterpret_cast<float*>(a.out)[nid] = m;
  } else if (OUT == OUT_S32) {
    if (lane < 16) {                                          // lane = time step: one nibble of the (position, t) record
      const int g = co >> 5, G = a.Cout >> 5, byte = (co & 31) >> 1;
      uint8_t* rec = reinterpret_cast<uint8_t*>(a.out) + ((((long long)b * G + g) * Ho * Wo + oy * Wo + ox) * T16 + lane) * 16;
      unsigned* wp = reinterpret_cast<unsigned*>(rec + (byte & ~3));
      const unsigned shw = 8u * (byte & 3) + 4u * (co & 1);
      atomicAnd(wp, ~(0xFu << shw));
      if ((bits >> lane) & 1u) atomicOr(wp, 0x2u << shw);
    }
  } else {
    if (lane < 16) reinterpret_cast<uint8_t*>(a.out)[(pos * T16 + lane) * a.Cout + co] = (uint8_t)((bits >> lane) & 1u);
  }
}

template <int GEO, int H, int W, int NCH, int OUT>
__device__ __attribute__((noinline)) void vae_fix_neuron_call(const uint8_t* in, const int* qtab, const double* scale, const double* bias,
                                                              const float* bn_a, const float* bn_b, const float* coef, void* out,
                                                              int Cin, int Cout, long long nid, int lane) {
  TArgs a;
  a.in = in; a.qtab = qtab; a.scale = scale; a.bias = bias; a.bn_a = bn_a; a.bn_b = bn_b; a.coef = coef; a.out = out;
  a.Cin = Cin; a.Cout = Cout;
  a.wq = nullptr; a.flags = nullptr; a.flag_cap = 0; a.ticket_idx = 0; a.B = 0;
  vae_fix_neuron<GEO, H, W, NCH, OUT>(a, nid, lane);
}

template <int GEO, int H, int W, int NCH, int OUT>
__global__ __launch_bounds__(256) void vae_fp6_fixup_kernel(TArgs a, long long n_words) {
  const int lane = threadIdx.x & 63;
  const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwv = (long long)gridDim.x * 4;
  const unsigned count = a.flags[1];
  const unsigned nlist = count < a.flag_cap ? count : a.flag_cap;
  for (long long e = wv; e < nlist; e += nwv) vae_fix_neuron<GEO, H, W, NCH, OUT>(a, (long long)a.flags[2 + e], lane);
  if (count > a.flag_cap) {                  // overflow: the rest sit in the bitmap; every wave scans a share, clearing as it goes
    unsigned* bm = a.flags + 2 + FLAG_CAP;
    for (long long wi = wv; wi < n_words; wi += nwv) {
      unsigned wd = bm[wi];
      if (wd && lane == 0) bm[wi] = 0u;
      while (wd) {
        const int bit = __ffs((int)wd) - 1;
        wd &= wd - 1;
        vae_fix_neuron<GEO, H, W, NCH, OUT>(a, wi * 32 + bit, lane);
      }
    }
  }
}

// one block per output channel: channel maximum -> shift s, every weight -> six balanced radix-32 digits, written as the
// per-lane 24-byte B fragments of the tiles listed at tiles_per_tap; also the quantised weights themselves (int32
// [Cout][9][Cin]) for the exact recomputation.  w: Conv2d [Cout][Cin][3][3] or ConvTranspose2d [Cin][Cout][3][3].
__global__ __launch_bounds__(256) void pack_vae_fp6_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                           uint8_t* __restrict__ wq, double* __restrict__ scale,
                                                           double* __restrict__ bias_d, int* __restrict__ qtab, int Cout,
                                                           int Cin, int transposed) {
  __shared__ float smax[256];
  const int co = blockIdx.x, n = Cin * 9;
  const int NCH = (Cin + 31) / 32, TPT = tiles_per_tap(NCH);
  auto wat = [&](int ci, int tap) -> float {
    return transposed ? w[((long long)ci * Cout + co) * 9 + tap] : w[((long long)co * Cin + ci) * 9 + tap];
  };
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(wat(i / 9, i % 9)));
  smax[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    __syncthreads();
  }
  m = smax[0];
  int e = 0;
  if (m > 0.f) frexpf(m, &e);
  const int sh = 29 - e;                      // |w| * 2^sh < 2^29 <= 16.5 * 32^5
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ci = i / 9, tap = i - 9 * ci;
    qtab[((long long)co * 9 + tap) * Cin + ci] = (int)rint(ldexp((double)wat(ci, tap), sh));
  }
  if (threadIdx.x == 0) { scale[co] = ldexp(1.0, -sh); bias_d[co] = bias ? (double)bias[co] : 0.0; }
  const int g = co >> 5, col = co & 31;
  for (int rec = threadIdx.x; rec < 9 * TPT * 2; rec += 256) {
    const int kh = rec & 1, tau = rec >> 1, tap = tau / TPT, j = tau % TPT;
    int chunk, digit;
    bool live = true;
    if (NCH == 2) { chunk = j < 4 ? (j >> 1) : kh; digit = j < 4 ? 2 * (j & 1) + kh : 4; }
    else { chunk = 0; digit = j < 2 ? 2 * j + kh : 4; live = j < 2 || kh == 0; }
    unsigned bits[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < 32 && live; ++k) {
      const int ci = chunk * 32 + k;
      long long q = ci < Cin ? (long long)rint(ldexp((double)wat(ci, tap), sh)) : 0ll;
      int dg[6];
#pragma unroll
      for (int p = 5; p >= 1; --p) {
        const int r = (int)(((q + 16) & 31) - 16);
        dg[p] = r;
        q = (q - r) >> 5;
      }
      dg[0] = (int)q;
      int d = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p) d = (p == digit) ? dg[p] : d;
      const unsigned code = (d < 0 ? 0x20u : 0u) | (unsigned)(d < 0 ? -d : d);
      const int bit = 6 * k, wd = bit >> 5, sft = bit & 31;
#pragma unroll
      for (int q2 = 0; q2 < 6; ++q2) {
        if (q2 == wd) bits[q2] |= code << sft;
        if (q2 == wd + 1 && sft > 26) bits[q2] |= code >> (32 - sft);
      }
    }
    uint8_t* tile = wq + ((long long)g * 9 * TPT + tau) * WT;
    const int ln = kh * 32 + col;
    unsigned* d16 = reinterpret_cast<unsigned*>(tile + ln * 16);
    unsigned* d8 = reinterpret_cast<unsigned*>(tile + 1024 + ln * 8);
    d16[0] = bits[0]; d16[1] = bits[1]; d16[2] = bits[2]; d16[3] = bits[3];
    d8[0] = bits[4]; d8[1] = bits[5];
  }
}

// u8 PTC [B][HW][16][C] -> S32 [B][ceil(C/32)][HW][16][16 B] (channels beyond C: zero nibbles); one thread per 16-byte record
__global__ void ptc_to_s32_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int B, int HW, int C) {
  const int nch = (C + 31) / 32;
  const long long total = (long long)B * nch * HW * T16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int t = (int)(r % T16); r /= T16;
    const int p = (int)(r % HW); r /= HW;
    const int cc = (int)(r % nch);
    const int b = (int)(r / nch);
    const uint8_t* src = in + (((long long)b * HW + p) * T16 + t) * C + cc * 32;
    const int nc = C - cc * 32 < 32 ? C - cc * 32 : 32;
    unsigned w[4] = {0, 0, 0, 0};
    if ((C & 15) == 0) {                                    // 16 or 32 channels: vector loads
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (16 * h < nc) {
          const uint4 v = *reinterpret_cast<const uint4*>(src + 16 * h);
          const unsigned q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {                     // four bytes (0 / 1) -> four nibbles (0 / 2)
            const unsigned x = q[k] & 0x01010101u;
            const unsigned n4 = ((x | (x >> 4)) & 0x00ff00ffu);
            const unsigned n16 = (n4 | (n4 >> 8)) & 0xffffu;
            w[2 * h + (k >> 1)] |= (n16 << 1) << (16 * (k & 1));
          }
        }
      }
    } else {
      for (int c = 0; c < nc; ++c) w[c >> 3] |= (src[c] ? 2u : 0u) << (4 * (c & 7));
    }
    *reinterpret_cast<uint4*>(out + i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

template <int GEO, int H, int W, int NCH, int OUT, int SPLIT, bool DB>
int launch_vae(const TArgs& a, long long n_words, hipStream_t stream) {
  constexpr int TPT = tiles_per_tap(NCH), TPL = SPK_VT_D4 ? TPT - 1 : TPT;
  constexpr int SROWS = GEO == 0 ? H / SPLIT + 1 : H + 1;
  constexpr int NPOS = GEO == 0 ? (H / SPLIT) * W : (H / 2) * (W / 2), NCLS = GEO == 0 ? 4 : 1;
  // (+ the four-digit form's counters: u8 [cells][16] and int [classes][positions])
  const size_t lds = (size_t)(DB ? 2 : 1) * NCH * SROWS * (W + 1) * POSB + 9 * TPL * WT +
                     (SPK_VT_D4 ? (size_t)((SROWS * (W + 1) * 16 + 15) & ~15) + (size_t)NCLS * NPOS * 4 : 0) + 16;
  if (lds > 160 * 1024) return SPK_ERR_UNSUPPORTED;
  const int cus = spk_cu_count(), G = a.Cout / 32;
  const int grid = cus >= G ? (cus / G) * G : G;
  hipLaunchKernelGGL((vae_fp6_kernel<GEO, H, W, NCH, OUT, SPLIT, DB>), dim3(grid), dim3(SPK_VT_NWV * 64), lds, stream, a);
  SPK_LAUNCH_CHECK();
  if constexpr (!SPK_VT_INWAVE_FIX) {
    hipLaunchKernelGGL((vae_fp6_fixup_kernel<GEO, H, W, NCH, OUT>), dim3(4 * cus), dim3(256), 0, stream, a, n_words);
    SPK_LAUNCH_CHECK();
  }
  return SPK_OK;
}

}  // namespace

extern "C" long long spk_vae_fp6_packed_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || Cin > 64 || (Cin % 8) || (Cout % 32)) return -1;
  return (long long)(Cout / 32) * 9 * tiles_per_tap((Cin + 31) / 32) * WT;
}

extern "C" int spk_vae_fp6_pack(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d, int* qtab,
                                int Cout, int Cin, int transposed, hipStream_t stream) {
  if (!w || !wq || !scale || !bias_d || !qtab || Cout <= 0 || Cin <= 0) return SPK_ERR_ARG;
  if (Cin > 64 || (Cin % 8) || (Cout % 32)) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_vae_fp6_kernel, dim3(Cout), dim3(256), 0, stream, w, bias, wq, scale, bias_d, qtab, Cout, Cin,
                     transposed);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" long long spk_vae_fp6_flag_words(int B, int Cout, int Ho, int Wo) {
  if (B <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0) return -1;
  return 2 + (long long)FLAG_CAP + ((long long)B * Cout * Ho * Wo + 31) / 32 + 1;
}

extern "C" int spk_ptc_to_s32(const uint8_t* in_ptc, uint8_t* out_s32, int T, int B, int HW, int C, hipStream_t stream) {
  if (!in_ptc || !out_s32 || B <= 0 || HW <= 0 || C <= 0) return SPK_ERR_ARG;
  if (T != T16) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)B * ((C + 31) / 32) * HW * T16;
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(ptc_to_s32_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, stream, in_ptc, out_s32,
                     B, HW, C);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_vae_fp6_fwd(const uint8_t* in_s32, const uint8_t* wq, const double* scale, const double* bias_d,
                               const int* qtab, const float* bn_a, const float* bn_b, const float* coef_or_null, void* out,
                               int out_kind, unsigned* flag_words, int T, int B, int H, int W, int Cin, int Cout, int transposed,
                               int flag_cap, hipStream_t stream) {
  if (!in_s32 || !wq || !scale || !bias_d || !qtab || !bn_a || !bn_b || !out || !flag_words || B <= 0) return SPK_ERR_ARG;
  if (out_kind == OUT_COLLAPSED && !coef_or_null) return SPK_ERR_ARG;
  if (T != T16 || (Cout % 32) || B > (1 << 22)) return SPK_ERR_UNSUPPORTED;
  TArgs a;
  a.in = in_s32; a.wq = wq; a.scale = scale; a.bias = bias_d; a.bn_a = bn_a; a.bn_b = bn_b; a.coef = coef_or_null; a.out = out;
  a.flags = flag_words; a.qtab = qtab; a.B = B; a.Cout = Cout; a.Cin = Cin;
  a.flag_cap = flag_cap < 0 || (unsigned)flag_cap > FLAG_CAP ? FLAG_CAP : (unsigned)flag_cap;   // id-list entries used (< 0: all); layout fixed
  const int Ho = transposed ? 2 * H : H / 2, Wo = transposed ? 2 * W : W / 2;
  const long long neurons = (long long)B * Cout * Ho * Wo;
  if (neurons >= (1ll << 32)) return SPK_ERR_UNSUPPORTED;                                   // neuron ids are 32-bit
  const long long n_words = (neurons + 31) / 32;
  a.ticket_idx = 2 + (long long)FLAG_CAP + n_words;
  if (transposed && Cin == 64 && out_kind == OUT_COLLAPSED) {                                // decoder convT2
#ifndef SPK_VT_T2_SPLIT
#define SPK_VT_T2_SPLIT 2       // items per image of the 14x14 -> 28x28 layer: 2 = half images, one input slab (two do not fit beside the weights);
                                // 7 = two class rows per item, two slabs (the copy of the next item runs under this one's passes): 231 against
                                // 195 us -- 28 items per workgroup pay the counting phases' barriers 28 times (15 % of the launch) and the waves
                                // then wait for one another at the item's first barrier instead (profiles/r5_ab_kernel_variants.txt (7))
#endif
    if (H == 14 && W == 14) return launch_vae<0, 14, 14, 2, OUT_COLLAPSED, SPK_VT_T2_SPLIT, (SPK_VT_T2_SPLIT > 2)>(a, n_words, stream);
    if (H == 16 && W == 16) return launch_vae<0, 16, 16, 2, OUT_COLLAPSED, 2, false>(a, n_words, stream);
  }
  if (transposed && Cin == 16 && out_kind == OUT_S32) {                                      // decoder convT1
    if (H == 7 && W == 7) return launch_vae<0, 7, 7, 1, OUT_S32, 1, true>(a, n_words, stream);
    if (H == 8 && W == 8) return launch_vae<0, 8, 8, 1, OUT_S32, 1, true>(a, n_words, stream);
  }
  if (!transposed && Cin == 32 && out_kind == OUT_PTC) {                                     // encoder conv2
    if (H == 14 && W == 14) return launch_vae<1, 14, 14, 1, OUT_PTC, 1, true>(a, n_words, stream);
    if (H == 16 && W == 16) return launch_vae<1, 16, 16, 1, OUT_PTC, 1, false>(a, n_words, stream);   // (two slabs do not fit)
  }
  return SPK_ERR_UNSUPPORTED;
}

#if SPK_VT_STAMP
// (timing builds only) the accumulated phase cycles of workgroup 0, [wave][8]; cleared on read
extern "C" int spk_vt_stamps(unsigned long long* out, int* n_waves) {
  unsigned long long z[SPK_VT_NWV][8] = {};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vt_stamp), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_vt_stamp), z, sizeof(z)) != hipSuccess) return -1;
  *n_waves = SPK_VT_NWV;
  return 0;
}
#endif
