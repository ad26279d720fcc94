// 3x3 convolution over binary spike tensors on the MI355X matrix cores with the CDNA4 block-scaled MFMA
// (v_mfma_scale_f32_32x32x64_f8f6f4): spikes as fp4 (e2m1), weights as six radix-32 fp6 (e2m3) digit planes.
// BN + LIF scan fused into the accumulator epilogue.  Denoiser conv2..conv5, R/snn_model/vq_diffusion.py:166-184,
// 201-204.  Same mapping as den_mfma.hip (read that header first); what differs:
//
// EXACT FORMULATION.  Each fp32 weight is re-encoded once (spk_den_pack_weight_fp6) as a 29-bit fixed-point number
// relative to its output channel's largest magnitude and split into six balanced radix-32 digits D0..D5 in [-16,16]:
// w_q = 2^-s * sum_p D_p * 32^(5-p).  A digit d is the e2m3 value d/8 (e2m3 holds every multiple of 1/8 up to 2.0
// exactly: code = sign<<5 | |d|), the B-side block scale 2^3 makes the products integers again.  Spikes are the e2m1
// codes 0x0 / 0x2 (0.0 / 1.0).  Products are exact, and the fp32 accumulators hold integers below 4608*16 < 2^24, so
// the accumulation is exact (probe: tools/f8f6f4_probe.hip).  The six digit sums are recombined exactly in fp64 and
// rounded ONCE to fp32 -- the same contract as the int8 kernel.  Weights within 2^6 of the channel maximum are
// represented exactly, smaller ones are rounded at 2^-29 of that maximum.
//
// WHY.  The fp6/fp4 MFMA retires K = 64 in the 33 cycles the int8 MFMA needs for K = 32 (measured), i.e. 5 digit bits
// x 2 per unit time against 8: six fp6 planes cost 3/4 of the matrix-core time of four int8 planes.
//
// MAPPING.  A work item = one image x 16 output channels = 7 row tiles x 3 column tiles (16 channels x 2 planes
// each) per wave = 336 accumulator registers: 15 tiles live in AGPRs, 6 in VGPRs.  hipcc cannot split an MFMA
// accumulator set over both files, so the MFMAs are inline assembly with explicit register classes; the wait states
// between the last MFMA and the first read of an accumulator are inserted by hand.
// K chunk = 64 input channels: spikes travel nibble-packed ("C4": [B][C/64][HW][T][32 B], channel c of a chunk in
// byte c/2, low nibble first), so one position is still 512 B and the LDS image / DMA geometry equals the int8
// kernel's.  The zero-bordered LDS image uses a pitch of W+1 cells (the zero column is shared by x = -1 of a row and
// x = W of the previous one) so that two images and two 40.5 KB weight slabs fit the 160 KB LDS.
#include "den_common.h"
#include "../../include/spkdiff.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int T16 = 16;
constexpr int CK = 64;                              // channels per K chunk
constexpr int POS_BYTES = T16 * CK / 2;             // 512 B per latent position per chunk
constexpr int W_TILE_BYTES = 64 * 24;               // one (tap, column tile): 64 lanes x 32 six-bit codes
constexpr int W_PIECES = (9 * 3 * W_TILE_BYTES + 1023) / 1024;   // 41 one-KiB DMA pieces
constexpr int W_CHUNK_BYTES = W_PIECES * 1024;      // 41984 B per (channel group, chunk): 27 tiles + 512 B of padding
constexpr int NPW = (W_PIECES + 3) / 4;             // W pieces per wave (wave w copies pieces w, w + 4, ...)

struct Fp6Args {
  const uint8_t* in0; int nch0;
  const uint8_t* wq; const double* scale; const double* bias; const float* bn_a; const float* bn_b;
  uint8_t* out; float* v_io; uint8_t* out_cnt;
  const int* n_dyn;   // optional device-side batch count (<= B): only images [0, *n_dyn) are processed
  float* pre;  // RAW form only: pre-activations (conv + bias) fp32, channels-last [T][B][HW][Cout]
  int B, H, W, Cout;
  int gx;      // > 0: XCD-aware item walk with gx channel groups per XCD (see the kernel); 0: image-major
};

// D = A(32 x 64 fp4) * B(64 x 32 fp6) + C, accumulator in AGPRs ("a") or VGPRs ("v"); *_Z: C = 0 (first MFMA of an item).
// s_nop 1: the two wait states between a VALU write of an operand register (the B fragment hand-over is v_mov) and
// the MFMA that reads it, which hipcc's hazard recognizer cannot insert around inline assembly.
#ifndef SPK_FP6_PRE
#define SPK_FP6_PRE "s_nop 1\n\t"
#endif
#define SPK_MFMA_FP6(CLS, acc, av, bv, sa, sb)                                                                       \
  asm volatile(SPK_FP6_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2"  \
               : "+" CLS(acc) : "v"(av), "v"(bv), "v"(sa), "v"(sb))
#define SPK_MFMA_FP6_Z(CLS, acc, av, bv, sa, sb)                                                                     \
  asm volatile(SPK_FP6_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2"   \
               : "=&" CLS(acc) : "v"(av), "v"(bv), "v"(sa), "v"(sb))

#ifndef SPK_FP6_DBG
#define SPK_FP6_DBG 0           // timing experiments only: 1 = no steady-state DMA, 4 = no epilogue (results are wrong)
#endif
#ifndef SPK_FP6_PF
#define SPK_FP6_PF 4
#endif
#ifndef SPK_FP6_DMA_EVERY
#define SPK_FP6_DMA_EVERY 2     // one piece every second step: all 18 are issued in the first 60 % of a chunk and land before its end
#endif
// NT = row tiles per wave: 6 when H*W is odd and H*W / 2 <= 24 (7x7: 24 tiles of 2 positions, no padding rows; the last
// position of every image is left to conv3x3_fp6_lastpos_kernel), else 7 (up to 56 positions, padded to 28 tiles)
constexpr int NPA = 7;         // A-slab DMA pieces per wave
constexpr int N_AGPR = 16;     // accumulators (row tile i, column tile j: index 3*i + j) that live in AGPRs (256 registers)

// RAW = true: the training forward -- the same exact pre-activations written as fp32 instead of
// the BN + LIF scan (batch-statistics BN needs them all before any neuron can be stepped).
// SPLIT = true (latents too large for one item, 8x8): an item is one of the two row BANDS of an image (H / 2 output rows,
// H / 2 + 1 input rows: one halo row from the other band).  Both bands sit in LDS rows 1..H/2+1 of a (H/2 + 3)-row padded
// image whose rows 0 and H/2 + 2 stay zero; the top band's outputs are centred on LDS rows 1.., the bottom band's on
// rows 2.. -- one row offset added to the fragment addresses per item.
// NWV = waves per workgroup: 4 (one per SIMD, NT = 6 / 7 / 4 tiles each) or 8 (two per SIMD with NT = 3 tiles each: the same
// item, LDS plan and DMA volume; meant to let the second wave of a SIMD fill the matrix pipe while the first reads
// fragments, issues copies or runs its epilogue).  Measured (SPKDIFF_FP6_WAVES=8, correct, same results): 7 % SLOWER on
// conv4 / conv5 (620 vs 580 us, 531 vs 501 us), with or without de-phasing the two waves -- the kernel is not starved
// for issue slots but limited by the power the matrix pipe may draw; kept as an experiment, not the default.
template <int NT, bool RAW = false, bool SPLIT = false, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, 1) void conv3x3_fp6_kernel(Fp6Args a) {
  constexpr int NPA_ = NWV == 8 ? 4 : NPA;                    // A pieces per wave
  constexpr int NAG = NWV == 8 ? 8 : N_AGPR;                  // accumulators kept in AGPRs (two waves per SIMD: 256 registers each)
  constexpr int NPW_ = (W_PIECES + NWV - 1) / NWV;            // W pieces per wave (wave w copies pieces w, w + NWV, ...)
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int HW = a.H * a.W, PW = a.W + 1;
  const int Hb = SPLIT ? a.H / 2 : a.H;               // output rows of an item
  const int Hin = SPLIT ? Hb + 1 : a.H;               // input rows staged per item
  const int HWb = Hb * a.W;                           // output positions of an item
  const int npp = (Hin + 2) * PW + 1;
  const int A_BYTES = npp * POS_BYTES;
  // LDS: [A buf0][A buf1][W buf0][W buf1]
  uint8_t* const sA = lds;
  uint8_t* const sW = lds + 2 * A_BYTES;
  const unsigned sA_addr = spk_lds_addr(sA), sW_addr = sA_addr + 2 * A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = a.nch0;
  const int G = a.Cout >> 4;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  const int total = Bn * G * (SPLIT ? 2 : 1);

  // zero both A images once: the borders stay zero for the whole kernel, interiors are overwritten by DMA
  for (int i = tid; i < 2 * A_BYTES / 16; i += NWV * 64) reinterpret_cast<uint4*>(sA)[i] = make_uint4(0, 0, 0, 0);

  // per-lane LDS byte offsets of this wave's A fragments (tile ti = wave + 4*i); absent tiles read garbage that the
  // epilogue discards
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  int a_off[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int p = 2 * (wave + NWV * i) + hsel;
    const int pp = p < HWb ? (p / a.W + 1) * PW + (p % a.W) + 1 : PW + 1;
    a_off[i] = pp * POS_BYTES + tt * 32 + 16 * (half ^ (tt >> 3));   // 16-B halves swapped for t >= 8: bank-conflict-free
  }

  // DMA piece table, wave-uniform (scalar) values, one packed word per A piece to keep the SGPR budget (a spilled SGPR
  // costs a v_readlane in the K loop): bits 0..14 source byte offset in the slab, 15..30 LDS byte offset in the image,
  // bit 31 = full piece (two positions; the last piece of an odd-width row covers one position = lanes 0..31).
  // Pieces beyond the slab repeat the last one (a harmless duplicate copy) so that the K loop issues unconditionally.
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int pprow = (a.W + 1) >> 1;
  const int nA = Hin * pprow;
  unsigned pa_pk[NPA_];
#pragma unroll
  for (int j = 0; j < NPA_; ++j) {
    int id = wave_s * NPA_ + j;
    id = id < nA ? id : nA - 1;
    const int y = id / pprow, px = id - y * pprow;
    const unsigned src = (unsigned)((y * a.W + 2 * px) * POS_BYTES), dst = (unsigned)(((y + 1) * PW + 1 + 2 * px) * POS_BYTES);
    pa_pk[j] = src | (dst << 15) | ((2 * px + 1 < a.W) ? 0x80000000u : 0u);
  }
  const unsigned lane_a = (unsigned)(lane ^ ((lane >> 4) & 1)) * 16u;     // swizzled source lane (see a_off)
  const unsigned lane_w = (unsigned)lane * 16u;
  const unsigned wave_k = (unsigned)wave_s * 1024u;

  // piece q of this wave: q < NPA -> A piece q, else W piece wave + 4 * (q - NPA) (the 42nd..44th repeat an earlier one)
  auto issue_piece = [&](int q, const uint8_t* aslab, const uint8_t* wslab, unsigned dA, unsigned dW) {
    if (q < NPA_) {
      const unsigned pk = pa_pk[q];
      const unsigned long long mask = (pk >> 31) ? ~0ull : 0xffffffffull;
      spk_dma16s_masked(aslab + (pk & 0x7fffu), lane_a, dA + ((pk >> 15) & 0xffffu), mask);
    } else {
      unsigned ko = wave_k + (unsigned)NWV * 1024u * (unsigned)(q - NPA_);
      if (NWV * (q - NPA_) + NWV - 1 >= W_PIECES) ko = ko < (unsigned)W_PIECES * 1024u ? ko : ko - (unsigned)NWV * 1024u;
      spk_dma16s(wslab + ko, lane_w, dW + ko);
    }
  };
  // Item -> (image b, channel group g).  Workgroups are dealt round-robin over the 8 XCDs (k and k + 8 share one, each
  // XCD has its own 4 MB L2).  An XCD-aware walk (a.gx > 0, chosen by the host so that the packed weights of gx channel
  // groups stay L2-resident): XCD x owns group set x % nsets (gx consecutive groups) and image partition x / nsets,
  // and its gridDim.x / 8 workgroups take those gx groups of S / gx images at a time.  Against the image-major walk
  // this (1) fetches an image's spike slabs into 8 / npart L2s instead of all eight and (2) brings the four 8-byte
  // partial records that make up one 32-byte output record together in ONE L2, which merges them (measured on the
  // conv4 shape: HBM-side writes 218 -> 56 MB per launch).  a.gx == 0: image-major.
  const int S = gridDim.x >> 3;                       // workgroups per XCD
  const int gx = a.gx;
  const int nsets = gx > 0 ? G / gx : 1, npart = 8 / nsets, ipx = gx > 0 ? S / gx : 1;
  auto decode = [&](int item, int& b, int& g) {
    if (gx > 0) {
      const int j = item / (int)gridDim.x, k = item - j * (int)gridDim.x;
      const int x = k & 7, slot = k >> 3;
      const int set = x % nsets, xi = x / nsets;
      g = set * gx + slot % gx;
      b = (j * npart + xi) * ipx + slot / gx;
    } else {
      b = item / G;            // SPLIT: b = 2 * image + band
      g = item - b * G;
    }
  };
  auto slabs = [&](int item, int c, const uint8_t*& aslab, const uint8_t*& wslab) {
    int b, g;
    decode(item, b, g);
    if constexpr (SPLIT) aslab = a.in0 + ((long long)(b >> 1) * nchunks + c) * HW * POS_BYTES + (b & 1) * (Hb - 1) * a.W * POS_BYTES;
    else aslab = a.in0 + ((long long)b * nchunks + c) * HW * POS_BYTES;
    wslab = a.wq + ((long long)g * nchunks + c) * W_CHUNK_BYTES;
  };

  const int col = lane & 31, ch = col & 15, odd = col >> 4;
  const int sc_a = 0x7f7f7f7f;             // e8m0 block scales: spikes x 1
  const int sc_b = (int)0x82828282u;       //                    digits x 8 (e2m3 value d/8 -> d)

  long long dbg_t[4] = {0, 0, 0, 0};               // SPK_FP6_DBG & 64: cycles in DMA wait / barrier / K loop, chunk count
  int it = 0;                                      // running chunk counter: LDS buffer = it & 1
  if ((int)blockIdx.x < total) {
    const uint8_t *as0, *ws0;
    slabs(blockIdx.x, 0, as0, ws0);
#pragma unroll
    for (int q = 0; q < NPA_ + NPW_; ++q) issue_piece(q, as0, ws0, sA_addr, sW_addr);
  }
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    v16f acc[NT][3];      // written (not accumulated) by tap 0 of the first chunk: no explicit zeroing
    // epilogue constants of this item's channel: loaded now, their latency hides under the K loop
    int b, g;
    decode(item, b, g);
    const int band = SPLIT ? (b & 1) : 0;
    if constexpr (SPLIT) b >>= 1;
    const int band_off = band * PW * POS_BYTES;          // bottom band: fragment addresses one LDS row further down
    const int co = g * 16 + ch;
    const double sc = a.scale[co], bi = a.bias[co];
    const float bn_a = RAW ? 1.0f : a.bn_a[co], bn_b = RAW ? 0.0f : a.bn_b[co];
    for (int c = 0; c < nchunks; ++c, ++it) {
      const int buf = it & 1;
      long long tq0 = 0, tq1 = 0, tq2 = 0;
      if (SPK_FP6_DBG & 64) tq0 = __builtin_amdgcn_s_memtime();
      spk_dma_wait_all();  // this wave's share of the chunk's DMA has landed ...
      if (SPK_FP6_DBG & 64) tq1 = __builtin_amdgcn_s_memtime();
      __syncthreads();     // ... and so has everyone else's; everyone is done with the other buffer
      if (SPK_FP6_DBG & 64) tq2 = __builtin_amdgcn_s_memtime();
      // next chunk (possibly of the next item): its DMA pieces are issued between the MFMA groups below
      int nitem = item, nc = c + 1;
      if (nc == nchunks) { nc = 0; nitem = item + gridDim.x; }
      const bool have_next = nitem < total;      // otherwise the last chunk is copied once more (never read)
      const uint8_t *n_aslab, *n_wslab;
      slabs(have_next ? nitem : item, nc, n_aslab, n_wslab);
      if (SPK_FP6_DBG & 128) { n_aslab = a.in0; n_wslab = a.wq; }      // timing experiment: L2-hot, tiny DMA footprint
      const unsigned n_dA = sA_addr + (buf ^ 1) * A_BYTES;            // LDS byte addresses of the DMA destinations
      const unsigned n_dW = sW_addr + (buf ^ 1) * W_CHUNK_BYTES;

      // ---------------- 9 taps x NT row tiles x 3 column tiles, A fragments read four steps ahead ------------------
      auto compute = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint8_t* A = sA + buf * A_BYTES + band_off;
        const uint8_t* Wb = sW + buf * W_CHUNK_BYTES;
        auto lda = [&](int s) -> v4i {
          const int tap = s / NT, i = s % NT;
          const int toff = ((tap / 3 - 1) * PW + (tap % 3 - 1)) * POS_BYTES;
          return *reinterpret_cast<const v4i*>(A + a_off[i] + toff);
        };
        auto ldb = [&](int tap, int j) -> v6i {
          // 16 + 8 bytes per lane.  The 8-byte read is volatile so that hipcc does not pair the tails of two tiles in
          // one ds_read2st64_b64 -- which lands them in the wrong registers and costs a wait + v_mov per tap.
          const uint8_t* p = Wb + (tap * 3 + j) * W_TILE_BYTES;
          const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
          typedef const volatile __attribute__((address_space(3))) v2i* lds_v2i_ptr;
          const v2i y = *(lds_v2i_ptr)SPK_LDS(p + 1024 + lane * 8);
          const v6i r = {x[0], x[1], x[2], x[3], y[0], y[1]};
          return r;
        };
        v6i bc[3], bn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { bc[j] = ldb(0, j); bn[j] = bc[j]; }
        // A fragments in flight ahead of the MFMA that consumes them; with two waves per SIMD the other wave covers the
        // latency instead of registers (no second set of weight fragments either)
        constexpr int PF = NWV == 8 ? 2 : SPK_FP6_PF;
        constexpr bool BPF = NWV != 8;
        v4i af[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) af[s] = lda(s);
#pragma unroll
        for (int s = 0; s < 9 * NT; ++s) {
          const int tap = s / NT, i = s % NT;
          const v4i av = af[s % PF];
          if (s + PF < 9 * NT) af[s % PF] = lda(s + PF);
          auto mfma = [&](int j) {
#ifdef SPK_FP6_KEEP
            if (i >= SPK_FP6_KEEP) return;        // bound experiment: only the first KEEP row tiles of a wave are computed
#endif
            if (FIRST && tap == 0) {
              if (3 * i + j < NAG) SPK_MFMA_FP6_Z("a", acc[i][j], av, bc[j], sc_a, sc_b);
              else SPK_MFMA_FP6_Z("v", acc[i][j], av, bc[j], sc_a, sc_b);
            } else {
              if (3 * i + j < NAG) SPK_MFMA_FP6("a", acc[i][j], av, bc[j], sc_a, sc_b);
              else SPK_MFMA_FP6("v", acc[i][j], av, bc[j], sc_a, sc_b);
            }
          };
          // One MFMA keeps the matrix pipe busy for ~33 cycles, i.e. ~6 issue slots: the other work of a step is dealt
          // out over its three MFMA shadows (fragment prefetch | DMA piece | weight prefetch) instead of piling up in one.
          mfma(0);
          __builtin_amdgcn_sched_barrier(0);
          {
            // DMA schedule: NPA + NPW pieces spread over the 9*NT steps (every DMA_EVERY-th step issues one piece)
            constexpr int NPIECES = NPA_ + NPW_;
            constexpr int DMA_EVERY = SPK_FP6_DMA_EVERY;
            if (s % DMA_EVERY == 0 && s / DMA_EVERY < NPIECES) {
              const int q = s / DMA_EVERY;
              const bool skip = ((SPK_FP6_DBG & 8) && q >= NPA_) || ((SPK_FP6_DBG & 16) && q < NPA_);
              if (!(SPK_FP6_DBG & 1) && !skip) issue_piece(q, n_aslab, n_wslab, n_dA, n_dW);
            }
          }
          mfma(1);
          __builtin_amdgcn_sched_barrier(0);
          if (BPF && i == 0 && tap + 1 < 9) {
#pragma unroll
            for (int j = 0; j < 3; ++j) bn[j] = ldb(tap + 1, j);
          }
          mfma(2);
          if (i == NT - 1) {
            if constexpr (BPF) {
#pragma unroll
              for (int j = 0; j < 3; ++j) bc[j] = bn[j];
            } else if (tap + 1 < 9) {
#pragma unroll
              for (int j = 0; j < 3; ++j) bc[j] = ldb(tap + 1, j);
            }
          }
          __builtin_amdgcn_sched_barrier(0);     // keep the read-ahead distance (see den_mfma.hip)
        }
      };
      if (c == 0) compute(std::true_type{}); else compute(std::false_type{});
      if (SPK_FP6_DBG & 64) {
        const long long tq3 = __builtin_amdgcn_s_memtime();
        dbg_t[0] += tq1 - tq0; dbg_t[1] += tq2 - tq1; dbg_t[2] += tq3 - tq2; dbg_t[3] += 1;
      }
    }   // chunks

    // The MFMAs are opaque to hipcc's hazard recognizer: an accumulator may be read 18 wait states after the (16-pass)
    // MFMA that wrote it was issued.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (SPK_FP6_DBG & 4) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (3 * i + j < NAG) asm volatile("" : "+a"(acc[i][j]));
          sacc += acc[i][j][0];
        }
      if (sacc == 12345.f) a.out[0] = 1;
      continue;
    }

    // ---------------- epilogue: exact recombination, BN, LIF scan over the 16 accumulator registers ----------------
    // Partner lanes (col, col ^ 16) hold digit planes {0,2,4} and {1,3,5} of the same channel, for every row tile.
    // Tiles are finished in PAIRS (ia, ib):  v_permlane16_swap(acc[ia][j][r], acc[ib][j][r])  (A' = {A.row0, B.row0},
    // B' = {A.row1, B.row1} per 16-lane row pair) leaves the even lane with both digits of column tile j of tile ia and
    // the odd lane with both digits of tile ib -- 48 swaps hand every lane all six digits of ONE neuron for all 16 time
    // steps, and both lane parities then do useful, different work: recombine + BN + LIF scan of their own tile.
    // A digit pair 32*D_even + D_odd is exact in fp32, the three pairs are combined exactly in fp64, one rounding.
    // (The odd tile out is split over the lane parities by time steps instead, as in den_mfma.hip.)
    // The empty asm pins an AGPR-resident accumulator to its AGPRs up to that point: without it hipcc copies all the
    // 16-register tuples to VGPRs at the top of the epilogue (and spills).  The pair holding the VGPR-resident
    // accumulators goes first (frees their registers for the scan temporaries).
#pragma unroll
    for (int k = 0; k < (NT + 1) / 2; ++k) {
      // 7 tiles: (5,6) (0,1) (2,3) (4,-);  6 tiles: (4,5) (0,1) (2,3)
      const int ia = k == 0 ? NT - 2 : (((NT & 1) && k == (NT + 1) / 2 - 1) ? NT - 3 : 2 * (k - 1));
      const bool paired = !((NT & 1) && k == (NT + 1) / 2 - 1);
      const int ib = paired ? ia + 1 : ia;
#ifdef SPK_FP6_KEEP
      if (ia >= SPK_FP6_KEEP) continue;
#endif
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (3 * ia + j < NAG) asm volatile("" : "+a"(acc[ia][j]));
        if (paired && 3 * ib + j < NAG) asm volatile("" : "+a"(acc[ib][j]));
      }
      float x[16];
      auto recombine3 = [&](const float (&pr)[3]) -> float {
        const double s1 = fma((double)pr[0], 1024.0, (double)pr[1]);               // exact
        const double s = fma(s1, 1024.0, (double)pr[2]);                           // exact: |s| < 2^43
        return (float)fma(s, sc, bi);                                              // the one rounding to fp32
      };
      if (paired) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pr[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const v2u p = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ia][j][r]), __float_as_uint(acc[ib][j][r]),
                                                           false, false);
            pr[j] = fmaf(__uint_as_float(p[0]), 32.0f, __uint_as_float(p[1]));     // exact: |.| < 2^22
          }
          x[r] = recombine3(pr);
        }
      } else {
        // the odd tile out: swapping acc[r] with acc[r + 8] of the SAME tile gives the even lane both digits of step r and
        // the odd lane those of step r + 8; each recombines 8 steps and one more swap hands both all 16 values
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float pr[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const v2u p = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ia][j][r]), __float_as_uint(acc[ia][j][r + 8]),
                                                           false, false);
            pr[j] = fmaf(__uint_as_float(p[0]), 32.0f, __uint_as_float(p[1]));
          }
          const float xm = recombine3(pr);
          const v2u xx = __builtin_amdgcn_permlane16_swap(__float_as_uint(xm), __float_as_uint(xm), false, false);
          x[r] = __uint_as_float(xx[0]);                                 // t = r     (computed by the even lane)
          x[r + 8] = __uint_as_float(xx[1]);                             // t = r + 8 (computed by the odd lane)
        }
      }
      const int ti = wave + NWV * (odd ? ib : ia);
      const int pl = 2 * ti + half;                 // accumulator lane-half == position within the tile
      const int p = pl + band * HWb;                // position in the image
      const bool pos_ok = pl < HWb && (paired || !odd);
      if constexpr (RAW) {
        // x[r] = pre-activation of neuron (b, p, co) at t = r; 16 consecutive channels (64 B) per 16-lane row and step
        if (pos_ok) {
          float* dst = a.pre + ((long long)b * HW + p) * a.Cout + co;
          const long long tstride = (long long)a.B * HW * a.Cout;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[r * tstride] = x[r];
        }
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      const long long vidx = ((long long)b * a.Cout + co) * HW + (pos_ok ? p : 0);
      float v = a.v_io ? a.v_io[vidx] : 0.f;
      unsigned mybits = 0;                    // bit r = this lane's neuron fired at t = r
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool s = spk_lif_step_default(v, fmaf(x[r], bn_a, bn_b)) && pos_ok;
        mybits |= s ? (1u << r) : 0u;
      }
      const unsigned cnt = __popc(mybits);
      const unsigned bitsv = spk_transpose16_rows(mybits, lane);     // lane t of a 16-lane row: channel bits of step t
      if (a.v_io && pos_ok) a.v_io[vidx] = v;
      if (a.out_cnt && pos_ok)
        a.out_cnt[(((long long)b * (a.Cout >> 5) + (co >> 5)) * HW + p) * 32 + (co & 31)] = (uint8_t)cnt;
      // every lane stores one (position, time step): 16 channels = 16 e2m1 nibbles = 8 bytes
      if (pos_ok) {
        auto spread8 = [](unsigned x) -> unsigned {          // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
          x = (x | (x << 12)) & 0x000f000fu;
          x = (x | (x << 6)) & 0x03030303u;
          x = (x | (x << 3)) & 0x11111111u;
          return x << 1;
        };
        uint2 o;
        o.x = spread8(bitsv & 0xffu);
        o.y = spread8((bitsv >> 8) & 0xffu);
        const int co0 = g * 16;
        uint8_t* dst = a.out + ((((long long)b * (a.Cout >> 6) + (co0 >> 6)) * HW + p) * T16 + (lane & 15)) * 32 +
                       ((co0 & 63) >> 1);
        *reinterpret_cast<uint2*>(dst) = o;
      }
      __builtin_amdgcn_sched_barrier(0);        // keep the tile pairs from being interleaved (VGPR pressure)
    }
  }   // items
  spk_dma_wait_all();     // the copy issued during the very last chunk must not outlive the workgroup's LDS allocation
  if ((SPK_FP6_DBG & 64) && lane == 0 && blockIdx.x < 4) {
    long long* o = reinterpret_cast<long long*>(a.out) + (blockIdx.x * 4 + wave) * 4;
    o[0] = dbg_t[0]; o[1] = dbg_t[1]; o[2] = dbg_t[2]; o[3] = dbg_t[3];
  }
}

// ------------------------------------------------------------------------------------------------ the odd position out
// With an odd number of positions (7x7 = 49) the 25th row tile of an image would hold ONE position: carried along in the
// main kernel it costs a seventh tile per wave -- 1/7 of the MFMAs and a fourth epilogue pass for three padding tiles
// and one real half tile.  Instead the main kernel runs 24 full tiles (NT = 6) and this kernel finishes position
// H*W - 1 of every image: one wave = four images x 16 output channels, two 32-row tiles whose lane halves are two
// images each (the same accumulator layout, so the same pairwise epilogue); only the taps that fall inside the image exist (4 of 9
// for the corner); operands come straight from L2 (16 B of spikes and 24 B of weights per lane and MFMA), no LDS.
// ~3 % of the main kernel's MFMA count.
constexpr int LP_TILES = 2;      // 32-row tiles (= image pairs) per wave: a weight fragment serves 2 x 3 MFMAs
template <bool RAW = false>
__global__ __launch_bounds__(256) void conv3x3_fp6_lastpos_kernel(Fp6Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HW = a.H * a.W, nchunks = a.nch0;
  const int g = blockIdx.y * 4 + wave;                      // Cout % 64 == 0: groups come in fours
  const int b0 = blockIdx.x * 2 * LP_TILES;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  if (b0 >= Bn) return;
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);       // A row -> (image parity, time step)
  const int py = a.H - 1, px = a.W - 1, p = HW - 1;
  typedef int v8i __attribute__((ext_vector_type(8)));
  v16f acc[LP_TILES][3];
#pragma unroll
  for (int i = 0; i < LP_TILES; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the taps that reach the last position (H-1, W-1) from inside the image are (dy, dx) in {-1, 0}^2 = taps 0, 1, 3, 4
  // (H, W >= 2): a fixed list, so a (chunk, tap) step is straight-line code
  for (int c = 0; c < nchunks; ++c) {
    const uint8_t* wslab = a.wq + ((long long)g * nchunks + c) * W_CHUNK_BYTES;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tap = (q >> 1) * 3 + (q & 1);
      const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
      v8i b8[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint8_t* wt = wslab + (tap * 3 + j) * W_TILE_BYTES;
        const v4i bx = *reinterpret_cast<const v4i*>(wt + lane * 16);
        const v2i by = *reinterpret_cast<const v2i*>(wt + 1024 + lane * 8);
        b8[j] = v8i{bx[0], bx[1], bx[2], bx[3], by[0], by[1], 0, 0};
      }
#pragma unroll
      for (int i = 0; i < LP_TILES; ++i) {
        const int bA = b0 + 2 * i + hsel;
        v4i av = {0, 0, 0, 0};
        if (bA < Bn)
          av = *reinterpret_cast<const v4i*>(a.in0 + ((long long)bA * nchunks + c) * HW * POS_BYTES +
                                             ((yy * a.W + xx) * T16 + tt) * 32 + 16 * half);
        const v8i a8 = {av[0], av[1], av[2], av[3], 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[j], acc[i][j], 4, 2, 0, 0x7f7f7f7f, 0,
                                                                      (int)0x82828282u);
      }
    }
  }
  const int col = lane & 31, ch = col & 15, odd = col >> 4;
  const int co = g * 16 + ch;
  const double sc = a.scale[co], bi = a.bias[co];
  const float bn_a = RAW ? 1.0f : a.bn_a[co], bn_b = RAW ? 0.0f : a.bn_b[co];
  // tiles are finished in pairs exactly like the main kernel's: even lanes tile ia, odd lanes tile ia + 1
#pragma unroll
  for (int ia = 0; ia < LP_TILES; ia += 2) {
    float x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pr[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const v2u q = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ia][j][r]), __float_as_uint(acc[ia + 1][j][r]),
                                                       false, false);
        pr[j] = fmaf(__uint_as_float(q[0]), 32.0f, __uint_as_float(q[1]));
      }
      const double s1 = fma((double)pr[0], 1024.0, (double)pr[1]);
      const double s = fma(s1, 1024.0, (double)pr[2]);
      x[r] = (float)fma(s, sc, bi);
    }
    const int b = b0 + 2 * (ia + odd) + half;          // accumulator lane-half == image within the tile's pair
    const bool ok = b < Bn;
    if constexpr (RAW) {
      if (ok) {
        float* dst = a.pre + ((long long)b * HW + p) * a.Cout + co;
        const long long tstride = (long long)a.B * HW * a.Cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * tstride] = x[r];
      }
      continue;
    }
    const long long vidx = ((long long)(ok ? b : 0) * a.Cout + co) * HW + p;
    float v = (a.v_io && ok) ? a.v_io[vidx] : 0.f;
    unsigned mybits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool s = spk_lif_step_default(v, fmaf(x[r], bn_a, bn_b)) && ok;
      mybits |= s ? (1u << r) : 0u;
    }
    const unsigned cnt = __popc(mybits);
    const unsigned bitsv = spk_transpose16_rows(mybits, lane);
    if (a.v_io && ok) a.v_io[vidx] = v;
    if (a.out_cnt && ok) a.out_cnt[(((long long)b * (a.Cout >> 5) + (co >> 5)) * HW + p) * 32 + (co & 31)] = (uint8_t)cnt;
    if (ok) {
      auto spread8 = [](unsigned q) -> unsigned {
        q = (q | (q << 12)) & 0x000f000fu;
        q = (q | (q << 6)) & 0x03030303u;
        q = (q | (q << 3)) & 0x11111111u;
        return q << 1;
      };
      uint2 o;
      o.x = spread8(bitsv & 0xffu);
      o.y = spread8((bitsv >> 8) & 0xffu);
      const int co0 = g * 16;
      uint8_t* dst = a.out + ((((long long)b * (a.Cout >> 6) + (co0 >> 6)) * HW + p) * T16 + (lane & 15)) * 32 + ((co0 & 63) >> 1);
      *reinterpret_cast<uint2*>(dst) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight packing
// one block per output channel: channel maximum -> shift s, then every weight -> 6 balanced radix-32 digits, written
// as the per-lane 24-byte B fragments of the MFMA (lane = k-half * 32 + plane parity * 16 + channel, 32 six-bit codes,
// little-endian; bytes 0..15 in the ds_read_b128 part of the tile, bytes 16..23 in its ds_read_b64 part; a chunk slab
// is padded to whole KiB DMA pieces)
__device__ __forceinline__ void pack_fp6_channel(const float* __restrict__ w, const float* __restrict__ bias,
                                                 uint8_t* __restrict__ wq, double* __restrict__ scale,
                                                 double* __restrict__ bias_d, int Cout, int Cin, int w_cl, const int co) {
  __shared__ float smax[256];
  const int n = Cin * 9;
  const float* wc = w + (long long)co * n;     // w_cl: the channel's weights are stored [3][3][Cin] (channels-last memory format)
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(wc[i]));
  smax[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    __syncthreads();
  }
  m = smax[0];
  int e = 0;
  if (m > 0.f) frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)  ->  m < 2^e
  const int sh = 29 - e;                      // |w| * 2^sh < 2^29 <= 16.5 * 32^5
  if (threadIdx.x == 0) { scale[co] = ldexp(1.0, -sh); bias_d[co] = bias ? (double)bias[co] : 0.0; }
  const int nchunks = Cin / CK, g = co >> 4, ch = co & 15;
  // Every thread converts elements to their six digit codes (bytes in LDS, [plane][ci * 9 + tap]); then one thread packs the 16
  // k values of half a record (96 bits = three words per plane).  (One thread per whole record -- 72 active threads of 256
  // for the 256 -> 512 layer, each converting 32 elements in double precision -- took 25 us per layer in every training iteration.)
  extern __shared__ uint8_t codes[];                            // [6][n]
  for (int i = threadIdx.x; i < n; i += 256) {
    // |w| 2^sh < 2^29: the scaled value is exact in fp32 and rintf rounds it like the fp64 form did
    long long q = (long long)rintf(ldexpf(wc[w_cl ? (i % 9) * Cin + i / 9 : i], sh));
#pragma unroll
    for (int p = 5; p >= 1; --p) {
      const int r = (int)(((q + 16) & 31) - 16);
      codes[p * n + i] = (uint8_t)((r < 0 ? 0x20u : 0u) | (unsigned)(r < 0 ? -r : r));
      q = (q - r) >> 5;
    }
    const int r0 = (int)q;                                      // in [-16, 16]
    codes[i] = (uint8_t)((r0 < 0 ? 0x20u : 0u) | (unsigned)(r0 < 0 ? -r0 : r0));
  }
  __syncthreads();
  // one record = the 32 k values of (chunk, tap, k-half); unit = (record, half of it)
  for (int unit = threadIdx.x; unit < nchunks * 36; unit += 256) {
    const int hf = unit & 1, rec = unit >> 1, kh = rec & 1, tap = (rec >> 1) % 9, c = rec / 18;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      unsigned long long lo = 0;                                // bits 0..63 of the 96
      unsigned hi = 0;                                          // bits 64..95
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int ci = c * CK + kh * 32 + hf * 16 + j;
        const unsigned long long code = codes[p * n + ci * 9 + tap];
        const int bit = 6 * j;
        if (bit < 64) lo |= code << bit;
        if (bit + 6 > 64) hi |= (unsigned)(bit >= 64 ? code << (bit - 64) : code >> (64 - bit));
      }
      const int ct = p >> 1, ln = kh * 32 + (p & 1) * 16 + ch;
      uint8_t* tile = wq + (long long)(g * nchunks + c) * W_CHUNK_BYTES + (tap * 3 + ct) * W_TILE_BYTES;
      unsigned* d16 = reinterpret_cast<unsigned*>(tile + ln * 16);
      unsigned* d8 = reinterpret_cast<unsigned*>(tile + 1024 + ln * 8);
      const unsigned w0 = (unsigned)lo, w1 = (unsigned)(lo >> 32), w2 = hi;
      if (hf == 0) { d16[0] = w0; d16[1] = w1; d16[2] = w2; }
      else { d16[3] = w0; d8[0] = w1; d8[1] = w2; }
    }
  }
}

// fp32 spikes [T,B,C,HW] <-> nibble-packed C4 [B][C/64][HW][T][32] (tests, module boundaries)
__global__ void spikes_to_fp4_kernel(const float* __restrict__ s, uint8_t* __restrict__ o, int T, int B, int C, int HW) {
  const long long total = (long long)B * (C / 64) * HW * T * 32;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int byte = (int)(i & 31);
    long long r = i >> 5;
    const int t = (int)(r % T); r /= T;
    const int p = (int)(r % HW); r /= HW;
    const int cc = (int)(r % (C / 64));
    const int b = (int)(r / (C / 64));
    const int c0 = cc * 64 + 2 * byte;
    const float s0 = s[(((long long)t * B + b) * C + c0) * HW + p], s1 = s[(((long long)t * B + b) * C + c0 + 1) * HW + p];
    o[i] = (uint8_t)((s0 != 0.f ? 0x02 : 0) | (s1 != 0.f ? 0x20 : 0));
  }
}
__global__ void fp4_to_spikes_kernel(const uint8_t* __restrict__ q, float* __restrict__ s, int T, int B, int C, int HW) {
  const long long total = (long long)T * B * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    long long r = i / HW;
    const int c = (int)(r % C); r /= C;
    const int b = (int)(r % B);
    const int t = (int)(r / B);
    const uint8_t by = q[((((long long)b * (C / 64) + c / 64) * HW + p) * T + t) * 32 + (c % 64) / 2];
    s[i] = ((by >> (4 * (c & 1))) & 0xf) ? 1.0f : 0.0f;
  }
}

__global__ __launch_bounds__(256) void pack_fp6_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                       uint8_t* __restrict__ wq, double* __restrict__ scale,
                                                       double* __restrict__ bias_d, int Cout, int Cin, int w_cl) {
  pack_fp6_channel(w, bias, wq, scale, bias_d, Cout, Cin, w_cl, blockIdx.x);
}

// the channels of up to eight layers in one launch (spk_den_pack_weight_fp6_cl_multi: the training iteration re-packs every
// spike-input layer's weights once per optimizer step -- five launches of 6-18 us before)
constexpr int PK_MULTI_MAX = 8;
struct PackMulti {
  const float* w[PK_MULTI_MAX]; const float* bias[PK_MULTI_MAX]; uint8_t* wq[PK_MULTI_MAX]; double* scale[PK_MULTI_MAX];
  double* bias_d[PK_MULTI_MAX];
  int Cout[PK_MULTI_MAX], Cin[PK_MULTI_MAX], first[PK_MULTI_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void pack_fp6_multi_kernel(PackMulti m) {
  int L = 0;
  while (L + 1 < m.n && (int)blockIdx.x >= m.first[L + 1]) ++L;
  pack_fp6_channel(m.w[L], m.bias[L], m.wq[L], m.scale[L], m.bias_d[L], m.Cout[L], m.Cin[L], 1, (int)blockIdx.x - m.first[L]);
}

}  // namespace

extern "C" long long spk_den_packed_weight_fp6_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || (Cout % 16) || (Cin % CK)) return -1;
  return (long long)(Cout / 16) * (Cin / CK) * W_CHUNK_BYTES;
}

static int pack_weight_fp6(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d, int Cout, int Cin,
                           int w_cl, hipStream_t stream) {
  if (!w || !wq || !scale || !bias_d || Cout <= 0 || Cin <= 0) return SPK_ERR_ARG;
  if ((Cout % 16) || (Cin % CK)) return SPK_ERR_UNSUPPORTED;
  if ((long long)6 * Cin * 9 > spk_lds_limit()) return SPK_ERR_UNSUPPORTED;   // (one channel's six digit planes are staged in LDS)
  hipLaunchKernelGGL(pack_fp6_kernel, dim3(Cout), dim3(256), (size_t)6 * Cin * 9, stream, w, bias, wq, scale, bias_d, Cout, Cin,
                     w_cl);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_den_pack_weight_fp6(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d,
                                       int Cout, int Cin, hipStream_t stream) {
  return pack_weight_fp6(w, bias, wq, scale, bias_d, Cout, Cin, 0, stream);
}

// the same packing of a weight kept in the channels-last memory format, [Cout][3][3][Cin] -- what the training path keeps its
// convolution parameters in (the library's NHWC kernels and the native gradients read and write that layout): packed every
// iteration, it cost a layout copy per layer and iteration in front of this launch
extern "C" int spk_den_pack_weight_fp6_cl(const float* w_cl, const float* bias, uint8_t* wq, double* scale, double* bias_d,
                                          int Cout, int Cin, hipStream_t stream) {
  return pack_weight_fp6(w_cl, bias, wq, scale, bias_d, Cout, Cin, 1, stream);
}

// n layers' channels-last weights in ONE launch (host arrays of n <= 8 entries; bias[i] may be null); per layer the result is
// spk_den_pack_weight_fp6_cl's, byte for byte
extern "C" int spk_den_pack_weight_fp6_cl_multi(const float* const* w_cl, const float* const* bias, uint8_t* const* wq,
                                                double* const* scale, double* const* bias_d, const int* Cout, const int* Cin,
                                                int n, hipStream_t stream) {
  if (!w_cl || !wq || !scale || !bias_d || !Cout || !Cin || n <= 0) return SPK_ERR_ARG;
  if (n > PK_MULTI_MAX) return SPK_ERR_UNSUPPORTED;
  PackMulti m;
  m.n = n;
  int blocks = 0, cin_max = 0;
  for (int i = 0; i < n; ++i) {
    if (!w_cl[i] || !wq[i] || !scale[i] || !bias_d[i] || Cout[i] <= 0 || Cin[i] <= 0) return SPK_ERR_ARG;
    if ((Cout[i] % 16) || (Cin[i] % CK)) return SPK_ERR_UNSUPPORTED;
    m.w[i] = w_cl[i]; m.bias[i] = bias ? bias[i] : nullptr; m.wq[i] = wq[i]; m.scale[i] = scale[i]; m.bias_d[i] = bias_d[i];
    m.Cout[i] = Cout[i]; m.Cin[i] = Cin[i]; m.first[i] = blocks;
    blocks += Cout[i];
    cin_max = Cin[i] > cin_max ? Cin[i] : cin_max;
  }
  for (int i = n; i <= PK_MULTI_MAX; ++i) m.first[i] = blocks;
  if ((long long)6 * cin_max * 9 > spk_lds_limit()) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_fp6_multi_kernel, dim3(blocks), dim3(256), (size_t)6 * cin_max * 9, stream, m);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

namespace {
template <bool RAW>
int launch_fp6(const uint8_t* in_c4, int nch, const uint8_t* wq, const double* scale, const double* bias_d, const float* bn_a,
               const float* bn_b, float* v_inout, uint8_t* out_c4, uint8_t* out_counts, float* pre, const int* n_dyn, int T,
               int B, int H, int W, int Cout, hipStream_t stream) {
  if (T != T16 || (Cout % 64)) return SPK_ERR_UNSUPPORTED;
  const int ntiles = (H * W + 1) / 2;
  // latents that do not fit one item (8x8): two row bands per image, 4 tiles per wave each (see the kernel)
  const bool bands = (ntiles + 3) / 4 > 7 && (H % 2) == 0 && (H / 2) * W <= 32 && H >= 4;
  const int Hin = bands ? H / 2 + 1 : H;
  const int npa = (Hin * ((W + 1) / 2) + 3) / 4;
  const size_t lds = 2 * ((size_t)((Hin + 2) * (W + 1) + 1) * POS_BYTES + W_CHUNK_BYTES);
  if ((!bands && (ntiles + 3) / 4 > 7) || npa > NPA || lds > 160 * 1024) return SPK_ERR_UNSUPPORTED;
  if (bands && ((Hin * W + 2 * (W + 1)) * POS_BYTES >= 32768)) return SPK_ERR_UNSUPPORTED;     // piece-table field widths
  // odd position count with at most 24 full tiles: 6 tiles per wave + the last-position kernel (see there)
  const bool split_last = !bands && ((H * W) & 1) && (H * W) / 2 <= 24 && H >= 2 && W >= 2;
  Fp6Args a;
  a.in0 = in_c4; a.nch0 = nch; a.wq = wq; a.scale = scale; a.bias = bias_d; a.bn_a = bn_a; a.bn_b = bn_b;
  a.out = out_c4; a.v_io = v_inout; a.out_cnt = out_counts; a.pre = pre; a.n_dyn = n_dyn; a.B = B; a.H = H; a.W = W; a.Cout = Cout;
  const int cus = spk_cu_count();
  const int G = Cout / 16, total = B * G * (bands ? 2 : 1);
  dim3 grid(total < cus ? total : cus), blk(256);          // persistent: one workgroup per CU
  // XCD-aware walk: the largest power-of-two group count whose packed weights (gx * nch slabs) fit ~1.5 MB of an XCD's
  // 4 MB L2, if the shape tiles exactly (see decode() in the kernel)
  // Measured on the conv4 shape (B = 256): L2-miss reads 206 -> 111 MB and HBM-side writes 218 -> 56 MB per launch at
  // the same launch time (with 7 tiles per wave it was ~2 % slower; rotating the chunk order per workgroup to spread the
  // slab requests of an XCD made it slower still).  option fp6_xcd_walk = 0 selects the image-major walk.
  const bool xcd_walk = spk_opt(SPK_OPT_FP6_XCD_WALK) != 0;
  a.gx = 0;
  if (xcd_walk && !n_dyn && !bands && (grid.x & 7) == 0) {      // (a device-side batch count / row bands walk image-major)
    const int S = grid.x / 8;
    int gx = 1;
    while (gx * 2 <= G && (long long)gx * 2 * nch * W_CHUNK_BYTES <= 1536 * 1024) gx *= 2;
    const int nsets = G / gx;
    if (gx >= 4 && G % gx == 0 && nsets <= 8 && 8 % nsets == 0 && S % gx == 0 && B % ((8 / nsets) * (S / gx)) == 0) a.gx = gx;
  }
  if (bands) {
    hipLaunchKernelGGL((conv3x3_fp6_kernel<4, RAW, true>), grid, blk, lds, stream, a);
  } else if (split_last) {
    const bool eight = spk_opt(SPK_OPT_FP6_WAVES) == 8;
    if (eight && (H * ((W + 1) / 2) + 7) / 8 <= 4)
      hipLaunchKernelGGL((conv3x3_fp6_kernel<3, RAW, false, 8>), grid, dim3(512), lds, stream, a);
    else
      hipLaunchKernelGGL((conv3x3_fp6_kernel<6, RAW>), grid, blk, lds, stream, a);
    SPK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv3x3_fp6_lastpos_kernel<RAW>, dim3((B + 2 * LP_TILES - 1) / (2 * LP_TILES), G / 4), blk, 0, stream, a);
  } else {
    hipLaunchKernelGGL((conv3x3_fp6_kernel<7, RAW>), grid, blk, lds, stream, a);
  }
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
}  // namespace

extern "C" int spk_den_conv3x3_mfma_fp6(const uint8_t* in_c4, int nch, const uint8_t* wq, const double* scale,
                                        const double* bias_d, const float* bn_a, const float* bn_b, float* v_inout,
                                        uint8_t* out_c4, uint8_t* out_counts, int T, int B, int H, int W, int Cout,
                                        const int* n_dyn_or_null, hipStream_t stream) {
  if (!in_c4 || nch <= 0 || !wq || !scale || !bias_d || !bn_a || !bn_b || !out_c4 || B <= 0 || H <= 0 || W <= 0 ||
      Cout <= 0)
    return SPK_ERR_ARG;
  return launch_fp6<false>(in_c4, nch, wq, scale, bias_d, bn_a, bn_b, v_inout, out_c4, out_counts, nullptr, n_dyn_or_null, T,
                           B, H, W, Cout, stream);
}

extern "C" int spk_den_conv3x3_fp6_raw(const uint8_t* in_c4, int nch, const uint8_t* wq, const double* scale,
                                       const double* bias_d, float* pre_nhwc, int T, int B, int H, int W, int Cout,
                                       hipStream_t stream) {
  if (!in_c4 || nch <= 0 || !wq || !scale || !bias_d || !pre_nhwc || B <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return SPK_ERR_ARG;
  return launch_fp6<true>(in_c4, nch, wq, scale, bias_d, nullptr, nullptr, nullptr, nullptr, nullptr, pre_nhwc, nullptr, T, B,
                          H, W, Cout, stream);
}

namespace {
// channels-last fp32 spikes [T][B][HW][C] -> C4: one thread = 4 consecutive channels of one (t, b, hw) = one 16-bit word
__global__ void spikes_nhwc_to_fp4_kernel(const float* __restrict__ s, uint8_t* __restrict__ o, int T, int B, int C, int HW) {
  const long long total = (long long)T * B * HW * (C / 4);
  const int Q = C / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long long row = i / Q;                       // (t * B + b) * HW + hw
    const int hw = (int)(row % HW);
    const long long tb = row / HW;
    const int b = (int)(tb % B), t = (int)(tb / B);
    const float4 v = reinterpret_cast<const float4*>(s)[i];
    const unsigned w = (v.x != 0.f ? 0x2u : 0u) | (v.y != 0.f ? 0x20u : 0u) | (v.z != 0.f ? 0x200u : 0u) |
                       (v.w != 0.f ? 0x2000u : 0u);
    const int c = q * 4;
    uint8_t* dst = o + ((((long long)b * (C >> 6) + (c >> 6)) * HW + hw) * T + t) * 32 + ((c & 63) >> 1);
    *reinterpret_cast<uint16_t*>(dst) = (uint16_t)w;
  }
}
}  // namespace

namespace {
// the same conversion with the per-neuron spike COUNTS over T as a by-product (fp32 [B][HW][C], channels-last): one thread = 4
// consecutive channels of one (b, hw) for all T steps.  The training step's last layer convolves its weight gradient with these
// counts (ops.SpikeConvMeanTrainFunction); they were a separate reduction over the [T,B,320,7,7] spike tensor (32 us at B = 32).
__global__ void spikes_nhwc_to_fp4_counts_kernel(const float* __restrict__ s, uint8_t* __restrict__ o, float* __restrict__ cnt,
                                                 int T, int B, int C, int HW) {
  const int Q = C / 4;
  const long long total = (long long)B * HW * Q, plane = (long long)B * HW * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const long long row = i / Q;                       // b * HW + hw
    const int hw = (int)(row % HW), b = (int)(row / HW);
    const int c = q * 4;
    uint8_t* dst = o + (((long long)b * (C >> 6) + (c >> 6)) * HW + hw) * T * 32 + ((c & 63) >> 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      const float4 v = reinterpret_cast<const float4*>(s)[t * plane + i];
      const unsigned w = (v.x != 0.f ? 0x2u : 0u) | (v.y != 0.f ? 0x20u : 0u) | (v.z != 0.f ? 0x200u : 0u) |
                         (v.w != 0.f ? 0x2000u : 0u);
      acc.x += v.x != 0.f ? 1.f : 0.f; acc.y += v.y != 0.f ? 1.f : 0.f;
      acc.z += v.z != 0.f ? 1.f : 0.f; acc.w += v.w != 0.f ? 1.f : 0.f;
      *reinterpret_cast<uint16_t*>(dst + t * 32) = (uint16_t)w;
    }
    reinterpret_cast<float4*>(cnt)[i] = acc;
  }
}
}  // namespace

extern "C" int spk_spikes_nhwc_to_fp4_counts(const float* spikes_nhwc, uint8_t* out_c4, float* counts_nhwc, int T, int B, int C,
                                             int HW, hipStream_t stream) {
  if (!spikes_nhwc || !out_c4 || !counts_nhwc || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 64) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)B * HW * (C / 4);
  hipLaunchKernelGGL(spikes_nhwc_to_fp4_counts_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)),
                     dim3(256), 0, stream, spikes_nhwc, out_c4, counts_nhwc, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_spikes_nhwc_to_fp4(const float* spikes_nhwc, uint8_t* out_c4, int T, int B, int C, int HW,
                                      hipStream_t stream) {
  if (!spikes_nhwc || !out_c4 || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 64) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)T * B * HW * (C / 4);
  hipLaunchKernelGGL(spikes_nhwc_to_fp4_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)),
                     dim3(256), 0, stream, spikes_nhwc, out_c4, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_spikes_to_fp4(const float* spikes, uint8_t* out_c4, int T, int B, int C, int HW, hipStream_t stream) {
  if (!spikes || !out_c4 || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 64) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)B * (C / 64) * HW * T * 32;
  hipLaunchKernelGGL(spikes_to_fp4_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)), dim3(256),
                     0, stream, spikes, out_c4, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_fp4_to_spikes(const uint8_t* in_c4, float* spikes, int T, int B, int C, int HW, hipStream_t stream) {
  if (!spikes || !in_c4 || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 64) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)T * B * C * HW;
  hipLaunchKernelGGL(fp4_to_spikes_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)), dim3(256),
                     0, stream, in_c4, spikes, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
