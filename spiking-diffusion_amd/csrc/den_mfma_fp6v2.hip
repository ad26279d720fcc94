// Second-generation 3x3 spike convolution for the denoiser's conv2..conv5 (R/snn_model/vq_diffusion.py:166-184,201-204;
// SURVEY.md §8 a8) on the CDNA4 block-scaled MFMA -- the sampler's fast path (7x7 latents, fresh LIF state).  Same results,
// bit for bit, as den_mfma_fp6.hip (the correctly rounded value of the exact 29-bit fixed-point dot product, then the
// reference's fp32 BN and LIF arithmetic); what changes is how much matrix and vector work it takes to get there.
//
// 1. DIGIT PAIRS SHARE AN ACCUMULATOR.  v_mfma_scale_f32_32x32x64_f8f6f4 takes one E8M0 scale per operand row / column and
//    32-wide K block, i.e. per lane.  The two K halves of an instruction carry the SAME 32 input channels with two
//    adjacent radix-32 digits of the weights, the even digit scaled by 2^8 and the odd one by 2^3 (e2m3 holds d/8): the
//    fp32 accumulator then holds 32 * D_even + D_odd directly -- an integer below 4608 * 528 < 2^24, still exact.  The
//    column tile is 32 output channels of one digit pair, so the three values of a neuron-step (pairs 01, 23 and the
//    fifth digit) sit in ONE lane: no cross-lane exchange in the epilogue, and all 64 lanes scan their own neuron.
// 2. FOUR DIGITS ON THE MATRIX CORES, THE LAST TWO ONLY WHERE THEY MATTER.  The fifth and sixth digit move a pre-activation by
//    at most 528 units of 2^-s per ACTIVE input.  The main kernel multiplies the four leading digits (18 instead of 27 MFMAs
//    per row tile and 32-channel chunk: two digit pairs per tap), counts the active inputs of every row while their fragments
//    pass through its registers (four v_bcnt per fragment), recombines in fp32, and CERTIFIES every spike decision: every
//    neuron carries a running bound D_t on |h_approx - h_exact| (the dropped digits for the counted inputs of each step, the
//    fp32 recombination and every rounding of BN / LIF on either path, see "Certification" below) and is flagged when its
//    membrane potential ever comes within D_t of the threshold (1 - 6e-4 of the neurons at the denoiser's firing rates of
//    3 - 7 %).  The fixup launch recomputes the flagged neurons EXACTLY (all six digits as the pack kernel's int32 quantised
//    weights, 64-bit sums, fp64 recombination, one rounding: the arithmetic of den_mfma_fp6.hip) and patches their spikes.
//    Unflagged neurons provably emit the spikes the exact arithmetic would; flagged ones are the exact arithmetic.  Membrane
//    potentials are not an output here (fresh state in, nothing written back): callers that carry LIF state, and the
//    training forward, use den_mfma_fp6.hip.  (-DSPK_V2_D4=1 builds the earlier form: five digits on the matrix cores -- 23
//    MFMAs, the fifth pairing two taps per instruction -- and a bound that takes every input as active; it flags 2 - 3x fewer
//    neurons and is 8 - 10 % slower end to end.)
// 3. A WORK ITEM = one image x 32 output channels, K chunk = 32 input channels: the spike slab of an image is fetched half
//    as often, a workgroup keeps ONE channel group for the whole launch (its BN / margin constants are loaded once, its
//    weight slabs stay L2-hot), and two images + two 35 KB weight slabs need 107 KB of LDS.
//
// Spike layout "S32": [B][C/32][H*W][T = 16][16 B], channel c of a group in byte (c % 32) / 2, low nibble first, e2m1 codes
// 0x0 / 0x2.  The last position of a 7x7 latent (the 49th: 24 full 32-row tiles + 1) is computed exactly by the tail
// launch from the same packed weights plus two sixth-digit tiles stored for its four taps.
#include "den_common.h"
#include "../../include/spkdiff.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int T16 = 16;
constexpr int CK = 32;                                   // input channels per K chunk
constexpr int POSB = T16 * CK / 2;                       // 256 B per latent position per chunk
constexpr int WT = 64 * 24;                              // one B tile: 64 lanes x 32 six-bit codes
constexpr int N_PAIR = 18;                               // tiles 0..17: (tap, digit pair 01 | 23)
constexpr int N_D4 = 5;                                  // tiles 18..22: fifth digit, taps (0,1) (2,3) (4,5) (6,7) (8,-)
constexpr int N_MAIN = N_PAIR + N_D4;
constexpr int N_L5 = 2;                                  // tiles 23, 24: sixth digit of taps (0,1) and (3,4), last position only
#ifndef SPK_V2_D4
#define SPK_V2_D4 0             // 1: the fifth digit is multiplied on the matrix cores as well (23 instead of 18 MFMAs per tile and chunk)
#endif
constexpr bool USE_D4 = SPK_V2_D4 != 0;
constexpr int N_MM = USE_D4 ? N_MAIN : N_PAIR;           // tiles the main launch multiplies (the tail launches read all of them)
constexpr int NACC = USE_D4 ? 3 : 2;                     // accumulators per row tile
constexpr int W_PIECES = (N_MM * WT + 1023) / 1024;      // one-KiB DMA pieces per chunk (27 / 35)
constexpr int W_LDS = W_PIECES * 1024;
constexpr int W_SLAB = ((N_MAIN + N_L5) * WT + 1023) / 1024 * 1024;   // bytes per (channel group, chunk) in memory

// Workspace layout (spk_den_fp6v2_flag_words): ws[0] live count of flagged neurons, ws[1] the count published for the tail launch,
// ws[2 .. 2 + FLAG_CAP) their ids, then the overflow bitmap (one bit per neuron of the layer), then the hand-over ticket.  The id list
// holds the first `flag_cap` <= FLAG_CAP flagged neurons of a launch (a per-call argument: the parity suite runs the overflow path with
// 64 and 0); every further one sets its bit in the bitmap, which the tail launch scans and clears.  The LAYOUT never depends on flag_cap.
constexpr unsigned FLAG_CAP = 1u << 20;

struct V2Args {
  const uint8_t* in0; int nch;               // S32 spikes, nch = Cin / 32
  const uint8_t* wq; const double* scale; const double* bias; const float* wl1;
  const float* bn_a; const float* bn_b;
  uint8_t* out; uint8_t* out_cnt;
  unsigned* flags;                           // ws[0]: number of flagged neurons (live), ws[1]: the count of the last finished
                                             //  launch (published by the hand-over or the re-arming), ws[2..2+cap): their ids,
  unsigned flag_cap;                         //  then the overflow bitmap (one bit per neuron; used only beyond cap), then the
  long long ticket_idx;                      //  ticket of the hand-over (ws[1] = the count published for the tail launch)
  int handover;                              // 1: the main launch publishes the count (fp6v2_handover); 0: the tail launches read
                                             //    ws[0] and the last-position launch, which follows the repair launch, re-arms it
  const int* qtab;                           // quantised weights int32 [Cout][9][Cin] (exact recomputation)
  const int* n_dyn;
  // position lists (spk_select_needed, one radius): 64-byte record per image slot; the slots by tile count (classes
  // 1..6 tiles per wave): cls_cnt[k - 1] slots listed in cls_list[(k - 1) * B ..]
  const uint8_t* need; const int* cls_cnt; const int* cls_list;
  int B, Cout, Cin;
  int gx, nsets;                             // XCD-aware walk (gx > 0) or flat walk (gx == 0)
  unsigned long long* dbg_out;               // (-DSPK_V2_DUO_DBG builds: per-workgroup time stamps; otherwise null)
  unsigned* cu_slots;                        // duo form: one arrival counter per CU (2048 words behind the ticket)
  unsigned* item_ctr;                        // duo form: item claim counters, one per (channel group, XCD partition): 128 words behind them
  float* zstage;                             // deferred-scan form: 96 KB per workgroup (an item's pre-activations between two K loops)
  int duo_delay;                             // duo form: head start of a CU's first workgroup over its second, in 10 ns ticks (0: none)
  int fix_lds;                               // tail launches: 1 = the repair stages a neuron's 9 x Cin weights in (dynamic) LDS (SPK_V2_FIX_LDS)
};

#ifndef SPK_FP6_PRE
#define SPK_FP6_PRE "s_nop 1\n\t"
#endif
#define SPK_MFMA2(CLS, acc, av, bv, sa, sb)                                                                          \
  asm volatile(SPK_FP6_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2"  \
               : "+" CLS(acc) : "v"(av), "v"(bv), "v"(sa), "v"(sb))
#define SPK_MFMA2_Z(CLS, acc, av, bv, sa, sb)                                                                        \
  asm volatile(SPK_FP6_PRE "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:2"   \
               : "=&" CLS(acc) : "v"(av), "v"(bv), "v"(sa), "v"(sb))

#ifndef SPK_V2_AGPR12
#define SPK_V2_AGPR12 0         // accumulators in AGPRs with three waves per SIMD (168 registers per wave)
#endif
#ifndef SPK_V2_DBG
#define SPK_V2_DBG 0            // experiments only: 1 = no steady-state DMA, 4 = no epilogue (results are wrong), 32 = zero
                                // certification margin (nothing flagged: shows what the exact recomputation repairs),
                                // 64 = the tail launch leaves the flag bitmap alone (flagged neurons can be counted), 512 = no chunk barrier in the K loop,
                                // 128 = every workgroup stamps {s_memtime, s_memrealtime} around its item loop into the id
                                // list (shader clock under this kernel's own load = d memtime / d memrealtime * 100 MHz)
#endif
#ifndef SPK_V2_KMIN
#define SPK_V2_KMIN 2           // listed positions: an item of fewer tiles per wave still costs about this many (operand copies)
#endif
#ifndef SPK_V2_SPARE
#define SPK_V2_SPARE 1.0f       // four-digit form: factor on the certification bound (2.0f = the first builds' spare factor)
#endif
#ifndef SPK_V2_REC_INLOOP
#define SPK_V2_REC_INLOOP 1     // record counting inside the K-loop's MFMA stream (0: in front of it, as in round 2)
#endif
#ifndef SPK_V2_REC_STEP
#define SPK_V2_REC_STEP 13      // the step whose gap takes the popcounts (the reads are issued at step 1; 7: no gain, 13: +0.4 %)
#endif
#ifndef SPK_V2_STAGE2
#define SPK_V2_STAGE2 1         // four-digit form: flagged lanes are re-examined with the per-step running bound (see the epilogue)
#endif
#ifndef SPK_V2_LAG_DEFAULT
#define SPK_V2_LAG_DEFAULT 0    // 1: full 7x7 batches run the staggered form (conv3x3_fp6v2_lag_kernel); SPKDIFF_V2_LAG=0/1 overrides
#endif
#ifndef SPK_V2_NAGPR8
#define SPK_V2_NAGPR8 0         // two waves per SIMD: accumulator tiles kept in AGPRs.  0 (round 4): all in VGPRs -- 243 registers, no
                                // scratch, and the scan reads its operands directly instead of through 32 v_accvgpr_read per tile
                                // (-8.6 % vector instructions per item; dense reverse process 91.2 / 92.4 -> 90.6 / 90.2 ms on one
                                // box, profiles/r4_ab_kernel_variants.txt).  8: the round-2/3 form (128 + 105 registers)
#endif
#ifndef SPK_V2_MERGE_FULL
#define SPK_V2_MERGE_FULL 1     // full 7x7 batches also run repair + last position as ONE tail launch (hand-over by the main launch).
                                // Round 2 measured this 1.3 % slower; with the second certification stage (half the repairs) it is
                                // 1.6 % FASTER (90.3 / 90.2 against 91.2 / 92.4 ms; both changes: 89.3 / 89.4).  0: two launches
#endif
#ifndef SPK_V2_MASKSTORE
#define SPK_V2_MASKSTORE 0      // 1: the sixteen spike decisions of a tile stay what the compare makes them -- 64-bit lane masks in
                                // SGPRs, i.e. ALREADY transposed -- and reach the lanes that store them through 32 v_writelane + one
                                // v_permlane16_swap instead of a per-lane word (24 instructions) + spk_transpose16_rows (14 + the
                                // exchanges).  Built, bit-equal on every fp6v2 test, and measured (round 4, three alternating passes on
                                // one box): 90.47 / 90.73 / 90.46 ms per dense reverse process against 90.41 / 90.35 / 90.15 -- 0.2 %
                                // SLOWER (a v_writelane costs a full vector issue slot plus the hazard nop hipcc puts around inline
                                // assembly): not adopted
#endif
#ifndef SPK_V2_NMAX_LDS
#define SPK_V2_NMAX_LDS 1       // the first certification stage reads max_t n_t of its position from LDS (one atomic per (position, step)
                                // in the per-item count pass) instead of sixteen counts + twelve v_max per tile and lane.  0: rounds 2-3
#endif
#ifndef SPK_V2_FIX_LDS
#define SPK_V2_FIX_LDS 0        // 1 (built in round 5, measured SLOWER, off): the exact repair of a flagged neuron stages the neuron's 9 x Cin
                                // quantised weights in LDS once (coalesced) and a lane = (step t, quarter of a 32-channel chunk) takes 4 bytes of
                                // the spike record and eight weights from LDS -- 54 KB instead of 332 KB of requests through the CU's texture
                                // path per neuron of the 512-channel layers.  Same box, two passes, bit-equal: den.conv5 launches 376.5 / 375.8
                                // against 355.5 / 356.7 us, den.conv4 374.7 / 374.5 against 367.1 / 367.8, dense reverse process 89.7 / 89.4
                                // against 88.2 / 88.0 ms: the copy + barrier per neuron cost more than the sixteen-fold weight requests, which
                                // the L1 serves (profiles/r5_ab_kernel_variants.txt (8))
#endif
#ifndef SPK_V2_SIGNBITS
#define SPK_V2_SIGNBITS 1       // the sixteen spike bits of a lane are shifted in from the sign of h - 1 (h - 1 exists for the certification: one
                                // v_alignbit per step, one bit reversal per tile) instead of sixteen selects on the compare masks + eight
                                // three-way ORs: -23 vector instructions and -4 hazard nops per tile; dense reverse process 88.84 -> 88.33 ms
                                // (same box, two passes, bit-equal: profiles/r4_ab_kernel_variants.txt (13)).  0: the select form
#endif
#ifndef SPK_V2_LP_PAIRS
#define SPK_V2_LP_PAIRS 1       // last-position part of the tail launch: image pairs (32-row tiles) per unit.  2 halves the weight-tile
                                // reads (327 -> 164 MB through L2 for the 256 -> 512 layer at B = 256) and needs 146 + 96 registers
                                // (two waves per SIMD instead of three, for the repair part of the same launch too): measured
                                // +1.8 % on the dense reverse process (86.8 -> 88.4 ms, same box, profiles/r4_ab_kernel_variants.txt)
                                // -- the launch is bound by the depth of its chains of dependent reads, not by L2 bandwidth
#endif
#ifndef SPK_V2_GX_KB
#define SPK_V2_GX_KB 2560       // XCD-aware walk: packed weights of the channel groups one XCD keeps (its L2 is 4 MB).  Round 6, same box
                                // (profiles/r6_ab_kernel_variants.txt (2)): 1536 (rounds 2-5: four sets of 2 / 4 groups for conv5 / conv4, every
                                // image slab fetched by four XCDs) -> 2560 (two sets of 4 / 8 groups, 2.4 MB of weights per L2): L2-miss traffic
                                // of the conv5 / conv4 launches 320 / 194 -> 214 / 150 MB at the SAME time (88.2 ms per batch either way);
                                // 6144 (one set, 4.9 MB of weights per L2): 312 / 346 MB -- the weights no longer stay resident
#endif
#ifndef SPK_V2_HALF_FILL
#define SPK_V2_HALF_FILL 2      // the small-batch split is taken while B x Cout / 32 x this <= workgroups (2: the halves still fit one per CU)
#endif
#ifndef SPK_V2_LPS_MIN_B
#define SPK_V2_LPS_MIN_B 64     // full batches below this take the merged tail launch of the active-set calls (last position per image
                                // pair from L2 + repairs): the LDS-shared form puts eight images on a workgroup, i.e. B / 8 x Cout / 32
                                // workgroups -- at R/main.py's own B = 16 that is 2 x G serial chains (12.3 us per launch against 7.4)
#endif
#ifndef SPK_V2_LP_SPLIT_MAX
#define SPK_V2_LP_SPLIT_MAX 512    // last-position part: up to this many units the four waves of a workgroup split a unit's K chunks
                                   // (1024 until round 4; same box, dense / elimination / lists, ms per 100-step sample at B = 256:
                                   //  1024: 88.0 / 43.5 / 34.2, 512: 87.4 / 43.3 / 34.1, 256 and 0: 87.4 / 44.3 / 35.2)
#endif
#ifndef SPK_V2_AHEAD
#define SPK_V2_AHEAD 1          // round 5: the chunk barrier four steps before the end of the chunk, the next chunk's first fragments read behind it
#endif
#ifndef SPK_V2_NBUF
#define SPK_V2_NBUF 2           // slab buffers of the AHEAD form: 2 = copies issued one chunk ahead; 3 = two chunks ahead (147 KB of LDS, the chunk
                                // barrier waits with a counted s_waitcnt): built because the no-copy ablation runs den.conv4 / conv5 5 / 10 %
                                // faster, and measured 1 % SLOWER (same box: conv4 376 against 373 us, dense 91.3 against 90.3 ms;
                                // profiles/r5_ab_kernel_variants.txt (4), (5)) -- what the copies cost is not the wait for their landing
#endif
#ifndef SPK_V2_PF
#define SPK_V2_PF 6             // A fragments requested this many steps ahead of the MFMA that consumes them
#endif
// accumulator tiles (index 3 * i + j) that live in AGPRs: 16 (256 registers) with one wave per SIMD; all of them with two

// Certification.  The approximate path (the leading digits, fp32 recombination, folded constants) and the exact path (six
// digits, fp64 recombination, the reference's BN / LIF operations) run the same LIF recursion on pre-activations that differ by
//   |z~_t - z_t| <= c_t + 2 eps |z_t|
//   four digits:  c_t = |a| * 528 * 2^-s * n_t + 2 eps (|b| + |Bc|),   n_t = active inputs of the row at step t (counted),
//                 |32 d4 + d5| <= 528 per input;
//   five digits:  c_t = cE = |a| E5 + 2 eps (|b| + |Bc|),  E5 = 16 * 9 * Cin * 2^-s  (the dropped digit, every input active
//                 with the largest residue)
// with eps = 2^-22 (each of the handful of fp32 roundings on either path is <= 2^-24 relative to a quantity bounded by
// |z|, |b| or |Bc|).  One LIF step h = v + (z - v) / 2 halves the carried difference and adds its own roundings:
//   dh_t <= dh_{t-1} / 2 + c_t / 2 + 2 eps (|z_t| + |v_{t-1}|)
// as long as the spike decisions agreed so far (after a spike both paths restart from v = 0; the bound is simply kept).
// Five digits: the epilogue carries D_t = 2 dh_t (a factor 2 to spare) per neuron and flags it when |h_t - 1| <= D_t for some t.
// Four digits: the closed form of the same recursion, dh_t <= max c + 8 eps max |z| (|v| <= max |z|), WITHOUT the spare factor --
// the digit term of c_t is exact (528 is the largest residue there is), eps already holds every rounding 2.5-4 times over, and
// the number of flagged neurons (the repair launch: 7 % of a dense reverse step) is proportional to the bound.  Every
// unflagged neuron provably emits the exact path's spikes; flagged ones are recomputed exactly.
__device__ __forceinline__ float cert_const(float bias_f, float bna, float bnb, float Bc, float scale_f, int Cin) {
  const float E5 = 16.0f * 9.0f * (float)Cin * scale_f;
  return fabsf(bna) * E5 + 2.0f * 2.38418579e-07f * (fabsf(bnb) + fabsf(Bc)) + 1e-30f;
}
constexpr float CERT_4EPS = 4.0f * 2.38418579e-07f;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The K loop must be straight-line
// code with constant accumulator indices (a runtime index would send the accumulators through scratch); this does not
// depend on the unroller's size heuristics.
template <typename F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ unsigned spread8(unsigned x) {          // bit k -> nibble k, as the e2m1 code of 1.0 (0x2)
  x = (x | (x << 12)) & 0x000f000fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}

// Store the spikes of one 32-row tile: every lane holds the 16 step bits of its neuron (channel = lane & 31 of the group,
// position = its lane half); a 16x16 bit transpose per 16-lane row gives lane t the 16 channel bits of step t = 8 bytes of
// the (position, t) record.
__device__ __forceinline__ void store_tile_spikes(uint8_t* out, uint8_t* out_cnt, unsigned mybits, int lane, long long rec_base,
                                                  long long cnt_base, bool ok) {
  const unsigned bitsv = spk_transpose16_rows(mybits, lane);
  if (out_cnt && ok) out_cnt[cnt_base + (lane & 31)] = (uint8_t)__popc(mybits);
  if (ok) {
    uint2 o;
    o.x = spread8(bitsv & 0xffu);
    o.y = spread8((bitsv >> 8) & 0xffu);
    *reinterpret_cast<uint2*>(out + rec_base + (lane & 15) * 16 + 8 * ((lane >> 4) & 1)) = o;
  }
}

// The same store from the sixteen step masks of the scan (mk[t]: bit L = lane L's neuron spiked at step t; lanes 0..31 = the
// tile's first position, 32..63 its second): lane (16 j + t) stores the 16-channel piece j of step t = bits [16 j, 16 j + 16)
// of mk[t].  v_writelane puts the low word of mk[t] into lane t and the high word into lane 32 + t, v_permlane16_swap copies rows
// 0 / 2 of the register into rows 1 / 3, and every lane picks its half-word.
__device__ __forceinline__ void store_tile_masks(uint8_t* out, const unsigned long long (&mk)[16], int lane, long long rec_base,
                                                 bool ok) {
  unsigned x = 0;
  // (this clang has no writelane builtin; the masks are wave-uniform values hipcc keeps in SGPRs, the lane index is a constant)
#define SPK_WRITELANE(X, VAL, LANE) asm("v_writelane_b32 %0, %1, %2" : "+v"(X) : "s"(VAL), "i"(LANE))
#define SPK_WL_STEP(T)                                                          \
  do {                                                                          \
    const unsigned lo_ = (unsigned)mk[T], hi_ = (unsigned)(mk[T] >> 32);        \
    SPK_WRITELANE(x, lo_, T);                                                   \
    SPK_WRITELANE(x, hi_, 32 + T);                                              \
  } while (0)
  SPK_WL_STEP(0); SPK_WL_STEP(1); SPK_WL_STEP(2); SPK_WL_STEP(3); SPK_WL_STEP(4); SPK_WL_STEP(5); SPK_WL_STEP(6); SPK_WL_STEP(7);
  SPK_WL_STEP(8); SPK_WL_STEP(9); SPK_WL_STEP(10); SPK_WL_STEP(11); SPK_WL_STEP(12); SPK_WL_STEP(13); SPK_WL_STEP(14); SPK_WL_STEP(15);
#undef SPK_WL_STEP
#undef SPK_WRITELANE
  typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
  const v2u_ sw = __builtin_amdgcn_permlane16_swap(x, x, false, false);       // rows 1 / 3 <- rows 0 / 2
  const unsigned bitsv = sw[0] >> (16u * ((unsigned)(lane >> 4) & 1u));
  if (ok) {
    uint2 o;
    o.x = spread8(bitsv & 0xffu);
    o.y = spread8((bitsv >> 8) & 0xffu);
    *reinterpret_cast<uint2*>(out + rec_base + (lane & 15) * 16 + 8 * ((lane >> 4) & 1)) = o;
  }
}

// NWV = waves per workgroup: 4 (one per SIMD, 6 row tiles each at 7x7, 512 registers) or 8 (two per SIMD, 3 row tiles
// each, 256 registers: the partner wave's MFMAs run under this wave's copy issue, fragment waits and epilogue, and two
// waves scanning at once get the SIMD's full vector rate).  Same item, LDS plan and DMA volume either way.
// SPLIT (latents too large for one item, 8x8): an item is one of the two ROW BANDS of an image -- H / 2 output rows, H / 2 + 1
// input rows (one halo row from the other band).  Both bands sit in LDS rows 1 .. H/2 + 1 of a padded image whose rows 0 and
// H/2 + 2 stay zero; the top band's outputs are centred on LDS rows 1.., the bottom band's on rows 2.. (one row offset added
// to the fragment addresses per item).  An even position count needs no last-position launch.
// NTP > 0 (7x7, one wave per SIMD): the row tiles of an item are the positions LISTED for the image (a.need: what the
// sampler will read at this reverse step, dilated by the layers in between) instead of all 48 -- tile k = list entries 2k,
// 2k + 1, round-robin over the waves, NTP = ceil(entries / 8) tiles per wave; `slots` are the image slots with that tile
// count.  Positions outside the list keep whatever the output buffer held: nothing downstream of a listed position reads them.
// Small batches (round 6; R/main.py's own call is B = 16): an image x 32 output channels is ONE item, so a layer with B x Cout / 32
// items below the CU count leaves CUs idle (B = 16: conv2 64, conv3 / conv5 128 items on 256 CUs).  HALF = the listed-positions
// form (NTP = 3 tiles per wave of four) on two FIXED lists per image -- positions 0..23 and 24..47 -- i.e. two items per image, one
// wave per SIMD.  Same arithmetic per position as every other form (bit-equal: test_fp6v2_small_batch_split_...).
__device__ const uint8_t V2_HALF_REC[2][64] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
     23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 23, 24},
    {24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47,
     47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 47, 24}};

template <int H, int W, int NWV, bool SPLIT, int NTP, bool HALF = false>
__device__ __forceinline__ void fp6v2_body(const V2Args& a, const int g, const int il, const int lanes, const int n_images,
                                           const int* __restrict__ slots) {
  constexpr bool PRUNE = NTP > 0;
  static_assert(!HALF || (PRUNE && !SPLIT && H == 7 && W == 7 && NWV == 4 && NTP == 3), "half-image items: 7x7, four waves x three tiles");
  // how the four-digit form counts the active inputs of a row: REC: once per input record and chunk for the whole workgroup
  // (full items: the per-fragment popcounts were 29 % of the launch's vector issue), else per A fragment in the K loop
  // (listed positions: an item has few fragments, and the per-item passes of REC cost more than they save there: -2.5 %)
  constexpr bool REC = !USE_D4 && !PRUNE && !(SPK_V2_DBG & 256);     // (DBG 256: no counting at all -- timing only, wrong flags)
  // AH (round 5): the chunk barrier sits four steps BEFORE the end of the chunk instead of at its start.  Every read of the
  // current buffers has been issued by then (the last spike fragment at step NSTEP - 5, the last weight tile at the first step of
  // tap 7), so that barrier both publishes the next chunk's copies and releases the current buffers -- and the four steps behind
  // it read the next chunk's first fragments: no LDS round trip in front of a chunk's first MFMA any more.  The first chunk of an
  // item still starts with its reads (the fragment registers must not live through the epilogue).
  // SPK_V2_NBUF = 3 (an experiment, off): three slab buffers, the copies of chunk c + 2 issued during chunk c, the barrier waiting
  // with s_waitcnt vmcnt(pieces of this chunk) -- copies complete in issue order; the first chunk of an item drains the counter
  // (the epilogue's stores share it and may retire out of order).  1 % slower than two buffers, see the knob.
  constexpr int HW = H * W, PW = W + 1;
  constexpr int Hb = SPLIT ? H / 2 : H;                // output rows of an item
  constexpr int Hin = SPLIT ? Hb + 1 : H;              // input rows staged per item
  constexpr int HWb = Hb * W;                          // output positions of an item
  static_assert(SPLIT ? ((H % 2) == 0 && (HWb % (2 * NWV)) == 0)
                      : ((HW & 1) == 1 && ((HW / 2) % NWV) == 0), "whole 32-row tiles on every wave (+ one odd position)");
  static_assert(!PRUNE || (!SPLIT && (NWV == 4 || NWV == 8) && NTP <= (HWb / 2) / NWV), "position lists: 7x7 items");
  constexpr int NT = PRUNE ? NTP : (HWb / 2) / NWV;    // row tiles per wave (7x7: 6 or 3; 8x8 bands: 4)
  constexpr int N_AGPR = NWV == 4 ? (NACC * NT < 16 ? NACC * NT : 16) : NWV == 12 ? SPK_V2_AGPR12 : (NACC * NT < SPK_V2_NAGPR8 ? NACC * NT : SPK_V2_NAGPR8);   // (two waves per SIMD: with any accumulator in AGPRs hipcc splits 256 registers 128 / 128)
  constexpr int NPP = (Hin + 2) * PW + 1;              // cells of the zero-bordered LDS image (pitch W + 1: the zero
  constexpr int A_BYTES = NPP * POSB;                  //  column is shared by x = -1 of a row and x = W of the previous)
  constexpr int PPR = (W + 3) / 4;                     // DMA pieces per image row (4 positions per KiB piece)
  constexpr int NA = Hin * PPR;
  constexpr int NPA = (NA + NWV - 1) / NWV;            // A pieces per wave
  constexpr int NPW = (W_PIECES + NWV - 1) / NWV;      // W pieces per wave
  constexpr int NS_PAIR = 9 * NT, NSTEP = NS_PAIR + (USE_D4 ? N_D4 * NT : 0);
  static_assert(NWV >= 8 || NACC * NT <= 16 || (NT - 1) * NACC <= N_AGPR + 2, "only the last tile may straddle the register files");
  // AH (round 5, two waves per SIMD on full items): chunk barrier four steps before the chunk's end + THREE slab buffers, see below
  constexpr bool AH = SPK_V2_AHEAD && NWV == 8 && !PRUNE && !USE_D4;
  constexpr int NBUF = AH ? SPK_V2_NBUF : 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* const sA = lds;
  uint8_t* const sW = lds + NBUF * A_BYTES;
  // four-digit form: active inputs per input cell and step (s_cin, borders stay zero) and per output position and step (s_row)
  int* const s_cin = reinterpret_cast<int*>(lds + NBUF * A_BYTES + NBUF * W_LDS);    // [NPP][16]
  int* const s_row = s_cin + NPP * 16;                                               // [HWb + 1][16]
  int* const s_nmax = s_row + (HWb + 1) * 16;                                        // [HWb + 1]: max over the steps of s_row[p][.]
  const unsigned sA_addr = spk_lds_addr(sA), sW_addr = sA_addr + NBUF * A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = a.nch;
  const int G = a.Cout >> 5;
  // zero the A images once: the borders stay zero for the whole kernel, interiors are overwritten by DMA
  for (int i = tid; i < NBUF * A_BYTES / 16; i += NWV * 64) reinterpret_cast<uint4*>(sA)[i] = make_uint4(0, 0, 0, 0);
  if (REC)
    for (int i = tid; i < NPP * 16; i += NWV * 64) s_cin[i] = 0;
  __syncthreads();         // no wave's first DMA piece may land in a cell another wave has yet to zero

  // per-lane LDS byte offsets of this wave's A fragments (tile ti = wave + NWV * i), relative to tap (0, 0).
  // a_off: the same cell for both K halves (digit-pair instructions); a_off1 / a_off2: K half 1 one cell / PW - 2 cells
  // further (fifth-digit instructions pair the taps (0,1) (4,5) (6,7) / (2,3))
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  int a_off[NT], a_off1[NT], a_off2[NT];
  int p_out[NT];                                          // output position of this lane's accumulator rows
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int p = 2 * (wave + NWV * i) + hsel;
    a_off[i] = ((p / W) * PW + (p % W)) * POSB + tt * 16;
    a_off1[i] = a_off[i] + half * POSB;
    a_off2[i] = a_off[i] + half * (PW - 2) * POSB;
    p_out[i] = 2 * (wave + NWV * i) + half;
  }

  // DMA piece table (wave-uniform): bits 0..13 source byte offset in the slab, 14..28 LDS byte offset in the image,
  // 29..30 positions in the piece - 1.  Pieces beyond the slab repeat the last one so that the K loop issues unconditionally.
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  unsigned pa_pk[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    int id = wave_s * NPA + j;
    id = id < NA ? id : NA - 1;
    const int y = id / PPR, px = id - y * PPR;
    const int np = (W - 4 * px) < 4 ? (W - 4 * px) : 4;
    const unsigned src = (unsigned)((y * W + 4 * px) * POSB), dst = (unsigned)(((y + 1) * PW + 1 + 4 * px) * POSB);
    pa_pk[j] = src | (dst << 14) | ((unsigned)(np - 1) << 29);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned wave_k = (unsigned)wave_s * 1024u;
  auto issue_piece = [&](int q, const uint8_t* aslab, const uint8_t* wslab, unsigned dA, unsigned dW) {
    if (q < NPA) {
      const unsigned pk = pa_pk[q];
      const unsigned np = ((pk >> 29) & 3u) + 1u;
      const unsigned long long mask = np == 4 ? ~0ull : ((1ull << (16 * np)) - 1ull);
      spk_dma16s_masked(aslab + (pk & 0x3fffu), lane16, dA + ((pk >> 14) & 0x7fffu), mask);
    } else {
      unsigned ko = wave_k + 1024u * NWV * (unsigned)(q - NPA);
      if (NWV * (q - NPA) + NWV - 1 >= W_PIECES) ko = ko < (unsigned)W_PIECES * 1024u ? ko : ko - 1024u * NWV;
      spk_dma16s(wslab + ko, lane16, dW + ko);
    }
  };
  const uint8_t* const wbase = a.wq + (long long)g * nch * W_SLAB;
  // item index -> (image, band); the slab of a band starts (H/2 - 1) rows into the image for the bottom band
  auto aslab_of = [&](int itm, int c) -> const uint8_t* {
    const int b = HALF ? itm >> 1 : (PRUNE ? slots[itm] : (SPLIT ? itm >> 1 : itm)), band = SPLIT ? itm & 1 : 0;
    return a.in0 + ((long long)b * nch + c) * HW * POSB + band * (Hb - 1) * W * POSB;
  };
  const int nitems = (SPLIT || HALF) ? 2 * n_images : n_images;

  // per-channel constants (the group is fixed: loaded once)
  const int co = g * 32 + (lane & 31);
  const float scale_f = (float)a.scale[co], bias_f = (float)a.bias[co];
  const float bna = a.bn_a[co], bnb = a.bn_b[co];
  const float Ac = 32.0f * scale_f * bna;                 // z = fma(Q5, Ac, Bc),  Q5 = P01 * 2^15 + P23 * 2^5 + P4
  const float Bc = fmaf(bias_f, bna, bnb);
  // certification constants: five-digit form: cE covers the dropped sixth digit with every input active (cert_const); four-digit
  // form: cE holds the rounding terms only and cT the dropped digits PER ACTIVE INPUT (|32 d4 + d5| <= 528 units of 2^-s each,
  // rounded up), multiplied in the epilogue by the number of active inputs of the row -- 5-7x tighter than "every input
  // active" at the firing rates of the denoiser, which is what lets the fifth digit leave the matrix cores
  const float cE = USE_D4 ? cert_const(bias_f, bna, bnb, Bc, scale_f, a.Cin)
                          : 2.0f * 2.38418579e-07f * (fabsf(bnb) + fabsf(Bc)) + 1e-30f;
  const float cT = 528.0f * scale_f * fabsf(bna) * 1.000001f;
  const float Ac4 = 1024.0f * scale_f * bna;               // (four-digit form)

  const int sc_a = 0x7f7f7f7f;                            // e8m0 block scales: spikes x 1
  const int sc_p = half ? (int)0x82828282u : (int)0x87878787u;   // digit pairs: even digit (K half 0) x 2^8, odd digit x 2^3
  const int sc_4 = (int)0x82828282u;                      // fifth digit x 2^3 (e2m3 value d / 8 -> d)

  unsigned long long dbg_c0 = 0, dbg_r0 = 0;
  if (SPK_V2_DBG & 128) { dbg_c0 = __builtin_amdgcn_s_memtime(); dbg_r0 = __builtin_amdgcn_s_memrealtime(); }
  int it = 0;                                             // running chunk counter: LDS buffer = it & 1
  int bi = 0;                                             // (NBUF 3: the current chunk's buffer, it mod 3)
  if (il < nitems) {
    const uint8_t* as0 = aslab_of(il, 0);
#pragma unroll
    for (int q = 0; q < NPA + NPW; ++q) issue_piece(q, as0, wbase, sA_addr, sW_addr);
    if constexpr (AH) {                                   // (the only chunks whose copies no chunk barrier has waited for)
      if constexpr (NBUF == 3) {
        // the second chunk of the stream as well: chunk 1 of this item, or chunk 0 of the next
        const bool two = nch > 1 || il + lanes < nitems;
        const uint8_t* as1 = nch > 1 ? aslab_of(il, 1) : aslab_of(two ? il + lanes : il, 0);
        const uint8_t* ws1 = wbase + (long long)(nch > 1 ? 1 : 0) * W_SLAB;
#pragma unroll
        for (int q = 0; q < NPA + NPW; ++q) issue_piece(q, as1, ws1, sA_addr + A_BYTES, sW_addr + W_LDS);
      }
      spk_dma_wait_all();
      __syncthreads();
    }
  }
  for (int itm = il; itm < nitems; itm += lanes) {
    const int b = HALF ? itm >> 1 : (PRUNE ? slots[itm] : (SPLIT ? itm >> 1 : itm)), band = SPLIT ? itm & 1 : 0;
    const int band_off = band * PW * POSB;                // bottom band: fragment addresses one LDS row further down
    int n_list = 2 * NT * NWV;
    if constexpr (PRUNE) {
      const uint8_t* rec = HALF ? &V2_HALF_REC[itm & 1][0] : a.need + (long long)b * 64;
      n_list = __builtin_amdgcn_readfirstlane((int)rec[48]);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int k2 = 2 * (wave + NWV * i);
        const int p = rec[k2 + hsel];
        a_off[i] = ((p / W) * PW + (p % W)) * POSB + tt * 16;
        a_off1[i] = a_off[i] + half * POSB;
        a_off2[i] = a_off[i] + half * (PW - 2) * POSB;
        p_out[i] = rec[k2 + half];
      }
    }
    v16f acc[NT][NACC];   // [i][0]: pair 01, [i][1]: pair 23, ([i][2]: fifth digit;) written (not accumulated) by the first MFMA
    int cnt[NT];          // !REC: active inputs of this lane's A row (position, step) over all taps and chunks
#pragma unroll
    for (int i = 0; i < NT; ++i) cnt[i] = 0;
    // REC: every thread counts the active inputs of NR input records (cell, step) of the item, chunk by chunk
    constexpr int NREC = Hin * W * 16, NR = (NREC + NWV * 64 - 1) / (NWV * 64);
    int creg[NR];
    int rec_off[NR];                                      // LDS byte offset of the record inside a slab; threads without one read
    bool rec_ok[NR];                                      //  cell 0, a border cell (always zero): an unconditional read (a conditional
                                                          //  one is a branch with an LDS wait of its own, twice per chunk)
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      creg[k] = 0;
      const int r = tid + k * NWV * 64;
      const int cl = r >> 4, t = r & 15;
      rec_ok[k] = r < NREC;
      rec_off[k] = rec_ok[k] ? (((cl / W) + 1) * PW + 1 + (cl % W)) * POSB + t * 16 : 0;
    }
    constexpr int PFX = NWV == 4 ? SPK_V2_PF : 4;
    v6i bp[2][2];                                         // digit-pair tiles of tap parity [tap & 1][pair]   (AH: carried over the
    v4i af[PFX];                                          //  chunks of an item: a chunk's last steps fill them for the next one)
    for (int c = 0; c < nch; ++c, ++it) {
      const int buf = NBUF == 3 ? bi : (it & 1);
      const int buf1 = NBUF == 3 ? (bi == 2 ? 0 : bi + 1) : (buf ^ 1);          // the next chunk's buffers
      const int buf2 = NBUF == 3 ? (bi == 0 ? 2 : bi - 1) : (buf ^ 1);          // where this chunk's copies go (NBUF 3: chunk + 2)
      if constexpr (NBUF == 3) bi = buf1;
      if constexpr (!AH) {
        spk_dma_wait_all();  // this wave's share of the chunk's DMA has landed ...
        __syncthreads();     // ... and so has everyone else's; everyone is done with the other buffer
      }
      int nb = itm, nc = c + 1;
      if (nc == nch) { nc = 0; nb = itm + lanes; }
      if constexpr (NBUF == 3) {                          // (the chunk after that)
        nc = nc + 1;
        if (nc == nch) { nc = 0; nb = nb + lanes; }
      }
      const bool have_next = nb < nitems;                 // otherwise the last chunk is copied once more (never read)
      const uint8_t* n_aslab = aslab_of(have_next ? nb : itm, have_next ? nc : c);
      const uint8_t* n_wslab = wbase + (long long)(have_next ? nc : c) * W_SLAB;
      const unsigned n_dA = sA_addr + buf2 * A_BYTES;
      const unsigned n_dW = sW_addr + buf2 * W_LDS;

      // REC: the chunk's record counts.  SPK_V2_REC_INLOOP: the two LDS reads are issued after the chunk's first MFMA step and
      // their popcounts a few steps later, inside the MFMA stream (in front of it they held the chunk's first MFMA back by an
      // LDS round trip at every chunk barrier).
      if constexpr (REC && !SPK_V2_REC_INLOOP) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
          const v4i rv = *reinterpret_cast<const v4i*>(sA + buf * A_BYTES + rec_off[k]);
          creg[k] += __builtin_popcount((unsigned)rv[0]) + __builtin_popcount((unsigned)rv[1]) +
                     __builtin_popcount((unsigned)rv[2]) + __builtin_popcount((unsigned)rv[3]);
        }
      }
      v4i rvq[NR];
      auto compute = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint8_t* A = sA + buf * A_BYTES + band_off;
        const uint8_t* Wb = sW + buf * W_LDS;
        auto toff = [](int tap) constexpr -> int { return ((tap / 3) * PW + (tap % 3)) * POSB; };
        // Step order: blocks of NT steps (one per row tile) -- tap 0, tap 1, D(0), tap 2, tap 3, D(1), ..., tap 8, D(4), where a
        // tap block issues the two digit-pair MFMAs per step and D(q) the single fifth-digit MFMA of taps (2q, 2q + 1).  The
        // one-MFMA blocks sit BETWEEN two-MFMA blocks (six in a row left the fragment prefetch only ~130 cycles ahead).
        // (four-digit form, USE_D4 == false: the nine tap blocks only)
        auto blk_is_d = [](int blk) constexpr -> bool { return USE_D4 && (blk < 12 ? (blk % 3 == 2) : (blk == 13)); };
        auto blk_tap = [](int blk) constexpr -> int { return !USE_D4 ? blk : (blk < 12 ? 2 * (blk / 3) + (blk % 3) : 8); };   // tap blocks
        auto blk_q = [](int blk) constexpr -> int { return blk < 12 ? blk / 3 : 4; };                          // D blocks
        constexpr int NBLK = USE_D4 ? 14 : 9;
        static_assert(NSTEP == NBLK * NT, "blocks of NT steps");
        auto lda = [&](auto s_tag) -> v4i {
          constexpr int s = decltype(s_tag)::value;
          constexpr int blk = s / NT, i = s % NT;
          if constexpr (!blk_is_d(blk)) {
            return *reinterpret_cast<const v4i*>(A + a_off[i] + toff(blk_tap(blk)));
          } else {
            constexpr int q = blk_q(blk);
            const int base = q == 1 ? a_off2[i] : (q == 4 ? a_off[i] : a_off1[i]);
            return *reinterpret_cast<const v4i*>(A + base + toff(2 * q));
          }
        };
        auto ldb = [&](int tile) -> v6i {
          // 16 + 8 bytes per lane; the 8-byte read is volatile so that hipcc does not pair the tails of two tiles
          const uint8_t* p = Wb + tile * WT;
          const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
          typedef const volatile __attribute__((address_space(3))) v2i* lds_v2i_ptr;
          const v2i y = *(lds_v2i_ptr)SPK_LDS(p + 1024 + lane * 8);
          const v6i r = {x[0], x[1], x[2], x[3], y[0], y[1]};
          return r;
        };
        v6i b4[2];                                        // fifth-digit tiles [q & 1]
        constexpr int PF = PFX;
        if constexpr (!AH || FIRST) {
          bp[0][0] = ldb(0); bp[0][1] = ldb(1);
          static_for<PF>([&](auto s_tag) { af[decltype(s_tag)::value] = lda(s_tag); });
        }
        // AH: the next chunk's buffers (this item's; the last chunk of an item reads nothing ahead)
        const uint8_t* An = sA + buf1 * A_BYTES + band_off;
        const uint8_t* Wn = sW + buf1 * W_LDS;
        const bool ahead = AH && c + 1 < nch;
        static_for<NSTEP>([&](auto s_tag) {
          constexpr int s = decltype(s_tag)::value;
          constexpr int blk = s / NT, i = s % NT;
          if constexpr (AH && s == NSTEP - PF) {
            // every read of this chunk's buffers is issued; this wave's copies of the next chunk have landed: the chunk barrier
            if constexpr (NBUF == 3 && !FIRST) {
              static_assert(NPA + NPW <= 15, "vmcnt immediate");
              asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPA + NPW) : "memory");     // (this chunk's pieces may still fly)
            } else if constexpr (SPK_V2_DBG & 512) {            // (timing only, results wrong: no chunk barrier -- what does it cost?)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
          }
          const v4i av = af[s % PF];
          if constexpr (s + PF < NSTEP) af[s % PF] = lda(std::integral_constant<int, s + PF>{});
          else if constexpr (AH) {
            // the slot just consumed takes the next chunk's step with the same slot number: steps NSTEP - 4 .. NSTEP - 1 free slots
            // (NSTEP - 4) % 4 ..., i.e. the next chunk's steps s' with s' % 4 == s % 4
            constexpr int k = s % PF;                         // (PF == 4: next step k lives in slot k)
            if (ahead) af[k] = *reinterpret_cast<const v4i*>(An + a_off[k % NT] + toff(blk_tap(k / NT)));
          }
          constexpr int NPIECES = NPA + NPW;
          // two copy slots per block (its first step and its middle step): 18 slots for the 11 pieces of the four-digit form
#define V2_DMA_SLOT()                                                                              \
  do {                                                                                             \
    if constexpr (i == 0 && 2 * blk < NPIECES) {                                                   \
      if (!(SPK_V2_DBG & 1)) issue_piece(2 * blk, n_aslab, n_wslab, n_dA, n_dW);                   \
    }                                                                                              \
    if constexpr (i == (NT > 1 ? NT / 2 : 0) && 2 * blk + 1 < NPIECES) {                           \
      if (!(SPK_V2_DBG & 1)) issue_piece(2 * blk + 1, n_aslab, n_wslab, n_dA, n_dW);               \
    }                                                                                              \
  } while (0)
          // the next block's weight tiles are requested at the first step of this block
#define V2_NEXT_TILES()                                                                            \
  do {                                                                                             \
    if constexpr (i == 0 && blk + 1 < NBLK) {                                                      \
      if constexpr (blk_is_d(blk + 1)) b4[blk_q(blk + 1) & 1] = ldb(N_PAIR + blk_q(blk + 1));      \
      else {                                                                                       \
        bp[blk_tap(blk + 1) & 1][0] = ldb(2 * blk_tap(blk + 1));                                   \
        bp[blk_tap(blk + 1) & 1][1] = ldb(2 * blk_tap(blk + 1) + 1);                               \
      }                                                                                            \
    }                                                                                              \
  } while (0)
          if constexpr (!blk_is_d(blk)) {
            constexpr int tap = blk_tap(blk);
            if constexpr (!USE_D4 && !REC && !(SPK_V2_DBG & 256)) {
              // (volatile: left to itself hipcc defers the pure popcounts and keeps every fragment of the chunk alive)
              asm volatile("v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %0, %2, %0\n\tv_bcnt_u32_b32 %0, %3, %0\n\t"
                           "v_bcnt_u32_b32 %0, %4, %0" : "+v"(cnt[i]) : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]));
            }
#define V2_PAIR_MFMA(J)                                                                                      \
  do {                                                                                                        \
    if constexpr (FIRST && tap == 0) {                                                                        \
      if constexpr (NACC * i + (J) < N_AGPR) SPK_MFMA2_Z("a", acc[i][J], av, bp[0][J], sc_a, sc_p);              \
      else SPK_MFMA2_Z("v", acc[i][J], av, bp[0][J], sc_a, sc_p);                                             \
    } else {                                                                                                  \
      if constexpr (NACC * i + (J) < N_AGPR) SPK_MFMA2("a", acc[i][J], av, bp[tap & 1][J], sc_a, sc_p);          \
      else SPK_MFMA2("v", acc[i][J], av, bp[tap & 1][J], sc_a, sc_p);                                         \
    }                                                                                                         \
  } while (0)
            V2_PAIR_MFMA(0);
            __builtin_amdgcn_sched_barrier(0);
            V2_DMA_SLOT();
            V2_NEXT_TILES();
            if constexpr (REC && SPK_V2_REC_INLOOP) {
              if constexpr (s == 1) {
#pragma unroll
                for (int k = 0; k < NR; ++k) rvq[k] = *reinterpret_cast<const v4i*>(sA + buf * A_BYTES + rec_off[k]);
              }
              if constexpr (s == SPK_V2_REC_STEP) {
#pragma unroll
                for (int k = 0; k < NR; ++k)
                  creg[k] += __builtin_popcount((unsigned)rvq[k][0]) + __builtin_popcount((unsigned)rvq[k][1]) +
                             __builtin_popcount((unsigned)rvq[k][2]) + __builtin_popcount((unsigned)rvq[k][3]);
              }
            }
            V2_PAIR_MFMA(1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AH && s == NSTEP - 1) {
              if (ahead) {
                auto ldb_n = [&](int tile) -> v6i {
                  const uint8_t* p = Wn + tile * WT;
                  const v4i x = *reinterpret_cast<const v4i*>(p + lane * 16);
                  typedef const volatile __attribute__((address_space(3))) v2i* lds_v2i_ptr;
                  const v2i y = *(lds_v2i_ptr)SPK_LDS(p + 1024 + lane * 8);
                  const v6i r = {x[0], x[1], x[2], x[3], y[0], y[1]};
                  return r;
                };
                bp[0][0] = ldb_n(0); bp[0][1] = ldb_n(1);
              }
            }
          } else {
            constexpr int q = blk_q(blk);
            if constexpr (USE_D4) {
              if constexpr (FIRST && q == 0) {
                if constexpr (NACC * i + 2 < N_AGPR) SPK_MFMA2_Z("a", acc[i][NACC - 1], av, b4[0], sc_a, sc_4);
                else SPK_MFMA2_Z("v", acc[i][NACC - 1], av, b4[0], sc_a, sc_4);
              } else {
                if constexpr (NACC * i + 2 < N_AGPR) SPK_MFMA2("a", acc[i][NACC - 1], av, b4[q & 1], sc_a, sc_4);
                else SPK_MFMA2("v", acc[i][NACC - 1], av, b4[q & 1], sc_a, sc_4);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            V2_DMA_SLOT();
            V2_NEXT_TILES();
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      };
      if (c == 0) compute(std::true_type{}); else compute(std::false_type{});
    }   // chunks

    if constexpr (REC) {
      // publish the record counts, then add the nine taps of every output position: s_row[p][t] = active inputs of row (p, t)
#pragma unroll
      for (int k = 0; k < NR; ++k)
        if (rec_ok[k]) s_cin[(rec_off[k] / POSB) * 16 + ((rec_off[k] % POSB) >> 4)] = creg[k];
      if (tid <= HWb) s_nmax[tid] = 0;
      __syncthreads();
      for (int e = tid; e < HWb * 16; e += NWV * 64) {
        const int pp = e >> 4, t = e & 15;
        const int* c0 = s_cin + (((pp / W) + band) * PW + (pp % W)) * 16 + t;
        int sum = 0;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sum += c0[(dy * PW + dx) * 16];
        s_row[e] = sum;
        // the certification's first stage only needs max_t n_t of a position: one LDS atomic here instead of four 16-byte reads
        // and twelve v_max per tile and lane in the scan (the second stage, a few percent of the tiles, reads the sixteen counts)
        if (SPK_V2_NMAX_LDS) atomicMax(&s_nmax[pp], sum);
      }
      __syncthreads();
    }
    // The MFMAs are opaque to hipcc's hazard recognizer: an accumulator may be read 18 wait states after the (16-pass)
    // MFMA that wrote it was issued.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (SPK_V2_DBG & 4) {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
          if (NACC * i + j < N_AGPR) asm volatile("" : "+a"(acc[i][j]));
          sacc += acc[i][j][0];
        }
      if (sacc == 12345.f) a.out[0] = 1;
      continue;
    }

    // ---------------- epilogue: fp32 recombination, BN, LIF scan, certification -------------------------------------
    // The tile that holds the VGPR-resident accumulators goes first (frees their registers for the scan temporaries).
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const int i = k == 0 ? NT - 1 : k - 1;
#pragma unroll
      for (int j = 0; j < NACC; ++j)
        if (NACC * i + j < N_AGPR) asm volatile("" : "+a"(acc[i][j]));
      float v = 0.f, D = 0.f;
      unsigned mybits = 0;
      unsigned long long mk[16];                          // (SPK_V2_MASKSTORE: the step masks of the four-digit scan)
      bool flg = false;
      int cntv[16];                                       // four-digit form: active inputs of this lane's rows (its position, step r)
      if constexpr (SPK_V2_DBG & 256) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cntv[r] = 0;
      } else if constexpr (!USE_D4 && !REC) {
        // the count of accumulator row r sits in the A-layout lane (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rowA = (r & 3) + 8 * (r >> 2);
          const int c0 = __builtin_amdgcn_readlane(cnt[i], rowA), c1 = __builtin_amdgcn_readlane(cnt[i], rowA + 4);
          cntv[r] = half ? c1 : c0;
        }
      }
      int nmax_rec = 0;
      if constexpr (REC && SPK_V2_NMAX_LDS) nmax_rec = s_nmax[2 * (wave + NWV * i) + half];     // (position within the item)
      if constexpr (REC && !SPK_V2_NMAX_LDS) {
        const v4i* rp = reinterpret_cast<const v4i*>(s_row + (2 * (wave + NWV * i) + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4i c4 = rp[q];
          cntv[4 * q] = c4[0]; cntv[4 * q + 1] = c4[1]; cntv[4 * q + 2] = c4[2]; cntv[4 * q + 3] = c4[3];
        }
      }
      if constexpr (USE_D4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float q5 = fmaf(fmaf(acc[i][0][r], 1024.0f, acc[i][1][r]), 32.0f, acc[i][NACC - 1][r]);
          const float z = fmaf(q5, Ac, Bc);
          D = fmaf(fabsf(z) + fabsf(v), CERT_4EPS, fmaf(D, 0.5f, cE));      // D_t = D_{t-1} / 2 + cE + 4 eps (|z| + |v|)
          const float h = v + (z - v) * 0.5f;
          const bool s = h >= 1.0f;
          flg = flg || (fabsf(h - 1.0f) <= D);
          v = s ? 0.0f : h;
          mybits |= s ? (1u << r) : 0u;
        }
      } else {
        // four digits: z = Q4 * (1024 Ac) + Bc, Q4 = P01 * 2^10 + P23 (two steps at a time on the packed fp32 pipe); the dropped
        // digits move z_t by at most c_t = cE + cT n_t (n_t active inputs of the row).  D_t = D_{t-1} / 2 + c_t + 4 eps (|z_t| +
        // |v_{t-1}|) and |v| <= max |z| give D_t <= 2 (cE + cT max_t n_t) + 16 eps max_t |z_t| for every t: track max |z|,
        // max n and min |h - 1| (three instructions per step instead of eight) and compare once, against dh = D / 2
        typedef float v2f __attribute__((ext_vector_type(2)));
        float zmax = 0.f, dmin = 3.0e38f;
        int nmax = nmax_rec;
        bool sp[16];
#pragma unroll
        for (int r2 = 0; r2 < 16; r2 += 2) {
          const v2f p0 = {acc[i][0][r2], acc[i][0][r2 + 1]}, p1 = {acc[i][1][r2], acc[i][1][r2 + 1]};
          const v2f q4 = __builtin_elementwise_fma(p0, (v2f){1024.0f, 1024.0f}, p1);
          const v2f z2 = __builtin_elementwise_fma(q4, (v2f){Ac4, Ac4}, (v2f){Bc, Bc});
          if constexpr (!REC || !SPK_V2_NMAX_LDS) nmax = max(nmax, max(cntv[r2], cntv[r2 + 1]));
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float z = z2[e];
            zmax = fmaxf(zmax, fabsf(z));
            const float h = fmaf(z - v, 0.5f, v);            // == v + (z - v) * 0.5f: the product is exact (v is not an output here)
            const float hm = h - 1.0f;
            dmin = fminf(dmin, fabsf(hm));
            const bool s = h >= 1.0f;
            v = s ? 0.0f : h;
            if constexpr (SPK_V2_MASKSTORE) sp[r2 + e] = s;
            else if constexpr (SPK_V2_SIGNBITS) mybits = __builtin_amdgcn_alignbit(mybits, __float_as_uint(hm), 31);   // (mybits << 1) | sign(h - 1)
            else mybits |= s ? (1u << (r2 + e)) : 0u;
          }
        }
        if constexpr (SPK_V2_MASKSTORE) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[r] = __builtin_amdgcn_ballot_w64(sp[r]);
          if (a.out_cnt) {                                   // (conv5 only: the per-neuron spike counts conv6 reads)
#pragma unroll
            for (int r = 0; r < 16; ++r) mybits += sp[r] ? 1u : 0u;          // here: the COUNT, not the bit word
          }
        }
        if constexpr (SPK_V2_SIGNBITS && !SPK_V2_MASKSTORE) mybits = ~(__builtin_bitreverse32(mybits) >> 16) & 0xffffu;   // bit r = NOT sign(h_r - 1)
        flg = dmin <= SPK_V2_SPARE * fmaf(zmax, 2.5f * CERT_4EPS, fmaf((float)nmax, cT, cE));   // (10 eps for 8: a little to spare)
        if (SPK_V2_STAGE2 && __builtin_amdgcn_ballot_w64(flg) != 0ull) {
          // SECOND STAGE (a wave with a flagged lane: a few percent of the tiles).  The closed form above compares the closest
          // approach of ANY step with the bound of the WORST step.  The recursion it was derived from is tighter twice over: the
          // bound of step t only carries the counts of the steps before it, halved once per step (dh_t <= dh_{t-1} / 2 + c_t / 2
          // + 2 eps (|z_t| + |v_{t-1}|), c_t = cE + cT n_t; 2.5 eps for 2 as above), and each step's |h_t - 1| is compared with
          // ITS bound.  Flagged only if both stages flag: about half the exact recomputations (the repair launch's time is
          // proportional to them).  Every lane re-scans (the branch is wave-uniform); accumulators and counts are still live.
          float v2 = 0.f, dh = 0.f;
          bool f2 = false;
          if constexpr (REC && SPK_V2_NMAX_LDS) {
            const v4i* rp = reinterpret_cast<const v4i*>(s_row + (2 * (wave + NWV * i) + half) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4i c4 = rp[q];
              cntv[4 * q] = c4[0]; cntv[4 * q + 1] = c4[1]; cntv[4 * q + 2] = c4[2]; cntv[4 * q + 3] = c4[3];
            }
          }
#pragma unroll
          for (int r2 = 0; r2 < 16; r2 += 2) {
            const v2f p0 = {acc[i][0][r2], acc[i][0][r2 + 1]}, p1 = {acc[i][1][r2], acc[i][1][r2 + 1]};
            const v2f q4 = __builtin_elementwise_fma(p0, (v2f){1024.0f, 1024.0f}, p1);
            const v2f z2 = __builtin_elementwise_fma(q4, (v2f){Ac4, Ac4}, (v2f){Bc, Bc});
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const float z = z2[e];
              const float ct = fmaf((float)cntv[r2 + e], cT, cE);
              dh = fmaf(fabsf(z) + fabsf(v2), 0.625f * CERT_4EPS, fmaf(dh, 0.5f, 0.5f * ct));
              const float h = fmaf(z - v2, 0.5f, v2);
              f2 = f2 || (fabsf(h - 1.0f) <= SPK_V2_SPARE * dh);
              v2 = h >= 1.0f ? 0.0f : h;
            }
          }
          flg = flg && f2;
        }
      }
      const int ti = wave + NWV * i;
      // accumulator lane half == position within the tile; a list that does not fill its last tiles repeats its last
      // position there: computed and dropped
      const int p = PRUNE ? p_out[i] : 2 * ti + half + band * HWb;
      const bool listed = !PRUNE || 2 * ti + half < n_list;
      if (flg && listed && !(SPK_V2_DBG & 32)) {
        const long long n = ((long long)b * a.Cout + co) * HW + p;
        const unsigned idx = atomicAdd(a.flags, 1u);
        if (idx < a.flag_cap) a.flags[2 + idx] = (unsigned)n;
        else atomicOr(a.flags + 2 + FLAG_CAP + (n >> 5), 1u << (n & 31));
      }
      const long long rec = (((long long)b * G + g) * HW + p) * POSB;
      if constexpr (SPK_V2_MASKSTORE && !USE_D4) {
        if (a.out_cnt && listed) a.out_cnt[(((long long)b * G + g) * HW + p) * 32 + (lane & 31)] = (uint8_t)mybits;
        store_tile_masks(a.out, mk, lane, rec, listed);
      } else {
        store_tile_spikes(a.out, a.out_cnt, mybits, lane, rec, (((long long)b * G + g) * HW + p) * 32, listed);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }   // images
  spk_dma_wait_all();     // the copy issued during the very last chunk must not outlive the workgroup's LDS allocation
  if ((SPK_V2_DBG & 128) && tid == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.flags + 2) + 2 * blockIdx.x;
    o[0] = __builtin_amdgcn_s_memtime() - dbg_c0;
    o[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
  }
}

// workgroup -> (channel group g, image lane il, lanes): the group is FIXED for the whole launch.  XCD-aware form:
// workgroups k and k + 8 share an XCD and its L2; XCD x owns channel-group set x % nsets (gx consecutive groups, chosen
// by the host so that their packed weights stay L2-resident) and image partition x / nsets.
__device__ __forceinline__ void fp6v2_wg_map(const V2Args& a, int& g, int& il, int& lanes) {
  const int k = blockIdx.x, G = a.Cout >> 5;
  if (a.gx > 0) {
    const int S = gridDim.x >> 3, x = k & 7, slot = k >> 3;
    const int npart = 8 / a.nsets, set = x % a.nsets, xi = x / a.nsets, lanes_x = S / a.gx;
    g = set * a.gx + slot % a.gx;
    il = xi * lanes_x + slot / a.gx;
    lanes = npart * lanes_x;
  } else {
    g = k % G;
    il = k / G;
    lanes = gridDim.x / G;
  }
}

// Hand-over to the tail launch: the workgroup that finishes last publishes the number of flagged neurons (ws[1]) and re-arms
// the live counter (ws[0]) for the next layer's main launch.  The workgroups of a main launch finish at different times, so
// this ticket costs nothing -- and the tail launch then needs no ordering between its repair part (which reads the count)
// and anything that resets it: repair and last position run as ONE launch.
__device__ __forceinline__ void fp6v2_handover(const V2Args& a) {
  if (!a.handover) return;
  // (no fence: every atomicAdd on the counter has RETURNED before its wave reaches the barrier, the count is read back with
  //  an atomic, and list entries are plain stores read by the next launch -- a __threadfence here cost 4.5 us per launch)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(a.flags + a.ticket_idx, 1u) == gridDim.x - 1) {
      a.flags[1] = atomicAdd(a.flags, 0u);
      if (!(SPK_V2_DBG & 64)) a.flags[0] = 0u;
      a.flags[a.ticket_idx] = 0u;
    }
  }
}

#if SPK_V2_VARIANTS
#include "variants/fp6v2_forms.inc"      // duo / deferred-scan / staggered forms (measured slower; not in the shipped library)
#endif


template <int H, int W, int NWV, bool SPLIT = false>
__global__ __launch_bounds__(NWV * 64, 1) void conv3x3_fp6v2_kernel(V2Args a) {
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  int g, il, lanes;
  fp6v2_wg_map(a, g, il, lanes);
  fp6v2_body<H, W, NWV, SPLIT, 0>(a, g, il, lanes, Bn, nullptr);
  fp6v2_handover(a);
}

// Small batches: two half-image items per image (fp6v2_body HALF), four waves
template <int H, int W>
__global__ __launch_bounds__(256, 1) void conv3x3_fp6v2_half_kernel(V2Args a) {
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  int g, il, lanes;
  fp6v2_wg_map(a, g, il, lanes);
  fp6v2_body<H, W, 4, false, 3, true>(a, g, il, lanes, Bn, nullptr);
  fp6v2_handover(a);
}

// Listed positions: the image lanes of every channel group are divided among the six tile-count classes in proportion to
// their work (slots x tiles, with a floor for the copy-bound small classes); a workgroup then runs the item loop
// instantiated for its class on that class's slots.  Every workgroup derives the same division from cls_cnt.
// (NWV = 8: two waves per SIMD; a class of k tiles per wave of four becomes ceil(k / 2) tiles per wave of eight)
template <int H, int W, int NWV>
__global__ __launch_bounds__(NWV * 64, 1) void conv3x3_fp6v2_listed_kernel(V2Args a) {
  constexpr int NC = (H * W / 2) / 4;                    // classes: 1 .. NC tiles per wave
  int g, il, lanes;
  fp6v2_wg_map(a, g, il, lanes);
  int cnt[NC], work[NC], u[NC];
  int total = 0, nonempty = 0;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    cnt[k] = a.cls_cnt[k];
    work[k] = cnt[k] * (k + 1 < SPK_V2_KMIN ? SPK_V2_KMIN : k + 1);
    total += work[k];
    nonempty += cnt[k] > 0;
  }
  if (total == 0) { fp6v2_handover(a); return; }
  // one lane per non-empty class, the rest in proportion, leftovers to the class with the most work per lane
  int rem = lanes - nonempty, used = 0;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    u[k] = cnt[k] > 0 ? 1 + (int)((long long)rem * work[k] / total) : 0;
    used += u[k];
  }
  for (int left = lanes - used; left > 0; --left) {
    int best = 0;
    long long bw = -1, bu = 1;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (cnt[k] > 0 && (long long)work[k] * bu > bw * u[k]) { bw = work[k]; bu = u[k]; best = k; }
#pragma unroll
    for (int k = 0; k < NC; ++k) u[k] += (k == best);
  }
  int start = 0, cls = -1, il_c = 0, lanes_c = 1;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    if (il >= start && il < start + u[k]) { cls = k; il_c = il - start; lanes_c = u[k]; }
    start += u[k];
  }
  const int* slots = a.cls_list + (long long)(cls < 0 ? 0 : cls) * a.B;
  if constexpr (NWV == 4) {
    switch (cls) {
      case 0: fp6v2_body<H, W, 4, false, 1>(a, g, il_c, lanes_c, cnt[0], slots); break;
      case 1: fp6v2_body<H, W, 4, false, 2>(a, g, il_c, lanes_c, cnt[1], slots); break;
      case 2: fp6v2_body<H, W, 4, false, 3>(a, g, il_c, lanes_c, cnt[2], slots); break;
      case 3: fp6v2_body<H, W, 4, false, 4>(a, g, il_c, lanes_c, cnt[3], slots); break;
      case 4: fp6v2_body<H, W, 4, false, 5>(a, g, il_c, lanes_c, cnt[4], slots); break;
      case 5: fp6v2_body<H, W, 4, false, 6>(a, g, il_c, lanes_c, cnt[5], slots); break;
      default: break;
    }
  } else {
    switch (cls) {
      case 0: case 1: fp6v2_body<H, W, 8, false, 1>(a, g, il_c, lanes_c, cnt[cls], slots); break;
      case 2: case 3: fp6v2_body<H, W, 8, false, 2>(a, g, il_c, lanes_c, cnt[cls], slots); break;
      case 4: case 5: fp6v2_body<H, W, 8, false, 3>(a, g, il_c, lanes_c, cnt[cls], slots); break;
      default: break;
    }
  }
  fp6v2_handover(a);
}

// ------------------------------------------------------------------------------------------------ tail launches
// (1) the last position of every image, exactly: one workgroup = two images x one channel group (a 32-row tile whose lane
//     halves are the images), its waves splitting the K chunks; the four taps that reach position (H-1, W-1) from inside
//     the image; operands straight from L2; all six digits (the two sixth-digit tiles of the slab), fp64 recombination.
// (2) the flagged neurons of the main launch, exactly (fixup_neuron / fp6v2_fixup_kernel below).
__device__ __forceinline__ float exact_preact(double s, double sc, double bi) { return (float)fma(s, sc, bi); }

// red: ONE [3][16][64] buffer (the partial sums of waves 1 .. 3 pass through it one after the other: the merged tail launch
// keeps eight workgroups on a CU) or, WIDE, three of them (one barrier: the stand-alone launch of full batches)
template <int H, int W, bool WIDE>
__device__ __forceinline__ void fp6v2_lastpos_body(const V2Args& a, const int bid, float (*red)[16][64]) {
  constexpr int HW = H * W, NP = SPK_V2_LP_PAIRS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = a.nch, G = a.Cout >> 5;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  // (without the hand-over this launch follows the repair launch on the stream and re-arms the flag counter)
  if (!a.handover && bid == 0 && threadIdx.x == 0 && !(SPK_V2_DBG & 64)) { a.flags[1] = a.flags[0]; a.flags[0] = 0u; }
  // one unit = NP 32-row tiles = NP image pairs (the lane halves of a tile are the two images) x one channel group: the
  // weight tiles of a chunk are read ONCE for all of them (with one pair per unit the launch moved 327 MB of weight tiles
  // through L2 for the 256 -> 512 layer at B = 256: bound by that).  The four waves of a workgroup split the K chunks
  // (wave w: chunks w, w + 4, ...) and add their partial sums -- exact integers below 2^24 in fp32, so the order of the
  // additions does not matter -- in LDS: few units are bound by the latency of a wave's chain of dependent gathers
  // (with many units the launch is bound by throughput instead: then every wave takes a unit of its own)
  const int n_units = ((Bn + 2 * NP - 1) / (2 * NP)) * G;
  const bool split = n_units <= SPK_V2_LP_SPLIT_MAX;        // (uniform over the launch)
  const int unit = split ? bid : bid * 4 + wave;
  const int g = unit % G, b0 = (unit / G) * 2 * NP;
  if (b0 >= Bn) return;
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  const int sc_a = 0x7f7f7f7f;
  const int sc_p = half ? (int)0x82828282u : (int)0x87878787u;
  const int sc_hi = (int)0x87878787u, sc_lo = (int)0x82828282u;
  v16f acc[NP][3];
#pragma unroll
  for (int q = 0; q < NP; ++q)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][j][r] = 0.f;
  const int py = H - 1, px = W - 1;
  for (int c = split ? wave : 0; c < nch; c += split ? 4 : 1) {
    const uint8_t* wslab = a.wq + ((long long)g * nch + c) * W_SLAB;
    auto ldb = [&](int tile) -> v8i {
      const uint8_t* wt = wslab + tile * WT;
      const v4i bx = *reinterpret_cast<const v4i*>(wt + lane * 16);
      const v2i by = *reinterpret_cast<const v2i*>(wt + 1024 + lane * 8);
      return v8i{bx[0], bx[1], bx[2], bx[3], by[0], by[1], 0, 0};
    };
    // spikes of the four contributing taps (0, 1, 3, 4)
    v4i sp[NP][4];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int bA = b0 + 2 * q + hsel;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tap = (u >> 1) * 3 + (u & 1);
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        sp[q][u] = v4i{0, 0, 0, 0};
        if (bA < Bn)
          sp[q][u] = *reinterpret_cast<const v4i*>(a.in0 + (((long long)bA * nch + c) * HW + yy * W + xx) * POSB + tt * 16);
      }
    }
    // all thirteen weight tiles of the chunk are requested before the first MFMA waits
    v8i bt[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tap = (u >> 1) * 3 + (u & 1);
      bt[2 * u] = ldb(2 * tap); bt[2 * u + 1] = ldb(2 * tap + 1);
    }
    const v8i t18 = ldb(N_PAIR), t19 = ldb(N_PAIR + 1), t20 = ldb(N_PAIR + 2), t23 = ldb(N_MAIN), t24 = ldb(N_MAIN + 1);
    auto mm = [&](v16f& d, const v4i& av, const v8i& bv, int sb) {
      const v8i a8 = {av[0], av[1], av[2], av[3], 0, 0, 0, 0};
      d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, bv, d, 4, 2, 0, sc_a, 0, sb);
    };
    const v4i zero = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { mm(acc[q][0], sp[q][u], bt[2 * u], sc_p); mm(acc[q][1], sp[q][u], bt[2 * u + 1], sc_p); }   // digit pairs 01, 23
      // fifth digit (x 32) and sixth digit into acc[2] = 32 * D4 + D5: tiles 18 (taps 0|1), 19 (2|3), 20 (4|5) and the
      // sixth-digit tiles 23 (taps 0|1), 24 (taps 3|4); taps 2 and 5 lie outside the image: zero spikes
      const v4i a01 = half ? sp[q][1] : sp[q][0];
      const v4i az3 = half ? sp[q][2] : zero;
      const v4i a4z = half ? zero : sp[q][3];
      const v4i a34 = half ? sp[q][3] : sp[q][2];
      mm(acc[q][2], a01, t18, sc_hi); mm(acc[q][2], az3, t19, sc_hi); mm(acc[q][2], a4z, t20, sc_hi);
      mm(acc[q][2], a01, t23, sc_lo); mm(acc[q][2], a34, t24, sc_lo);
    }
  }
  if (split) {
    const int nw = nch < 4 ? nch : 4;                       // waves that had a chunk
    if constexpr (WIDE) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        if (q) __syncthreads();
        if (wave != 0 && wave < nw) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[3 * (wave - 1) + j][r][lane] = acc[q][j][r];
        }
        __syncthreads();
        if (wave == 0) {
          for (int w = 1; w < nw; ++w)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[q][j][r] += red[3 * (w - 1) + j][r][lane];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q)
        for (int w = 1; w < 4; ++w) {
          if (wave == w && w < nw) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) red[j][r][lane] = acc[q][j][r];
          }
          __syncthreads();
          if (wave == 0 && w < nw) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[q][j][r] += red[j][r][lane];
          }
          __syncthreads();
        }
    }
    if (wave != 0) return;
  }
  const int co = g * 32 + (lane & 31);
  const double sc = a.scale[co], bi = a.bias[co];
  const float bna = a.bn_a[co], bnb = a.bn_b[co];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int b = b0 + 2 * q + half;                        // accumulator lane half == image within the tile's pair
    const bool ok = b < Bn;
    if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;  // (a ragged last unit: no image in this pair)
    float v = 0.f;
    unsigned mybits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const double s1 = fma((double)acc[q][0][r], 1024.0, (double)acc[q][1][r]);    // exact
      const double s = fma(s1, 1024.0, (double)acc[q][2][r]);                       // exact: |s| < 2^43
      const float y = exact_preact(s, sc, bi);                                      // the one rounding to fp32
      const bool sp1 = spk_lif_step_default(v, fmaf(y, bna, bnb)) && ok;
      mybits |= sp1 ? (1u << r) : 0u;
    }
    const long long cell = ((long long)(ok ? b : 0) * G + g) * HW + (HW - 1);
    store_tile_spikes(a.out, a.out_cnt, mybits, lane, cell * POSB, cell * 32, ok);
  }
}

// (2) One flagged neuron, exactly, by a 256-thread workgroup: thread unit (tap, 32-channel chunk, t) reads ONE 16-byte spike
// record and the 32 quantised weights of its (channel, tap, chunk) from the int32 table the pack kernel wrote (the same
// rint(w * 2^s) as the digit tiles) and sums the active ones in 64 bits; partial sums meet per time step through two lane
// shuffles and 16 LDS atomics per wave; then the first wave runs the exact epilogue in every lane (fp64 recombination of the
// exact sum -- the arithmetic of den_mfma_fp6.hip -- the reference's BN and LIF steps) and lane t patches the nibble of step t.
// Eight such workgroups fit a CU: a few thousand flagged neurons are one or two rounds, bound by the L2 reads (18 - 55 KB a neuron).
template <int H, int W>
__device__ __forceinline__ void fixup_neuron(const V2Args& a, long long n, unsigned long long* sS, int* sQ, int Bn) {
  constexpr int HW = H * W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nch = a.nch, Cin = a.Cin;
  const int p = (int)(n % HW);
  const long long r0 = n / HW;
  const int co = (int)(r0 % a.Cout), b = (int)(r0 / a.Cout);
  if (b >= Bn) return;                                     // (uniform over the workgroup)
  const int py = p / W, px = p % W;
  if (tid < 16) sS[tid] = 0ull;
  long long part = 0;
  if (SPK_V2_FIX_LDS && a.fix_lds) {
    // the channel's weights, once per neuron: [9][Cin] int32 (the previous neuron's readers are behind its last barrier)
    const int4* src = reinterpret_cast<const int4*>(a.qtab + (long long)co * 9 * Cin);
    for (int i = tid; i < 9 * Cin / 4; i += (int)blockDim.x) reinterpret_cast<int4*>(sQ)[i] = src[i];
    __syncthreads();
    const int wave = tid >> 6, nwv = (int)blockDim.x >> 6, t = lane & 15, q = lane >> 4;
    for (int pr = wave; pr < 9 * nch; pr += nwv) {          // a wave: one (tap, chunk) per trip; lane = (t, channels 8 q .. 8 q + 7)
      const int cc = pr % nch, tap = pr / nch;
      const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const unsigned nib = *reinterpret_cast<const unsigned*>(a.in0 + (((long long)b * nch + cc) * HW + yy * W + xx) * POSB + t * 16 + 4 * q);
      const int4* qp = reinterpret_cast<const int4*>(sQ + tap * Cin + cc * 32 + 8 * q);
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
  } else {
    __syncthreads();
    const int units = 9 * nch * 16;
    for (int u = tid; u < units; u += (int)blockDim.x) {
      const int t = u & 15, cc = (u >> 4) % nch, tap = (u >> 4) / nch;
      const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const uint4 sp = *reinterpret_cast<const uint4*>(a.in0 + (((long long)b * nch + cc) * HW + yy * W + xx) * POSB + t * 16);
      const int4* qp = reinterpret_cast<const int4*>(a.qtab + ((long long)co * 9 + tap) * Cin + cc * 32);
      const unsigned w4[4] = {sp.x, sp.y, sp.z, sp.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int4 q = qp[j];
        const unsigned nib = w4[j >> 1] >> (16 * (j & 1));     // four nibbles: channels 4j .. 4j + 3
        part += (nib & 0x000fu) ? (long long)q.x : 0ll;
        part += (nib & 0x00f0u) ? (long long)q.y : 0ll;
        part += (nib & 0x0f00u) ? (long long)q.z : 0ll;
        part += (nib & 0xf000u) ? (long long)q.w : 0ll;
      }
    }
  }
  // lanes l, l + 16, l + 32, l + 48 of a wave hold the same time step (t = u & 15; the block size is a multiple of 64)
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const int lo = __shfl_xor((int)(unsigned)(part & 0xffffffffll), off);
    const int hi = __shfl_xor((int)(part >> 32), off);
    part += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  }
  if (lane < 16 && part != 0) atomicAdd(&sS[lane], (unsigned long long)part);
  __syncthreads();
  if (tid < 64) {
    const double sc = a.scale[co], bi = a.bias[co];
    const float bna = a.bn_a[co], bnb = a.bn_b[co];
    float v = 0.f;
    unsigned bits = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float y = exact_preact((double)(long long)sS[t], sc, bi);
      bits |= spk_lif_step_default(v, fmaf(y, bna, bnb)) ? (1u << t) : 0u;
    }
    const int g = co >> 5;
    if (lane < 16) {
      uint8_t* rec = a.out + ((((long long)b * (a.Cout >> 5) + g) * HW + p) * POSB);
      const int byte = (co & 31) >> 1;
      const unsigned shw = 8u * (byte & 3) + 4u * (co & 1);
      unsigned* wp = reinterpret_cast<unsigned*>(rec + lane * 16 + (byte & ~3));
      atomicAnd(wp, ~(0xFu << shw));
      if ((bits >> lane) & 1u) atomicOr(wp, 0x2u << shw);
    }
    if (a.out_cnt && lane == 0) a.out_cnt[(((long long)b * (a.Cout >> 5) + g) * HW + p) * 32 + (co & 31)] = (uint8_t)__popc(bits);
  }
  __syncthreads();
}

template <int H, int W>
__device__ __forceinline__ void fp6v2_fixup_body(const V2Args& a, long long n_words, const unsigned bid, const unsigned nb,
                                                 unsigned long long* sS, int* sQ) {
  if (SPK_V2_DBG & 64) return;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  const unsigned count = a.flags[a.handover ? 1 : 0];       // published by the main launch (fp6v2_handover), or live
  const unsigned nlist = count < a.flag_cap ? count : a.flag_cap;
  for (unsigned e = bid; e < nlist; e += nb) fixup_neuron<H, W>(a, (long long)a.flags[2 + e], sS, sQ, Bn);
  if (count > a.flag_cap) {
    // overflow path (more than flag_cap flagged neurons): the rest sit in the bitmap; scan a share of it, clear as we go
    unsigned* bm = a.flags + 2 + FLAG_CAP;
    const long long per = (n_words + nb - 1) / nb;
    const long long w0 = (long long)bid * per;
    const long long w1 = w0 + per < n_words ? w0 + per : n_words;
    for (long long wi = w0; wi < w1; ++wi) {
      unsigned wv = bm[wi];                                // (uniform over the workgroup)
      __syncthreads();
      if (threadIdx.x == 0 && wv) bm[wi] = 0u;
      while (wv) {
        const int bit = __ffs((int)wv) - 1;
        wv &= wv - 1;
        fixup_neuron<H, W>(a, wi * 32 + bit, sS, sQ, Bn);
      }
    }
  }
}

// (1b) Round 5: the last position with the weight tiles SHARED through LDS.  fp6v2_lastpos_body reads the thirteen tiles of every
// chunk straight from L2 per image pair: 320 MB per launch for the 256 -> 512 layer at B = 256, which is what its ~20 us are.
// Here a workgroup takes FOUR image pairs of one channel group -- a wave each, every wave walks all the chunks -- and a chunk's
// thirteen tiles (19.5 KB: the four contributing taps' digit pairs, three fifth-digit tiles, two sixth-digit tiles; four runs
// of the slab) come in ONCE per workgroup by LDS-DMA, double-buffered (39 KB), one barrier per chunk: 80 MB.  No partial sums
// cross waves; the epilogue is the exact one of fp6v2_lastpos_body.
constexpr int LPS_TILES = 13, LPS_BUF = LPS_TILES * WT;                  // 19 968 B per chunk
constexpr size_t LPS_LDS = 2 * (size_t)LPS_BUF;
template <int H, int W>
__device__ __forceinline__ void fp6v2_lastpos_shared_body(const V2Args& a, const int bid) {
  constexpr int HW = H * W;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = a.nch, G = a.Cout >> 5;
  const int Bn = a.n_dyn ? (*a.n_dyn < a.B ? *a.n_dyn : a.B) : a.B;
  const int g = bid % G, b0 = (bid / G) * 8;
  if (b0 >= Bn) return;                                     // (uniform over the workgroup)
  const int row = lane & 31, half = lane >> 5;
  const int hsel = (row >> 2) & 1, tt = (row & 3) + 4 * (row >> 3);
  const int sc_a = 0x7f7f7f7f;
  const int sc_p = half ? (int)0x82828282u : (int)0x87878787u;
  const int sc_hi = (int)0x87878787u, sc_lo = (int)0x82828282u;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_addr = spk_lds_addr(lds), lane16 = (unsigned)lane * 16u;
  // the slab's four runs -> twenty pieces (one of them half a piece), five per wave: piece k of the wave = 5 * wave + k
  auto issue_chunk = [&](int c, int buf) {
    const uint8_t* slab = a.wq + ((long long)g * nch + c) * W_SLAB;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int pc = 5 * wave_s + k;                        // 0..19
      // pieces 0..5: tiles 0..3; 6..11: tiles 6..9; 12..16: tiles 18..20 (16 = the half piece); 17..19: tiles 23, 24
      const int src = pc < 6 ? pc * 1024 : pc < 12 ? 6 * WT + (pc - 6) * 1024 : pc < 17 ? 18 * WT + (pc - 12) * 1024
                                                                                        : 23 * WT + (pc - 17) * 1024;
      const int dst = pc < 6 ? pc * 1024 : pc < 12 ? 4 * WT + (pc - 6) * 1024 : pc < 17 ? 8 * WT + (pc - 12) * 1024
                                                                                        : 11 * WT + (pc - 17) * 1024;
      const unsigned long long mask = pc == 16 ? 0xffffffffull : ~0ull;
      spk_dma16s_masked(slab + src, lane16, lds_addr + (unsigned)(buf * LPS_BUF + dst), mask);
    }
  };
  const int py = H - 1, px = W - 1;
  const int bA = b0 + 2 * wave + hsel;                      // the image whose rows this lane feeds (A layout)
  auto load_spikes = [&](int c, v4i (&sp)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tap = (u >> 1) * 3 + (u & 1);
      const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
      sp[u] = v4i{0, 0, 0, 0};
      if (bA < Bn) sp[u] = *reinterpret_cast<const v4i*>(a.in0 + (((long long)bA * nch + c) * HW + yy * W + xx) * POSB + tt * 16);
    }
  };
  v16f acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v4i sp[4], spn[4];
  issue_chunk(0, 0);
  load_spikes(0, sp);
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of chunk c (and its spike fragments) are in
    __syncthreads();                                        // everyone's are; everyone is done with the other buffer
    if (c + 1 < nch) { issue_chunk(c + 1, buf ^ 1); load_spikes(c + 1, spn); }
    const uint8_t* Wb = lds + buf * LPS_BUF;
    // (the assembly form of the main kernel: four- and six-register operands; the builtin wants eight-register tuples padded
    //  with zeros for both, which is what pushed this body past 168 registers)
    auto ldb = [&](int tile) -> v6i {
      const uint8_t* wt = Wb + tile * WT;
      const v4i bx = *reinterpret_cast<const v4i*>(wt + lane * 16);
      const v2i by = *reinterpret_cast<const v2i*>(wt + 1024 + lane * 8);
      return v6i{bx[0], bx[1], bx[2], bx[3], by[0], by[1]};
    };
    auto mm = [&](v16f& d, const v4i& av, const v6i& bv, int sb) { SPK_MFMA2("v", d, av, bv, sc_a, sb); };
    // (fences: at most four tiles in flight -- left alone hipcc requests all thirteen first: 188 registers, two workgroups per CU)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const v6i t0 = ldb(2 * u), t1 = ldb(2 * u + 1);
      mm(acc[0], sp[u], t0, sc_p); mm(acc[1], sp[u], t1, sc_p);
      if (u & 1) __builtin_amdgcn_sched_barrier(0);
    }
    const v4i zero = {0, 0, 0, 0};
    const v4i a01 = half ? sp[1] : sp[0];
    const v4i az3 = half ? sp[2] : zero;
    const v4i a4z = half ? zero : sp[3];
    const v4i a34 = half ? sp[3] : sp[2];
    {
      const v6i t8 = ldb(8), t9 = ldb(9), t10 = ldb(10);
      mm(acc[2], a01, t8, sc_hi); mm(acc[2], az3, t9, sc_hi); mm(acc[2], a4z, t10, sc_hi);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
      const v6i t11 = ldb(11), t12 = ldb(12);
      mm(acc[2], a01, t11, sc_lo); mm(acc[2], a34, t12, sc_lo);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nch) {
#pragma unroll
      for (int u = 0; u < 4; ++u) sp[u] = spn[u];
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // (the MFMAs are opaque to the hazard recognizer: accumulator reads below)
  const int co = g * 32 + (lane & 31);
  const double sc = a.scale[co], bi = a.bias[co];
  const float bna = a.bn_a[co], bnb = a.bn_b[co];
  const int b = b0 + 2 * wave + half;                       // accumulator lane half == image within the pair
  const bool ok = b < Bn;
  if (__builtin_amdgcn_ballot_w64(ok) == 0ull) return;
  float v = 0.f;
  unsigned mybits = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const double s1 = fma((double)acc[0][r], 1024.0, (double)acc[1][r]);
    const double s2 = fma(s1, 1024.0, (double)acc[2][r]);
    const float y = exact_preact(s2, sc, bi);
    const bool sp1 = spk_lif_step_default(v, fmaf(y, bna, bnb)) && ok;
    mybits |= sp1 ? (1u << r) : 0u;
  }
  const long long cell = ((long long)(ok ? b : 0) * G + g) * HW + (HW - 1);
  store_tile_spikes(a.out, a.out_cnt, mybits, lane, cell * POSB, cell * 32, ok);
}

// The tail launches.  PART 0: workgroups [0, n_lp) compute last positions, the rest repair flagged neurons -- independent work
// (disjoint outputs, both read only the layer's input), each a chain of dependent L2 reads: in one launch they overlap (the
// sampler's active-set calls).  PART 1 / 2: repair only / last positions only, each with its own register and LDS budget (full
// batches, where either part fills the device by itself: the merged form measured 1.3 % slower there).
// PART 3 (round 5): PART 0 with the last positions of fp6v2_lastpos_shared_body (39 KB of dynamic LDS per workgroup: four per CU).
// (HIP's second launch-bound argument = waves per SIMD the kernel must leave room for: four workgroups of PART 3 per CU -> 128 registers;
//  one address spills, outside the chunk loop)
template <int H, int W, int PART>
__global__ __launch_bounds__(256, PART == 3 ? 4 : 1) void fp6v2_tail_kernel(V2Args a, long long n_words, int n_lp) {
  __shared__ float red[(PART == 1 || PART == 3) ? 1 : (PART == 2 ? 9 : 3)][16][64];
  __shared__ unsigned long long sS[16];
  extern __shared__ __attribute__((aligned(16))) uint8_t tail_lds[];          // repair: the neuron's weights (SPK_V2_FIX_LDS); PART 3: the shared tiles
  int* const sQ = reinterpret_cast<int*>(tail_lds);
  if constexpr (PART == 3) {
    if ((int)blockIdx.x < n_lp) { if constexpr ((H * W) & 1) fp6v2_lastpos_shared_body<H, W>(a, (int)blockIdx.x); }
    else fp6v2_fixup_body<H, W>(a, n_words, blockIdx.x - (unsigned)n_lp, gridDim.x - (unsigned)n_lp, sS, sQ);
    (void)red;
  } else {
    if (PART == 2 || (PART == 0 && (int)blockIdx.x < n_lp)) fp6v2_lastpos_body<H, W, PART == 2>(a, (int)blockIdx.x, red);
    else fp6v2_fixup_body<H, W>(a, n_words, blockIdx.x - (unsigned)n_lp, gridDim.x - (unsigned)n_lp, sS, sQ);
  }
}

// ------------------------------------------------------------------------------------------------ weight packing
// one block per output channel: channel maximum -> shift s, every weight -> six balanced radix-32 digits, written as the
// per-lane 24-byte B fragments (lane = K half * 32 + channel within the group of 32; 32 six-bit codes, little-endian;
// bytes 0..15 in the ds_read_b128 part of the tile, 16..23 in its ds_read_b64 part).  Tile order per (group, 32-channel
// chunk): (tap, pair) x 18 [K half 0: even digit, half 1: odd digit, same 32 input channels], fifth digit x 5 [K half 0:
// tap 2q, half 1: tap 2q + 1], sixth digit x 2 [taps (0, 1) and (3, 4)].  Also the L1 norm of the quantised weights.
__global__ __launch_bounds__(256) void pack_fp6v2_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                         uint8_t* __restrict__ wq, double* __restrict__ scale,
                                                         double* __restrict__ bias_d, float* __restrict__ wl1,
                                                         int* __restrict__ qtab, int Cout, int Cin) {
  __shared__ float smax[256];
  __shared__ double ssum[256];
  const int co = blockIdx.x, n = Cin * 9;
  const float* wc = w + (long long)co * n;
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
  double l1 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double q = rint(ldexp((double)wc[i], sh));
    l1 += fabs(q);
    const int ci = i / 9, tap = i - 9 * ci;
    qtab[((long long)co * 9 + tap) * Cin + ci] = (int)q;          // |q| < 2^29
  }
  ssum[threadIdx.x] = l1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) ssum[threadIdx.x] += ssum[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    scale[co] = ldexp(1.0, -sh);
    bias_d[co] = bias ? (double)bias[co] : 0.0;
    wl1[co] = (float)(ldexp(ssum[0], -sh) * 1.000001);      // rounded up: it feeds an upper bound
  }
  const int nchunks = Cin / CK, g = co >> 5, col = co & 31;
  constexpr int NTILE = N_MAIN + N_L5;
  for (int rec = threadIdx.x; rec < nchunks * NTILE * 2; rec += 256) {
    const int kh = rec & 1, tau = (rec >> 1) % NTILE, c = rec / (2 * NTILE);
    int tap, digit;
    bool valid = true;
    if (tau < N_PAIR) { tap = tau >> 1; digit = 2 * (tau & 1) + kh; }
    else if (tau < N_MAIN) { tap = 2 * (tau - N_PAIR) + kh; digit = 4; valid = tap < 9; }
    else { tap = (tau == N_MAIN ? 0 : 3) + kh; digit = 5; }
    unsigned bits[6] = {0, 0, 0, 0, 0, 0};
    if (valid) {
      for (int j = 0; j < 32; ++j) {
        const int ci = c * CK + j;
        long long q = (long long)rint(ldexp((double)wc[ci * 9 + tap], sh));
        int dg[6];
#pragma unroll
        for (int p = 5; p >= 1; --p) {
          const int r = (int)(((q + 16) & 31) - 16);
          dg[p] = r;
          q = (q - r) >> 5;
        }
        dg[0] = (int)q;                          // in [-16, 16]
        int d = 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) d = (p == digit) ? dg[p] : d;
        const unsigned code = (d < 0 ? 0x20u : 0u) | (unsigned)(d < 0 ? -d : d);
        const int bit = 6 * j, wd = bit >> 5, sft = bit & 31;
#pragma unroll
        for (int q2 = 0; q2 < 6; ++q2) {         // static register indexing
          if (q2 == wd) bits[q2] |= code << sft;
          if (q2 == wd + 1 && sft > 26) bits[q2] |= code >> (32 - sft);
        }
      }
    }
    uint8_t* tile = wq + (long long)(g * nchunks + c) * W_SLAB + tau * WT;
    const int ln = kh * 32 + col;
    unsigned* d16 = reinterpret_cast<unsigned*>(tile + ln * 16);
    unsigned* d8 = reinterpret_cast<unsigned*>(tile + 1024 + ln * 8);
    d16[0] = bits[0]; d16[1] = bits[1]; d16[2] = bits[2]; d16[3] = bits[3];
    d8[0] = bits[4]; d8[1] = bits[5];
  }
}

// fp32 spikes [T,B,C,HW] <-> nibble-packed S32 [B][C/32][HW][T][16] (tests, module boundaries)
__global__ void spikes_to_s32_kernel(const float* __restrict__ s, uint8_t* __restrict__ o, int T, int B, int C, int HW) {
  const long long total = (long long)B * (C / 32) * HW * T * 16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int byte = (int)(i & 15);
    long long r = i >> 4;
    const int t = (int)(r % T); r /= T;
    const int p = (int)(r % HW); r /= HW;
    const int cc = (int)(r % (C / 32));
    const int b = (int)(r / (C / 32));
    const int c0 = cc * 32 + 2 * byte;
    const float s0 = s[(((long long)t * B + b) * C + c0) * HW + p], s1 = s[(((long long)t * B + b) * C + c0 + 1) * HW + p];
    o[i] = (uint8_t)((s0 != 0.f ? 0x02 : 0) | (s1 != 0.f ? 0x20 : 0));
  }
}
__global__ void s32_to_spikes_kernel(const uint8_t* __restrict__ q, float* __restrict__ s, int T, int B, int C, int HW) {
  const long long total = (long long)T * B * C * HW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    long long r = i / HW;
    const int c = (int)(r % C); r /= C;
    const int b = (int)(r % B);
    const int t = (int)(r / B);
    const uint8_t by = q[((((long long)b * (C / 32) + c / 32) * HW + p) * T + t) * 16 + (c % 32) / 2];
    s[i] = ((by >> (4 * (c & 1))) & 0xf) ? 1.0f : 0.0f;
  }
}

}  // namespace

extern "C" long long spk_den_packed_weight_fp6v2_bytes(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0 || (Cout % 32) || (Cin % CK)) return -1;
  return (long long)(Cout / 32) * (Cin / CK) * W_SLAB;
}

extern "C" int spk_den_pack_weight_fp6v2(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d,
                                         float* wl1, int* qtab, int Cout, int Cin, hipStream_t stream) {
  if (!w || !wq || !scale || !bias_d || !wl1 || !qtab || Cout <= 0 || Cin <= 0) return SPK_ERR_ARG;
  if ((Cout % 32) || (Cin % CK)) return SPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pack_fp6v2_kernel, dim3(Cout), dim3(256), 0, stream, w, bias, wq, scale, bias_d, wl1, qtab, Cout, Cin);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" long long spk_den_fp6v2_flag_words(int B, int Cout, int H, int W) {
  if (B <= 0 || Cout <= 0 || H <= 0 || W <= 0) return -1;
  long long words = 2 + (long long)FLAG_CAP + ((long long)B * Cout * H * W + 31) / 32 + 1;   // (+ ticket)
#if SPK_V2_VARIANTS
  words += DUO_CU_SLOTS + DUO_ITEM_CTRS + (long long)spk_cu_count() * ZSTAGE_WORDS_PER_WG;   // duo counters, deferred-scan staging slabs
#endif
  return words;
}

static int fp6v2_launch(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale, const double* bias_d,
                        const float* wl1, const int* qtab, const float* bn_a, const float* bn_b, uint8_t* out_s32,
                        uint8_t* out_counts, unsigned* flag_words, int T, int B, int H, int W, int Cout,
                        const int* n_dyn_or_null, const uint8_t* need, int need_R, int need_r, int flag_cap, int form,
                        hipStream_t stream, int parts = 7) {
  const bool split_small = form == 0;                        // (form 1: whole-image items at any batch size)
  if (!in_s32 || nch <= 0 || !wq || !scale || !bias_d || !wl1 || !qtab || !bn_a || !bn_b || !out_s32 || !flag_words ||
      B <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return SPK_ERR_ARG;
  const bool bands = H == 8 && W == 8;
  if (T != T16 || (Cout % 32) || !((H == 7 && W == 7) || bands)) return SPK_ERR_UNSUPPORTED;
  if (need && (bands || !n_dyn_or_null)) return SPK_ERR_UNSUPPORTED;
  V2Args a;
  a.in0 = in_s32; a.nch = nch; a.wq = wq; a.scale = scale; a.bias = bias_d; a.wl1 = wl1; a.qtab = qtab;
  a.bn_a = bn_a; a.bn_b = bn_b; a.out = out_s32; a.out_cnt = out_counts; a.flags = flag_words;
  a.flag_cap = flag_cap < 0 || (unsigned)flag_cap > FLAG_CAP ? FLAG_CAP : (unsigned)flag_cap;   // (< 0: the whole list)
  a.n_dyn = n_dyn_or_null;
  a.dbg_out = nullptr; a.cu_slots = nullptr; a.item_ctr = nullptr; a.duo_delay = 0; a.zstage = nullptr;
  a.need = nullptr; a.cls_cnt = nullptr; a.cls_list = nullptr;
  if (need) {
    a.need = need + spk_need_off_rec(B, need_R, need_r);
    a.cls_cnt = reinterpret_cast<const int*>(need + spk_need_off_cnt(need_r));
    a.cls_list = reinterpret_cast<const int*>(need + spk_need_off_list(B, need_R, need_r));
  }
  a.B = B; a.Cout = Cout; a.Cin = nch * CK;
  const size_t fix_bytes = (size_t)9 * a.Cin * 4;             // (18 KB for the 512-channel layers)
  a.fix_lds = SPK_V2_FIX_LDS && fix_bytes <= LPS_LDS ? 1 : 0;
  const size_t fix_lds = a.fix_lds ? fix_bytes : 0;
  const int cus = spk_cu_count();
  const int G = Cout / 32;
  // XCD-aware walk: the largest power-of-two group count per XCD whose packed weights fit ~1.5 MB of its 4 MB L2
  int grid = 0;
  a.gx = 0; a.nsets = 1;
  // (wgs: workgroups of the launch -- one per CU, or two for the duo form; returns the grid, 0 if the XCD-aware walk does not fit)
  // (the sampler's active-set and listed calls keep the rounds 2-5 bound: with few image lanes per channel group two sets measured 1.2 %
  //  slower there -- elimination + lists 34.37 against 33.96 ms, three alternating passes -- while full batches are indifferent in time and
  //  fetch a third less: profiles/r6_ab_kernel_variants.txt (2))
  const long long gx_kb = (n_dyn_or_null || need) ? 1536 : SPK_V2_GX_KB;
  auto xcd_walk = [&](V2Args& v, int wgs) -> int {
    v.gx = 0; v.nsets = 1;
    if ((wgs & 7) != 0) return 0;
    const int S = wgs / 8;
    int gx = 1;
    while (gx * 2 <= G && gx * 2 <= S && (long long)gx * 2 * nch * W_SLAB <= gx_kb * 1024) gx *= 2;
    while (G / gx > 8 && gx * 2 <= G && gx * 2 <= S) gx *= 2;
    const int nsets = G / gx;
    if (G % gx == 0 && nsets <= 8 && 8 % nsets == 0 && S % gx == 0) { v.gx = gx; v.nsets = nsets; return wgs; }
    return 0;
  };
  grid = xcd_walk(a, cus);
  if (a.gx == 0) grid = cus >= G ? (cus / G) * G : G;         // flat walk: workgroup k -> group k % G
  const int a_bytes = bands ? ((8 / 2 + 1 + 2) * 9 + 1) * POSB : ((7 + 2) * 8 + 1) * POSB;
  // (+ the active-input counters of the four-digit form: s_cin [cells][16], s_row [positions + 1][16])
  const size_t lds = 2 * ((size_t)a_bytes + W_LDS) + (size_t)((a_bytes / POSB) + (bands ? 32 : 49) + 1) * 64 + 256;   // (+ s_nmax)
  // (the eight-wave kernels of full items keep SPK_V2_NBUF slab buffers)
  const size_t lds8 = lds + (SPK_V2_AHEAD && !USE_D4 ? (size_t)(SPK_V2_NBUF - 2) * ((size_t)a_bytes + W_LDS) : 0);
  const long long n_words = ((long long)B * Cout * H * W + 31) / 32;
  a.ticket_idx = 2 + (long long)FLAG_CAP + n_words;
  // full 7x7 batches keep the round-1 order (repair launch, then the last-position launch re-arms the counter): the hand-over
  // costs a barrier and an atomic per workgroup of the main launch (1.2 us), which only the merged tail launch of the
  // active-set calls (and the launch it saves on even latents) pays back
  a.handover = (bands || n_dyn_or_null || SPK_V2_MERGE_FULL) ? 1 : 0;
  if (parts != 7) {
    // measurement / debugging: re-run one tail part of the LAST launch on this workspace.  2 = the exact recomputation of
    // the neurons that launch flagged (count in ws[1], ids still listed: idempotent), 4 = the last position of every image.
    if (parts == 2) {
      a.handover = 1;                                         // (read the published count, leave the live counter alone)
      if (bands) hipLaunchKernelGGL((fp6v2_tail_kernel<8, 8, 1>), dim3(8 * cus), dim3(256), fix_lds, stream, a, n_words, 0);
      else hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 1>), dim3(8 * cus), dim3(256), fix_lds, stream, a, n_words, 0);
    } else if (parts == 4 && !bands) {
      a.handover = 1;                                         // (no re-arming)
      const int n_lp4 = ((B + 2 * SPK_V2_LP_PAIRS - 1) / (2 * SPK_V2_LP_PAIRS)) * G;
      hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 2>), dim3(n_lp4), dim3(256), 0, stream, a, n_words, n_lp4);
    } else return SPK_ERR_UNSUPPORTED;
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  if (bands) {
#if SPK_V2_VARIANTS
    if (spk_opt(SPK_OPT_V2_WAVES) == 4) hipLaunchKernelGGL((conv3x3_fp6v2_kernel<8, 8, 4, true>), dim3(grid), dim3(256), lds, stream, a);
    else
#endif
    hipLaunchKernelGGL((conv3x3_fp6v2_kernel<8, 8, 8, true>), dim3(grid), dim3(512), lds8, stream, a);   // (eight waves: +4 % over four)
    SPK_LAUNCH_CHECK();
    hipLaunchKernelGGL((fp6v2_tail_kernel<8, 8, 1>), dim3(8 * cus), dim3(256), fix_lds, stream, a, n_words, 0);   // (even latent: repair only)
    SPK_LAUNCH_CHECK();
    return SPK_OK;
  }
  // Two waves per SIMD are the default since the four-digit form: with 23 MFMAs per tile and chunk they needed 9 % fewer
  // cycles per MFMA and took the SAME time (the device lowered its clock by those 9 %: the launch was bound by the power the
  // matrix pipe may draw); with 18 the copies, fragment reads and popcounts of a chunk are no longer hidden behind one wave's
  // MFMAs, there is power to spare, and the second wave is worth 8 % of the reverse process.  A third one (SPKDIFF_V2_WAVES=12:
  // twelve waves of two tiles, 168 registers, accumulators in VGPRs; bit-equal) LOSES 11 % (round 3, same box: den.conv4 launch
  // 417-466 against 383-387 us, dense reverse process 103.3 against 92.3 ms) although three waves issue vector instructions
  // 1.4x faster than two (tools/coexec_probe.hip: 2.2 against 3.1 cycles per v_fma_f32): a weight tile read from LDS then
  // serves two row tiles instead of three and twelve waves meet at every chunk barrier.  SPKDIFF_V2_WAVES=4: one wave.
#if SPK_V2_VARIANTS
  const bool eight = spk_opt(SPK_OPT_V2_WAVES) != 4, twelve = spk_opt(SPK_OPT_V2_WAVES) == 12;
  const bool lag_form = spk_opt(SPK_OPT_V2_LAG) != 0 || SPK_V2_LAG_DEFAULT != 0;
  if (need) {
    if (grid / G < 6) return SPK_ERR_UNSUPPORTED;           // one image lane per tile-count class at least
    if (eight) hipLaunchKernelGGL((conv3x3_fp6v2_listed_kernel<7, 7, 8>), dim3(grid), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((conv3x3_fp6v2_listed_kernel<7, 7, 4>), dim3(grid), dim3(256), lds, stream, a);
  } else if (eight && !lag_form && !twelve && nch >= 4 && spk_opt(SPK_OPT_V2_DEFER) != 0 && spk_opt(SPK_OPT_V2_DUO) == 0) {
    // round 5: the scan of an item runs inside the next item's K loop (fp6v2_body_defer); two count buffers by item parity
    const size_t lds_d = 2 * ((size_t)a_bytes + W_LDS) + ((size_t)((a_bytes / POSB) * 16 + 2 * 50 * 16 + 2 * 50) * 4 + 15) / 16 * 16 + 8 * 4096;
    a.zstage = reinterpret_cast<float*>(flag_words + a.ticket_idx + 1 + DUO_CU_SLOTS + DUO_ITEM_CTRS);
    hipLaunchKernelGGL((conv3x3_fp6v2_defer_kernel<7, 7>), dim3(grid), dim3(512), lds_d, stream, a);
  } else if (eight && !lag_form && !twelve && nch >= 2 && G * 8 <= DUO_ITEM_CTRS && spk_opt(SPK_OPT_V2_DUO) != 0 &&
             (long long)DUO_LDS <= spk_lds_limit()) {
    // round 5: two independent four-wave workgroups per CU on half-image items (fp6v2_body_duo)
    V2Args ad = a;
    ad.cu_slots = flag_words + a.ticket_idx + 1;
    ad.item_ctr = ad.cu_slots + DUO_CU_SLOTS;
    ad.handover = 1;                                          // (the duo kernel always publishes the count: merged tail launch below)
    // v2_duo: 1 = the pair of a CU starts together (measured: the older wave of a SIMD gets the matrix pipe first, the two run at
    // different speeds and their scans fall into each other's K loops by themselves: 0.77 - 0.80 of the scan time with or without
    // a head start, profiles/r5_ab_duo_first_build.txt); >= 16: a head start of that many 10 ns ticks per chunk for a CU's first
    // workgroup (fp6v2_duo_phase)
    const int dmode = spk_opt(SPK_OPT_V2_DUO);
    ad.duo_delay = dmode >= 16 ? dmode * nch : 0;
    if (SPK_V2_DUO_DBG) ad.dbg_out = reinterpret_cast<unsigned long long*>(flag_words + 2 + FLAG_CAP / 2);   // (upper half of the id list)
    int grid2 = xcd_walk(ad, 2 * cus);
    if (grid2 == 0) grid2 = 2 * cus >= G ? (2 * cus / G) * G : G;
    hipLaunchKernelGGL((conv3x3_fp6v2_duo_kernel<7, 7>), dim3(grid2), dim3(256), DUO_LDS, stream, ad);
  } else if (eight && lag_form) {
    // (staggered experiment: three ring slots + counters + one 128-byte line per wave)
    const size_t lds3 = 3 * ((size_t)a_bytes + W_LDS) + (size_t)(a_bytes / POSB) * 64 + 8 * 128;
    hipLaunchKernelGGL((conv3x3_fp6v2_lag_kernel<7, 7>), dim3(grid), dim3(512), lds3, stream, a);
  } else if (twelve) hipLaunchKernelGGL((conv3x3_fp6v2_kernel<7, 7, 12>), dim3(grid), dim3(768), lds, stream, a);
  else if (eight) hipLaunchKernelGGL((conv3x3_fp6v2_kernel<7, 7, 8>), dim3(grid), dim3(512), lds8, stream, a);
  else hipLaunchKernelGGL((conv3x3_fp6v2_kernel<7, 7, 4>), dim3(grid), dim3(256), lds, stream, a);
#else
  // (the four- and twelve-wave, staggered, duo and deferred-scan forms are `make variants` builds: csrc/variants/fp6v2_forms.inc)
  if (need) {
    if (grid / G < 6) return SPK_ERR_UNSUPPORTED;           // one image lane per tile-count class at least
    hipLaunchKernelGGL((conv3x3_fp6v2_listed_kernel<7, 7, 8>), dim3(grid), dim3(512), lds, stream, a);
  } else if (split_small && (long long)B * G * SPK_V2_HALF_FILL <= grid) {
    // fewer items than workgroups: two half-image items per image on four-wave workgroups (B = 16: conv2 / conv3 / conv5)
    hipLaunchKernelGGL((conv3x3_fp6v2_half_kernel<7, 7>), dim3(grid), dim3(256), lds, stream, a);
  } else hipLaunchKernelGGL((conv3x3_fp6v2_kernel<7, 7, 8>), dim3(grid), dim3(512), lds8, stream, a);
#endif
  SPK_LAUNCH_CHECK();
  const int n_lp = ((B + 2 * SPK_V2_LP_PAIRS - 1) / (2 * SPK_V2_LP_PAIRS)) * G;
  if (!n_dyn_or_null && SPK_V2_MERGE_FULL && spk_opt(SPK_OPT_V2_LPS) != 0 && nch >= 2 && B >= SPK_V2_LPS_MIN_B &&
      (long long)LPS_LDS + 4096 <= spk_lds_limit()) {
    // round 5, full batches: last positions with LDS-shared weight tiles (eight images per workgroup), repairs beside them (four
    // workgroups per CU).  Same box, B = 256: den.conv4 / conv5 launches 386 / 373 -> 380 / 369 us, dense reverse process 91.6 -> 90.9 ms
    // (profiles/r5_ab_kernel_variants.txt (3)).  The sampler's active-set calls keep the form below: with few images the shared
    // form has too few units (elimination + lists 33.6 -> 34.7 ms)
    const int n_lps = ((B + 7) / 8) * G;
    hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 3>), dim3(n_lps + 4 * cus), dim3(256), LPS_LDS, stream, a, n_words, n_lps);
    SPK_LAUNCH_CHECK();
  } else if (n_dyn_or_null || SPK_V2_MERGE_FULL) {
    // the sampler's active-set calls: few images, both parts are latency bound -- one launch (-17 us per reverse step)
    hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 0>), dim3(n_lp + 8 * cus), dim3(256), fix_lds, stream, a, n_words, n_lp);
    SPK_LAUNCH_CHECK();
  } else {
    // full batches: both parts fill the device on their own; measured 1 % faster one after the other
    hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 1>), dim3(8 * cus), dim3(256), fix_lds, stream, a, n_words, 0);
    SPK_LAUNCH_CHECK();
    hipLaunchKernelGGL((fp6v2_tail_kernel<7, 7, 2>), dim3(n_lp), dim3(256), 0, stream, a, n_words, n_lp);
    SPK_LAUNCH_CHECK();
  }
  return SPK_OK;
}

extern "C" int spk_den_conv3x3_mfma_fp6v2(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale,
                                          const double* bias_d, const float* wl1, const int* qtab, const float* bn_a,
                                          const float* bn_b, uint8_t* out_s32, uint8_t* out_counts, unsigned* flag_words,
                                          int T, int B, int H, int W, int Cout, const int* n_dyn_or_null, int flag_cap,
                                          int form, hipStream_t stream) {
  if (form != 0 && form != 1) return SPK_ERR_ARG;
  return fp6v2_launch(in_s32, nch, wq, scale, bias_d, wl1, qtab, bn_a, bn_b, out_s32, out_counts, flag_words, T, B, H, W, Cout,
                      n_dyn_or_null, nullptr, 0, 0, flag_cap, form, stream);
}

extern "C" int spk_den_conv3x3_mfma_fp6v2_part(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale,
                                               const double* bias_d, const float* wl1, const int* qtab, const float* bn_a,
                                               const float* bn_b, uint8_t* out_s32, uint8_t* out_counts, unsigned* flag_words,
                                               int T, int B, int H, int W, int Cout, const int* n_dyn_or_null, int part,
                                               int flag_cap, hipStream_t stream) {
  if (part != 2 && part != 4) return SPK_ERR_ARG;
  return fp6v2_launch(in_s32, nch, wq, scale, bias_d, wl1, qtab, bn_a, bn_b, out_s32, out_counts, flag_words, T, B, H, W, Cout,
                      n_dyn_or_null, nullptr, 0, 0, flag_cap, 1, stream, part);
}

extern "C" int spk_den_conv3x3_mfma_fp6v2_listed(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale,
                                                 const double* bias_d, const float* wl1, const int* qtab, const float* bn_a,
                                                 const float* bn_b, uint8_t* out_s32, uint8_t* out_counts,
                                                 unsigned* flag_words, int T, int B, int H, int W, int Cout,
                                                 const int* n_dyn, const uint8_t* need, int need_radii, int radius,
                                                 int flag_cap, hipStream_t stream) {
  if (!need || !n_dyn || need_radii <= 0 || need_radii > 8 || radius < 1 || radius > need_radii) return SPK_ERR_ARG;
  return fp6v2_launch(in_s32, nch, wq, scale, bias_d, wl1, qtab, bn_a, bn_b, out_s32, out_counts, flag_words, T, B, H, W, Cout,
                      n_dyn, need, need_radii, radius - 1, flag_cap, 1, stream);
}

extern "C" int spk_spikes_to_s32(const float* spikes, uint8_t* out_s32, int T, int B, int C, int HW, hipStream_t stream) {
  if (!spikes || !out_s32 || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 32) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)B * (C / 32) * HW * T * 16;
  hipLaunchKernelGGL(spikes_to_s32_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)), dim3(256),
                     0, stream, spikes, out_s32, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}

extern "C" int spk_s32_to_spikes(const uint8_t* in_s32, float* spikes, int T, int B, int C, int HW, hipStream_t stream) {
  if (!spikes || !in_s32 || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SPK_ERR_ARG;
  if (C % 32) return SPK_ERR_UNSUPPORTED;
  const long long total = (long long)T * B * C * HW;
  hipLaunchKernelGGL(s32_to_spikes_kernel, dim3(spk_blocks(total, 256) > 65536 ? 65536 : spk_blocks(total, 256)), dim3(256),
                     0, stream, in_s32, spikes, T, B, C, HW);
  SPK_LAUNCH_CHECK();
  return SPK_OK;
}
