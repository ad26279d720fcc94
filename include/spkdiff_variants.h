/* libspkdiff VARIANT builds only (`make -C spiking-diffusion_amd/csrc variants` -> spkdiff/variants/libspkdiff_variants.so,
 * compiled with -DSPK_V2_VARIANTS=1).  The shipped libspkdiff.so exports NEITHER function and keeps no process-wide state: there the
 * options below are compile-time constants equal to the defaults (csrc/spk_common.h).  This header exists so that the recorded A/B
 * measurements (profiles/r*_ab_*.txt) can be repeated: tools/ab.py, tests/variants/. */
#ifndef SPKDIFF_VARIANTS_H
#define SPKDIFF_VARIANTS_H
#ifdef __cplusplus
extern "C" {
#endif

/* Measurement options (variant builds): launch-shape choices that were measured against each other (DESIGN.md quotes the numbers) and stay
 * selectable so that the measurements can be repeated.  The library never reads the environment; a host that wants another
 * form calls spk_set_option (process-wide, relaxed atomics; every launch reads the current value, so a host may switch between
 * calls).  spkdiff/_lib.py forwards the SPKDIFF_<NAME> environment variables once at import for the A/B tools under tools/.
 *   name               default  meaning
 *   "v2_waves"            8     waves per workgroup of spk_den_conv3x3_mfma_fp6v2's main launch: 8 (two per SIMD), 4 (one), 12
 *                               (three, accumulators in VGPRs: measured -11 %)
 *   "v2_lag"              0     1: full 7x7 batches run the staggered form (waves 4..7 one chunk behind: measured 4-13 % slower)
 *   "v2_duo"              0     1: full 7x7 batches run two independent four-wave workgroups per CU on half-image items (one workgroup's
 *                               LIF scan beside the other's MFMAs: measured 4-5 % slower, profiles/r5_ab_duo_*.txt); >= 16: with a head
 *                               start of that many 10 ns ticks per chunk for a CU's first workgroup
 *   "v2_defer"            0     1: full 7x7 batches, layers of >= 4 chunks: the LIF scan of an item runs inside the K loop of the same
 *                               waves' next item (software pipelining across items: measured 11-40 % slower, profiles/r5_ab_defer_builds.txt);
 *                               0: scan between two K loops (rounds 2-4)
 *   "v2_lps"              1     7x7 latents: the tail launch's last-position part shares a chunk's weight tiles through LDS (eight images per
 *                               workgroup); 0: every image pair reads them from L2 (rounds 2-4)
 *   "fp6_waves"           4     8: spk_den_conv3x3_mfma_fp6 with two waves per SIMD where an item has <= 4 row tiles per wave
 *   "fp6_xcd_walk"        1     0: image-major item walk of spk_den_conv3x3_mfma_fp6 (2.2x the HBM-side traffic)
 *   "conv6_shared"        1     0: spk_den_conv3x3_counts_mfma never shares operands through LDS
 *   "conv6_shared_dyn"    1     0: ... not in the sampler's active-set calls
 *   "mfma_debug"          0     ablation builds (-DSPK_MFMA_ABLATION) only: 1 no steady-state DMA, 2 no MFMAs, 4 no epilogue
 * Returns SPK_ERR_UNSUPPORTED for an unknown name. */
int spk_set_option(const char* name, int value);
int spk_get_option(const char* name, int* value_out);

#ifdef __cplusplus
}
#endif
#endif
