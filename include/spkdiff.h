/* libspkdiff -- C-ABI of the MI355X-native (gfx950) Spiking-Diffusion inference kernels.
 *
 * This is the drop-in boundary for the reference's time-stepped SNN inference path.  The reference has no
 * FFI of its own for this path: its operator-plugin point is spikingjelly's ``backend`` switch
 * (SJ/activation_based/base.py:199-208, functional.py:109-149), whose only native plugin is the CuPy one with the
 * call contract  LIFNodeATGF.apply(x_seq[T,N], v[N], v_th, v_reset, 1/tau, ...)  (SJ/activation_based/neuron.py:954-966,
 * auto_cuda/neuron_kernel.py:496-522).  Every entry point below names the reference interface it replaces
 * (R/ = Spiking-Diffusion-release/, SJ/ = member of R/spikingjelly.zip).  INTEGRATION.md shows the ctypes stub.
 *
 * Contract (all functions):
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch); the library never
 *     allocates or frees device memory, reads no environment variable and keeps no mutable global state (the launch-shape
 *     options of earlier rounds are compile-time constants here; only a `make variants` build -- include/spkdiff_variants.h --
 *     keeps them settable, for repeating the recorded A/B measurements);
 *   - work is enqueued asynchronously on ``stream`` (a hipStream_t; pass torch.cuda.current_stream().cuda_stream);
 *   - returns 0 on success, SPK_ERR_* (< 0) for argument errors, a positive hipError_t for launch failures;
 *   - reentrant across streams and host threads.
 *
 * Layouts: "TBCHW" = fp32 [T,B,C,H,W] contiguous (the reference's tensors); "PTC" = u8 {0,1} [B,H,W,T,C]
 * (the library's compact inter-layer spike format); packed conv weights = fp32 [k*k][Cin][Cout].
 */
#ifndef SPKDIFF_H
#define SPKDIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* spk_stream_t; /* == hipStream_t */

#define SPK_VERSION 105 /* 0.1.5 -- bumped whenever an exported signature changes or entry points are added (round 4 inserted `int K`
                           * before the stream of spk_select_active / spk_select_needed: 101; round 5 added the VectorQuantizer's
                           * training branch and the training convolutions: 102, the token-table spike generator: 103; round 6: `int flag_cap` (and `int form` for spk_den_conv3x3_mfma_fp6v2) in front of the
                           * stream of the four certified-kernel entry points, spk_set_option / spk_get_option left the shipped library: 104;
                           * `active` / `n_active` of spk_den_step_tail: 105); spkdiff/_lib.py refuses a library whose
                           * spk_version() differs from the signatures it declares */

/* fused-kernel epilogue modes (spk_conv_fused_fwd) */
#define SPK_CHUNK_C4 (-64) /* chunk_out value: fp4 nibble-packed output, 64 channels per chunk */
#define SPK_CHUNK_S32 (-32) /* chunk_out value: fp4 nibble-packed output, 32 channels per chunk ("S32", fp6v2 kernel) */
#define SPK_MODE_LIF 0    /* BN + LIF -> spikes                                    */
#define SPK_MODE_RAW 1    /* conv output per time step, fp32 TBCHW                 */
#define SPK_MODE_MEMOUT 2 /* sum_t x[t]*coef[t] (+tanh, +uint8)  -> [B,C,H,W]      */
#define SPK_MODE_MEAN 3   /* sum_t x[t] / T                      -> [B,C,H,W]      */

/* input kinds of spk_conv_fused_fwd */
#define SPK_IN_PTC 0  /* u8 spikes [B,H,W,T,C]                                               */
#define SPK_IN_TINV 1 /* fp32 [B,C,H,W], the same frame at every time step                  */
#define SPK_IN_SEQ 2  /* fp32 [T,B,C,H,W], arbitrary values (the reference's tensor format) */

int spk_version(void);
/* Returns a static string for an SPK_ERR_* / hipError_t code. */
const char* spk_error_string(int code);


/* ---- neuron surface -------------------------------------------------------------------------------------- */

/* Multi-step eval LIF (hard reset, decay_input): replaces LIFNode.multi_step_forward eval branch
 * SJ/activation_based/neuron.py:971-1011 -> jit_eval_multi_step_forward_hard_reset_decay_input :799-811, and mirrors the
 * cupy plugin contract :954-966.  x_seq [T,N] fp32; v_inout [N] fp32 (state before / after); spike_out [T,N] as
 * spike_dtype 0 = fp32, 1 = u8, 2 = bit-packed u64 words [T, ceil(N/64)] (bit l of word w = neuron 64w+l). */
int spk_lif_fwd(const float* x_seq, float* v_inout, void* spike_out, int T, long long N, float tau, float v_threshold,
                float v_reset, int spike_dtype, spk_stream_t stream);
/* The other eval forms of the reference neuron, SJ/activation_based/neuron.py:827-900 (dispatch :971-1011): soft reset
 * (v_reset = None there; `v_reset` is ignored), decay_input = False, and -- v_seq_out non-NULL -- the `..._with_v_seq` variants
 * (membrane potential after every step, fp32 [T, N]).  fp32 spikes; v [N] updated in place. */
int spk_lif_fwd_ex(const float* x_seq, float* v_inout, float* spike_out_f32, float* v_seq_out_or_null, int T, long long N,
                   float tau, float v_threshold, float v_reset, int soft_reset, int decay_input, spk_stream_t stream);


/* The table behind the time-invariant-input layers (spk_conv_fused_fwd with SPK_IN_TINV and no carried state): the default
 * neuron (tau 2, v_th 1, v_reset 0: SJ/activation_based/neuron.py:799-811 with the models' constructor arguments,
 * R/snn_model/vae_model.py:112) driven by a CONSTANT input from v = 0 fires with period p(x); thresholds16[k-1] is the
 * smallest fp32 input whose first spike comes at step k or earlier, patterns18[p] the sixteen spike bits of period p
 * (p = 17: none).  Host-side copy-out, no GPU needed: the CPU tests check it against the fp32 recurrence on every float. */
int spk_lif_const_input_table(float* thresholds16, unsigned* patterns18);

/* ---- stateless step-mode layers ---------------------------------------------------------------------------- */

/* Eval BatchNorm as PyTorch evaluates it (pinned by fixture F7): a = (1/sqrt(var+eps))*gamma, b = fma(-mean,a,beta). */
int spk_bn_prepare(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                   float eps, float* a_out, float* b_out, int C, spk_stream_t stream);
/* y = fma(x, a[c], b[c]) over x [M,C,HW]: layer.BatchNorm2d 'm' mode, SJ/activation_based/layer.py:458-465. */
int spk_bn_eval_fwd(const float* x, const float* a, const float* b, float* y, long long M, int C, int HW,
                    spk_stream_t stream);
/* layer.Conv2d 'm' mode body (T folded into M), SJ/activation_based/layer.py:164-173. w [Cout,Cin,k,k]. */
int spk_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int Cin, int H, int W,
                   int Cout, int k, int stride, int pad, spk_stream_t stream);
/* layer.ConvTranspose2d 'm' mode body, SJ/activation_based/layer.py:316-325. w [Cin,Cout,k,k]. */
int spk_conv_transpose2d_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int Cin, int H,
                             int W, int Cout, int k, int stride, int pad, int out_pad, spk_stream_t stream);
/* MembraneOutputLayer.forward, R/snn_model/snn_layers.py:36-41: out[N] = sum_t x_seq[t][N]*coef[t]. */
int spk_memout_fwd(const float* x_seq, const float* coef, float* out, int T, long long N, spk_stream_t stream);

/* ---- surrogate-gradient LIF (training path) ---------------------------------------------- */
/* LIFNode training forward (hard reset, decay_input): SJ/activation_based/neuron.py:739-749,133-135; the native pair it
 * mirrors: LIFNodeFPTTKernel / LIFNodeBPTTKernel, SJ/activation_based/auto_cuda/neuron_kernel.py:102-225,479-540.
 * h_seq [T,N] (membrane potential before reset, kept for the backward), spike_seq [T,N] fp32, v_out [N] final state. */
int spk_lif_train_fwd(const float* x_seq, const float* v_init, float* h_seq, float* spike_seq, float* v_out, int T,
                      long long N, float tau, float v_threshold, float v_reset, spk_stream_t stream);
/* BPTT with the ATan surrogate g'(x) = alpha/2 / (1 + (pi/2 alpha x)^2) (SJ/activation_based/surrogate.py:664-678).
 * grad_v_last [N] (gradient of the final state) and grad_v_init [N] may be NULL.  detach_reset as in the reference. */
int spk_lif_train_bwd(const float* grad_spike_seq, const float* grad_v_last, const float* h_seq, float* grad_x_seq,
                      float* grad_v_init, int T, long long N, float tau, float v_threshold, float v_reset, float alpha,
                      int detach_reset, spk_stream_t stream);

/* Training-mode BatchNorm2d ('m' mode, batch statistics: SJ/activation_based/layer.py:458-465 -> F.batch_norm(training=True))
 * fused with the surrogate-gradient LIF above -- one denoiser block tail of DummyModel.forward in train() mode
 * (R/snn_model/vq_diffusion.py:163-183,199-203).  CHANNELS-LAST memory: y, spike_seq, grad_* are [T][B][HW][C] fp32
 * (what the library's NHWC convolutions read and write), v_init / v_out / grad_v_* [B][HW][C]; gamma/beta/running_* /
 * save_* [C]  (v_init NULL = reset state, v_out / running_* / grad_v_* may be NULL).
 * running_mean / running_var are updated in place with `momentum` (unbiased variance), save_mean / save_invstd receive
 * the batch statistics the backward needs.  ws: caller-allocated scratch of spk_bn_lif_train_ws_bytes(B, C, HW) bytes.
 * The backward recomputes the membrane potentials from y (only y and the two statistics vectors are kept). */
long long spk_bn_lif_train_ws_bytes(int B, int C, int HW);
int spk_bn_lif_train_fwd(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, const float* v_init, float* spike_seq, float* v_out, float* save_mean,
                         float* save_invstd, void* ws, long long ws_bytes, int T, int B, int C, int HW, float tau,
                         float v_threshold, float v_reset, spk_stream_t stream);
int spk_bn_lif_train_bwd(const float* grad_spike_seq, const float* grad_v_last, const float* y, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd, const float* v_init,
                         float* grad_y, float* grad_gamma, float* grad_beta, float* grad_v_init, void* ws,
                         long long ws_bytes, int T, int B, int C, int HW, float tau, float v_threshold, float v_reset,
                         float alpha, int detach_reset, spk_stream_t stream);
/* The same backward with the layout of grad_spike_seq spelled out (in floats): grad_step_stride = 0 when every step receives
 * the SAME gradient -- the spikes in front of the denoiser's last layer do: its time mean hands g / T to every step
 * (R/snn_model/vq_diffusion.py:205-206) -- and grad_row_pitch >= C for a channel slice of a wider channels-last tensor (the x5 / x1
 * halves of cat(x5, x1), :205).  The gradient is read where autograd left it instead of being expanded and copied first.
 * spk_bn_lif_train_bwd == grad_step_stride B*HW*C, grad_row_pitch C. */
int spk_bn_lif_train_bwd_strided(const float* grad_spike_seq, long long grad_step_stride, long long grad_row_pitch,
                                 const float* grad_v_last, const float* y, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, const float* v_init, float* grad_y,
                                 float* grad_gamma, float* grad_beta, float* grad_v_init, void* ws, long long ws_bytes, int T,
                                 int B, int C, int HW, float tau, float v_threshold, float v_reset, float alpha,
                                 int detach_reset, spk_stream_t stream);
/* The same forward, also leaving the spikes as "C4" records [B][C/64][HW][T][32 bytes = 64 channels x e2m1] (spikes_c4_out, or
 * NULL) -- the input format of spk_den_conv3x3_fp6_raw, i.e. of the NEXT block's convolution in the training forward; the apply
 * launch writes them next to the fp32 spikes.  Needs C % 64 == 0 and 16-byte aligned tensors (SPK_ERR_UNSUPPORTED otherwise,
 * nothing launched). */
int spk_bn_lif_train_fwd_c4(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, const float* v_init, float* spike_seq, float* v_out, float* save_mean,
                            float* save_invstd, uint8_t* spikes_c4_out, void* ws, long long ws_bytes, int T, int B, int C,
                            int HW, float tau, float v_threshold, float v_reset, spk_stream_t stream);
/* Training forward of a denoiser convolution on spike input (layer.Conv2d 'm' mode in train(), SJ/activation_based/layer.py:164-173,
 * for conv2..conv5 of DummyModel, R/snn_model/vq_diffusion.py:166-184): the exact fp6 x fp4 MFMA convolution of
 * spk_den_conv3x3_mfma_fp6 with the pre-activations (conv + bias, correctly rounded fp32 of the exact dot product) written
 * out instead of the fused BN + LIF -- batch-statistics BN needs all of them first.  in_c4 as for the inference kernel;
 * pre_nhwc fp32 channels-last [T][B][H*W][Cout].  spk_spikes_nhwc_to_fp4 packs channels-last fp32 spikes [T][B][HW][C]
 * (the output of spk_bn_lif_train_fwd) into the C4 input layout. */
int spk_den_conv3x3_fp6_raw(const uint8_t* in_c4, int nch, const uint8_t* wq, const double* scale, const double* bias_d,
                            float* pre_nhwc, int T, int B, int H, int W, int Cout, spk_stream_t stream);
int spk_spikes_nhwc_to_fp4(const float* spikes_nhwc, uint8_t* out_c4, int T, int B, int C, int HW, spk_stream_t stream);
/* The same conversion with the per-neuron spike counts over T as a by-product: counts_nhwc fp32 [B][HW][C] (channels-last) -- what
 * the last layer's weight gradient is convolved with in the training step (its time mean hands every step the same gradient). */
int spk_spikes_nhwc_to_fp4_counts(const float* spikes_nhwc, uint8_t* out_c4, float* counts_nhwc, int T, int B, int C, int HW,
                                  spk_stream_t stream);

/* Masked cross-entropy of AbsorbingDiffusion._train_loss (R/snn_model/vq_diffusion.py:85-88: F.cross_entropy with
 * ignore_index=-1, reduction='none') and its gradient in one pass.  logits / dlogits [B,K,HW] fp32; target [B,HW] fp32
 * token ids (-1 = ignored, as x_0_ignore); coef [B] per-sample gradient factor; ce_out [B,HW] (0 where ignored).
 * dlogits (and coef) may be NULL for the forward alone. */
int spk_masked_ce(const float* logits, const float* target, const float* coef, float* ce_out, float* dlogits, int B, int K,
                  int HW, spk_stream_t stream);

/* PSP filter of the VQ-VAE training losses (R/snn_model/snn_layers.py:6-26; used at R/snn_model/vae_model.py:81-82):
 * out[t] = syn_t, syn_t = syn_{t-1} + (in[t] - syn_{t-1}) / tau_s, syn_{-1} = 0, over a dense [T][N] fp32 tensor.
 * backward != 0 runs the adjoint instead (in = dL/dsyn, out = dL/dx). */
int spk_psp(const float* in, float* out, int T, long long N, float tau_s, int backward, spk_stream_t stream);

/* ---- layout converters --------------------------------------------------------------------------------------- */
/* chunk = C gives plain PTC [B,HW,T,C]; chunk = 32 gives the channel-chunked "CPTC" [B,C/32,HW,T,32] the MFMA
 * kernel reads (one contiguous slab per image and 32-channel K chunk). */
int spk_spikes_to_ptc(const float* spikes_tbchw, uint8_t* out_bhwtc, int T, int B, int C, int HW, int chunk,
                      spk_stream_t stream);
int spk_ptc_to_spikes(const uint8_t* in_bhwtc, float* spikes_tbchw, int T, int B, int C, int HW, int chunk,
                      spk_stream_t stream);

/* ---- fused (Conv|ConvT) [+BN+LIF] --------------------------------------------------------------------------- */
int spk_conv_out_size(int in, int k, int stride, int pad, int transposed, int out_pad);
/* [Cout,Cin,k,k] (Conv2d) or [Cin,Cout,k,k] (ConvTranspose2d) -> packed [k*k][Cin][Cout]. */
int spk_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int k, int transposed, spk_stream_t stream);
/* One Conv|ConvT block of Encoder / poisson / Decoder / DummyModel with its BN + LIF (or read-out) fused:
 * R/snn_model/vae_model.py:34-38,109-124,139-155; R/snn_model/vq_diffusion.py:161-187,200-206.
 *   in0: per in_kind -- SPK_IN_TINV fp32 [B,C0,H,W] (same frame every step: R/main.py:309, vae_model.py:54-56,
 *        vq_diffusion.py:198), SPK_IN_PTC u8 [B,H,W,T,C0], SPK_IN_SEQ fp32 [T,B,C0,H,W];
 *   in1: optional second PTC source [B,H,W,T,C1] (channel concat of conv6, vq_diffusion.py:205).
 *   mode LIF:    bn_a/bn_b required; v_inout [B,Cout,Ho,Wo] or NULL (fresh state, not written back);
 *                out_ptc u8 PTC and/or out_f32 fp32 TBCHW spikes; out_pre optional BN output.
 *   mode RAW:    out_f32 TBCHW conv output.    mode MEMOUT: coef [T]; out_f32 [B,Cout,Ho,Wo] (tanh if apply_tanh),
 *                out_u8 = uint8(clip(p+0.5,0,1)*255) (R/main.py:401).   mode MEAN: out_f32 = sum_t x[t] / T.
 *   chunk0 / chunk1 / chunk_out: channel chunking of in0 / in1 / out_ptc (0: plain PTC); chunk_out = SPK_CHUNK_C4:
 *                out_ptc is written as nibble-packed fp4 "C4" (see spk_den_conv3x3_mfma_fp6; Cout % 64 == 0);
 *                SPK_CHUNK_S32: as "S32" (see spk_den_conv3x3_mfma_fp6v2; Cout % 32 == 0).
 *   out_counts (mode LIF, optional): per-neuron spike counts over T, u8 [B,Cout/32,Ho*Wo,32].
 *   n_dyn_or_null: optional device-side image count (<= B, see spk_select_active): only images [0, *n_dyn) are computed;
 *                buffers stay sized by B. */
int spk_conv_fused_fwd(const void* in0, const uint8_t* in1, int C0, int C1, int in_kind, const float* w_packed,
                       const float* bias, const float* bn_a, const float* bn_b, float* v_inout, uint8_t* out_ptc,
                       float* out_f32, float* out_pre, uint8_t* out_u8, const float* coef, int apply_tanh, int mode,
                       int T, int B, int H, int W, int Cout, int k, int stride, int pad, int transposed, int out_pad,
                       int chunk0, int chunk1, int chunk_out, uint8_t* out_counts, const int* n_dyn_or_null,
                       spk_stream_t stream);

/* ---- denoiser convolutions on the matrix cores ------------------------------------------------------------------ */
/* Bytes of the packed int8 digit-plane weights of one 3x3 layer ([Cout/16][Cin/32][9][2][32][32]); -1 if unsupported. */
long long spk_den_packed_weight_bytes(int Cout, int Cin);
/* fp32 conv weight [Cout,Cin,3,3] (+bias) -> four balanced base-256 int8 digit planes + per-channel 2^-s scale and
 * fp64 bias.  One-time weight preparation for spk_den_conv3x3_mfma. */
int spk_den_pack_weight_i8(const float* w, const float* bias, int8_t* wq, double* scale, double* bias_d, int Cout,
                           int Cin, spk_stream_t stream);
/* 3x3 / stride 1 / pad 1 convolution over binary spikes (CPTC u8, T = 16) + BN + LIF (mode SPK_MODE_LIF -> out_cptc)
 * or + time mean (mode SPK_MODE_MEAN -> out_f32 [B,Cout,h,w]): DummyModel conv2..conv6,
 * R/snn_model/vq_diffusion.py:166-187,201-206.  in1 (nch1 chunks) is concatenated after in0 along channels.
 * v_inout [B,Cout,h,w] or NULL (fresh LIF state, nothing written back).  out_counts (LIF mode, optional): per-neuron
 * spike counts over T as u8 [B,Cout/32,h*w,32], the input format of spk_den_conv3x3_counts_mfma.
 * n_dyn_or_null as in spk_conv_fused_fwd. */
int spk_den_conv3x3_mfma(const uint8_t* in0_cptc, int nch0, const uint8_t* in1_cptc, int nch1, const int8_t* wq,
                         const double* scale, const double* bias_d, const float* bn_a, const float* bn_b,
                         float* v_inout, uint8_t* out_cptc, uint8_t* out_counts, float* out_f32, int mode, int T, int B,
                         int H, int W, int Cout, const int* n_dyn_or_null, spk_stream_t stream);
/* conv6 + time mean of DummyModel (R/snn_model/vq_diffusion.py:185-187,205-206) in its time-collapsed form:
 * (sum_t conv(s_t)) / T = (conv_linear(sum_t s_t) + T*bias) / T.  cnt0 / cnt1: spike counts u8 [B,nch,h*w,32]
 * (channel concat: cnt1 after cnt0); same packed weights as spk_den_conv3x3_mfma; out_f32 [B,Cout,h,w].
 * n_dyn_or_null as in spk_conv_fused_fwd. */
int spk_den_conv3x3_counts_mfma(const uint8_t* cnt0, int nch0, const uint8_t* cnt1, int nch1, const int8_t* wq,
                                const double* scale, const double* bias_d, float* out_f32, int T, int B, int H, int W,
                                int Cout, const int* n_dyn_or_null, spk_stream_t stream);

/* ---- denoiser convolutions on the block-scaled fp6/fp4 MFMA (CDNA4 v_mfma_scale_f32_32x32x64_f8f6f4) -------------- */
/* Same operator and numerics contract as spk_den_conv3x3_mfma (LIF mode: DummyModel conv2..conv5,
 * R/snn_model/vq_diffusion.py:166-184,201-204) at 3/4 of its matrix-core time: weights as six radix-32 fp6 (e2m3)
 * digit planes, spikes as fp4 (e2m1) nibbles.  Spike tensors are "C4": [B][C/64][h*w][16][32 B], channel c of a
 * 64-channel chunk in byte c/2 (low nibble first), nibble = 0x2 for a spike.
 * Bytes of the packed weights of one layer ([Cout/16][Cin/64] slabs of [9][3][64 lanes x 24 B], each padded to 41 KiB);
 * -1 if unsupported. */
long long spk_den_packed_weight_fp6_bytes(int Cout, int Cin);
/* fp32 conv weight [Cout,Cin,3,3] (+bias) -> six balanced radix-32 digit planes as e2m3 codes + per-channel 2^-s
 * scale and fp64 bias. */
int spk_den_pack_weight_fp6(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d, int Cout,
                            int Cin, spk_stream_t stream);
/* The same packing of a weight stored in the channels-last memory format, [Cout][3][3][Cin] (how the training path keeps its
 * convolution parameters, R/main.py:226-252 with NHWC library kernels): no layout copy in front of the per-iteration packing. */
int spk_den_pack_weight_fp6_cl(const float* w_channels_last, const float* bias, uint8_t* wq, double* scale, double* bias_d,
                               int Cout, int Cin, spk_stream_t stream);
/* The channels-last packing of n <= 8 layers in ONE launch (a training iteration re-packs every spike-input layer once per
 * optimizer step, R/main.py:243-252).  HOST arrays of n entries (bias may be NULL, or hold NULL entries); per layer the result
 * is spk_den_pack_weight_fp6_cl's byte for byte. */
int spk_den_pack_weight_fp6_cl_multi(const float* const* w_channels_last, const float* const* bias, uint8_t* const* wq,
                                     double* const* scale, double* const* bias_d, const int* Cout, const int* Cin, int n,
                                     spk_stream_t stream);
/* in_c4: nch chunks of 64 channels; out_c4 [B][Cout/64][h*w][16][32]; v_inout / out_counts as in spk_den_conv3x3_mfma.
 * SPK_ERR_UNSUPPORTED unless T == 16, Cout % 64 == 0 and the latent fits one of the kernel's LDS plans (up to 7x8 as
 * one item per image and channel group, 8x8 as two row bands).
 * n_dyn_or_null as in spk_conv_fused_fwd (the work items are then walked image-major). */
int spk_den_conv3x3_mfma_fp6(const uint8_t* in_c4, int nch, const uint8_t* wq, const double* scale, const double* bias_d,
                             const float* bn_a, const float* bn_b, float* v_inout, uint8_t* out_c4, uint8_t* out_counts,
                             int T, int B, int H, int W, int Cout, const int* n_dyn_or_null, spk_stream_t stream);
/* Second-generation form of spk_den_conv3x3_mfma_fp6 for the sampler (7x7 latents, fresh LIF state in, no state out):
 * the SAME spikes bit for bit -- the four leading digits on the matrix cores with adjacent digits sharing an accumulator through
 * the per-block scales, fp32 recombination, every spike decision certified against a bound on the dropped digits (per counted
 * active input of the row; tiles with a flagged lane are re-examined with the per-step running bound), and the ~1e-4 of neurons
 * that come closer to the threshold recomputed exactly (all six digits, int64 / fp64) by a tail launch; another finishes the 49th
 * position (csrc/den_mfma_fp6v2.hip).  DummyModel conv2..conv5, R/snn_model/vq_diffusion.py:166-184,201-204.
 * Spikes travel as "S32": [B][C/32][H*W][16][16 B] (fp4 nibbles, channel c of a group in byte (c % 32) / 2, low nibble first).
 * spk_den_pack_weight_fp6v2: fp32 [Cout,Cin,3,3] (+bias) -> packed digit tiles (spk_den_packed_weight_fp6v2_bytes), fp64
 * scale / bias [Cout] as for the fp6 kernel, wl1 [Cout] = L1 norm of each channel's quantised weights, and qtab = the
 * quantised weights themselves as int32 [Cout][9][Cin] (read by the exact recomputation).  flag_words: zero-initialised
 * u32 workspace of spk_den_fp6v2_flag_words(B, Cout, H, W) words (counter, ticket, id list, overflow bitmap); it is clean
 * again when the call's launches have run (word 1 keeps the number of neurons the call flagged, for statistics).  One
 * workspace per stream: two calls in flight at once must not share it.
 * flag_cap: how many flagged neurons the id list takes before the rest go to the overflow bitmap, which the tail launch scans and
 * clears (< 0 or > 2^20: the whole list of 2^20 entries -- what every product call passes).  The workspace layout does not depend on
 * it; the parity suite passes 64 and 0 so that the overflow path runs (tests/test_gpu_parity.py::test_flag_overflow_*).
 * form: 0 = automatic (7x7 latents with fewer items -- B x Cout / 32 -- than half the CUs run two half-image items per image on
 * four-wave workgroups: R/main.py's own n_samples = 16 leaves half the chip idle otherwise), 1 = whole-image items always (the
 * reference form for the bit-equality test of the split).  Same spikes bit for bit either way.
 * SPK_ERR_UNSUPPORTED unless T == 16, H == W == 7 or 8, Cout % 32 == 0 (Cin = 32 * nch). */
long long spk_den_packed_weight_fp6v2_bytes(int Cout, int Cin);
int spk_den_pack_weight_fp6v2(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d, float* wl1,
                              int* qtab, int Cout, int Cin, spk_stream_t stream);
long long spk_den_fp6v2_flag_words(int B, int Cout, int H, int W);
int spk_den_conv3x3_mfma_fp6v2(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale, const double* bias_d,
                               const float* wl1, const int* qtab, const float* bn_a, const float* bn_b, uint8_t* out_s32,
                               uint8_t* out_counts, unsigned* flag_words, int T, int B, int H, int W, int Cout,
                               const int* n_dyn_or_null, int flag_cap, int form, spk_stream_t stream);
/* Measurement aid: re-run ONE tail part of the last spk_den_conv3x3_mfma_fp6v2 call on the same arguments and workspace --
 * part 2: the exact recomputation of the neurons that call flagged (flag_words[1] holds their number, the id list is intact;
 * recomputing them again writes the same spikes), part 4: the last position of every image (7x7).  bench.py times these to
 * report `repair_ms` / `last_position_ms` per layer; flag_words[1] / (B * Cout * H * W) is the flagged fraction. */
int spk_den_conv3x3_mfma_fp6v2_part(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale,
                                    const double* bias_d, const float* wl1, const int* qtab, const float* bn_a,
                                    const float* bn_b, uint8_t* out_s32, uint8_t* out_counts_or_null, unsigned* flag_words,
                                    int T, int B, int H, int W, int Cout, const int* n_dyn_or_null, int part,
                                    int flag_cap, spk_stream_t stream);
/* The same layer restricted to the positions the sampler will read (7x7 only): need = the buffer written by
 * spk_select_needed(..., R = need_radii) for the same B; radius (1..need_radii) selects the lists this layer computes --
 * 1 for the layer whose output the logits convolution reads, 2 for the one below it, ...  Listed positions (and the 49th)
 * receive exactly the spikes of the full call; the other positions of out_s32 / out_counts are left as they were. */
int spk_den_conv3x3_mfma_fp6v2_listed(const uint8_t* in_s32, int nch, const uint8_t* wq, const double* scale,
                                      const double* bias_d, const float* wl1, const int* qtab, const float* bn_a,
                                      const float* bn_b, uint8_t* out_s32, uint8_t* out_counts, unsigned* flag_words, int T,
                                      int B, int H, int W, int Cout, const int* n_dyn, const uint8_t* need, int need_radii,
                                      int radius, int flag_cap, spk_stream_t stream);
/* fp32 spikes [T,B,C,HW] <-> S32 (C % 32 == 0): module boundaries and tests. */
int spk_spikes_to_s32(const float* spikes, uint8_t* out_s32, int T, int B, int C, int HW, spk_stream_t stream);
int spk_s32_to_spikes(const uint8_t* in_s32, float* spikes, int T, int B, int C, int HW, spk_stream_t stream);
/* fp32 spikes [T,B,C,HW] <-> C4 (C % 64 == 0): module boundaries and tests. */
int spk_spikes_to_fp4(const float* spikes, uint8_t* out_c4, int T, int B, int C, int HW, spk_stream_t stream);
int spk_fp4_to_spikes(const uint8_t* in_c4, float* spikes, int T, int B, int C, int HW, spk_stream_t stream);

/* ---- spiking VQ-VAE layers on the matrix cores ------------------------------------------------------------------- */
/* Bytes of the packed int8 digit planes of a k x k (transposed) conv: [ceil(Cout/16)][ceil(Cin/32)][k*k][2][32][32]. */
long long spk_conv_packed_weight_i8_bytes(int Cout, int Cin, int k);
/* Conv2d [Cout,Cin,k,k] or ConvTranspose2d [Cin,Cout,k,k] weight (+bias) -> int8 digit planes; scale / bias_d have
 * ceil(Cout/16)*16 entries (padding channels are zero). */
int spk_pack_conv_weight_i8(const float* w, const float* bias, int8_t* wq, double* scale, double* bias_d, int Cout,
                            int Cin, int k, int transposed, spk_stream_t stream);
/* (Conv2d | ConvTranspose2d) over binary spikes (plain PTC u8 [B,H*W,16,Cin], T = 16, Cin % 16 == 0) with exact int8
 * MFMA accumulation, fused with BN + LIF (mode SPK_MODE_LIF -> out_ptc [B,Ho*Wo,16,Cout]) or with the membrane
 * read-out (mode SPK_MODE_MEMOUT: coef[16] -> out_f32 [B,Cout,Ho,Wo] (+tanh), out_u8): Encoder conv2/conv3, Decoder
 * convT1/convT2/convT3 of R/snn_model/vae_model.py:115-124,139-155,186.
 * SPK_MODE_LIF with coef AND out_f32 given: out_f32 [B,Ho*Wo,Cout] also receives sum_t coef[t] * spike[t] (the input of
 * spk_readout_collapsed_fwd); out_ptc may then be NULL (the spike frames are not stored). */
int spk_conv_mfma_fused_fwd(const uint8_t* in_ptc, const int8_t* wq, const double* scale, const double* bias_d,
                            const float* bn_a, const float* bn_b, float* v_inout, uint8_t* out_ptc, const float* coef,
                            float* out_f32, uint8_t* out_u8, int apply_tanh, int mode, int T, int B, int H, int W, int Cin,
                            int Cout, int k, int stride, int pad, int transposed, int out_pad, spk_stream_t stream);
/* The decoder's linear last layer + membrane read-out (R/snn_model/vae_model.py:152-154,186, R/snn_model/snn_layers.py:36-41)
 * on time-collapsed spikes:  sum_t coef[t] * (W * s_t + bias) = W * (sum_t coef[t] * s_t) + bias * sum_t coef[t].
 * x_bpc fp32 [B,H*W,Cin] = sum_t coef[t] * s_t (see spk_conv_mfma_fused_fwd); w = the layer's fp32 weight, Conv2d
 * [Cout,Cin,k,k] or ConvTranspose2d [Cin,Cout,k,k] (transposed = 1); coef_sum = sum_t coef[t]; stride 1, pad == k / 2.
 * out_f32 [B,Cout,H,W] (tanh if apply_tanh), out_u8 = uint8(clip(p + 0.5, 0, 1) * 255) (R/main.py:401).  fp32 arithmetic:
 * equal to the frame-by-frame sum up to fp32 round-off. */
int spk_readout_collapsed_fwd(const float* x_bpc, const float* w, const float* bias_or_null, float coef_sum, float* out_f32,
                              uint8_t* out_u8, int apply_tanh, int B, int H, int W, int Cin, int Cout, int k, int pad,
                              int transposed, spk_stream_t stream);
/* spk_conv_mfma_fused_fwd in SPK_MODE_LIF writing nibble-packed "S32" spikes [B][Cout/32][Ho*Wo][16][16 B] (Cout % 32 == 0):
 * the input layout of the fp6 kernels. */
int spk_conv_mfma_fused_lif_s32(const uint8_t* in_ptc, const int8_t* wq, const double* scale, const double* bias_d,
                                const float* bn_a, const float* bn_b, float* v_inout, uint8_t* out_s32, int T, int B, int H,
                                int W, int Cin, int Cout, int k, int stride, int pad, int transposed, int out_pad,
                                spk_stream_t stream);
/* The spike-input 3x3 stride-2 layers of the spiking VQ-VAE from the reset state on the block-scaled fp6 x fp4 MFMA
 * (csrc/vae_fp6.hip): the same spikes as spk_conv_mfma_fused_fwd -- five digit planes on the matrix cores, certified decisions,
 * exact recomputation of the flagged neurons.  Supported (T == 16, Cout % 32 == 0):
 *   transposed = 1, Cin = 64, (H, W) in {(14,14), (16,16)}, out_kind 0: Decoder convT2 (R/snn_model/vae_model.py:146-150) -> out =
 *       fp32 [B][4*H*W][Cout] = sum_t coef[t] * spike[t], the input of spk_readout_collapsed_fwd (the spike frames are not stored);
 *   transposed = 1, Cin = 16, (H, W) in {(7,7), (8,8)}, out_kind 1: Decoder convT1 (:139-144) -> out = S32 spikes [B][Cout/32][4*H*W][16][16 B];
 *   transposed = 0, Cin = 32, (H, W) in {(14,14), (16,16)}, out_kind 2: Encoder conv2 (:115-118) -> out = u8 PTC [B][H*W/4][16][Cout].
 * in_s32: S32 spikes [B][ceil(Cin/32)][H*W][16][16 B] with zero nibbles in the channels beyond Cin (spk_ptc_to_s32 converts u8
 * PTC spikes).  spk_vae_fp6_pack: fp32 weight (Conv2d [Cout][Cin][3][3] / ConvTranspose2d [Cin][Cout][3][3]) (+bias) -> digit
 * tiles (spk_vae_fp6_packed_bytes), fp64 scale / bias [Cout], qtab int32 [Cout][9][Cin].  flag_words: zero-initialised u32
 * workspace of spk_vae_fp6_flag_words(B, Cout, Ho, Wo) words, clean again after the call; flag_cap as for
 * spk_den_conv3x3_mfma_fp6v2 (< 0: the whole id list). */
long long spk_vae_fp6_packed_bytes(int Cout, int Cin);
int spk_vae_fp6_pack(const float* w, const float* bias, uint8_t* wq, double* scale, double* bias_d, int* qtab, int Cout, int Cin,
                     int transposed, spk_stream_t stream);
long long spk_vae_fp6_flag_words(int B, int Cout, int Ho, int Wo);
int spk_ptc_to_s32(const uint8_t* in_ptc, uint8_t* out_s32, int T, int B, int HW, int C, spk_stream_t stream);
/* The decoder's front end by token (R/main.py:388-392, R/snn_model/vae_model.py:54-56,66-71: embedding look-up, repeat(T), the 'poisson'
 * spike generator = 1x1 Conv2d + BN + LIF from the reset state): the generator's input at a position is one of the K codebook rows, so
 * its spike train is one of K patterns per output channel.  One call = a K-row pattern table (the arithmetic of spk_conv_fused_fwd's
 * time-invariant form: fp64 dot product from the bias, BN fma, sixteen LIF steps; an out-of-range token embeds as NaN: no spikes) and
 * the S32 spikes [n_positions][16][16 B] of the positions' tokens -- the input layout of spk_vae_fp6_fwd (Cout 16 or 32, T == 16).
 * w_packed: [1][D][Cout] (spk_pack_conv_weight); table_ws: spk_spikegen_table_bytes(K, Cout) bytes, written when build_table != 0 and
 * only read otherwise (the caller keeps it while codebook, weights and BN terms are unchanged). */
long long spk_spikegen_table_bytes(int K, int Cout);
int spk_spikegen_tokens_s32(const long long* tokens, const float* codebook, const float* w_packed, const float* bias, const float* bn_a,
                            const float* bn_b, unsigned short* table_ws, int build_table, uint8_t* out_s32, int T, long long n_positions,
                            int K, int D, int Cout, spk_stream_t stream);
int spk_vae_fp6_fwd(const uint8_t* in_s32, const uint8_t* wq, const double* scale, const double* bias_d, const int* qtab,
                    const float* bn_a, const float* bn_b, const float* coef_or_null, void* out, int out_kind, unsigned* flag_words,
                    int T, int B, int H, int W, int Cin, int Cout, int transposed, int flag_cap, spk_stream_t stream);

/* ---- vector quantizer ----------------------------------------------------------------------------------------- */
/* VectorQuantizer.forward eval up to the codebook gather, R/snn_model/vae_model.py:40-52,87-99.
 * z_ptc u8 [B,h,w,T,D]; coef [T]; alpha [1] (device); codebook [K,D]; idx_out int64 [B*HW];
 * zq_out_bdhw fp32 [B,D,h,w] or NULL; xm_out fp32 [B*HW,D] or NULL (the read-out, for tests). */
int spk_vq_readout_argmin(const uint8_t* z_ptc, const float* coef, const float* alpha, const float* codebook,
                          long long* idx_out, float* zq_out_bdhw, float* xm_out, int T, int B, int D, int HW, int K,
                          spk_stream_t stream);
/* VectorQuantizer.get_code_indices on explicit rows flat_x [N,D], R/snn_model/vae_model.py:87-95. */
int spk_vq_argmin(const float* flat_x, const float* codebook, long long* idx_out, long long N, int D, int K,
                  spk_stream_t stream);
/* VectorQuantizer.quantize (nn.Embedding), R/snn_model/vae_model.py:97-99; nchw=1 also applies the
 * permute(0,3,1,2) of R/main.py:391 and writes [B,D,h,w].  Out-of-range tokens give NaN rows. */
int spk_embedding_fwd(const long long* tokens, const float* codebook, float* out, long long N, int D, int K, int HW,
                      int nchw, spk_stream_t stream);

/* Training branch of VectorQuantizer.forward, R/snn_model/vae_model.py:61-85 (through autograd in the reference: ~40 element-wise,
 * reduce and embedding-backward launches), csrc/vq_train.hip.  Rows are the NHWC positions (N = B * HW).
 * spk_vq_train_readout: x_seq fp32 [T,B,D,HW] -> xm [N,D] = (1 - alpha) * sum_t x[t] * coef[t] + alpha * sum_t x[t] / T and
 *   dxa [N,D] = d xm / d alpha.  (The code search on xm is spk_vq_argmin.)
 * spk_vq_train_quant: out_bdhw fp32 [B,D,HW] = xm + (E[idx] - xm) (the straight-through value) and loss_out[0] =
 *   mse(q, xm) + beta * mse(xm, q).  ws: spk_vq_train_ws_bytes() bytes, zero-initialised once, clean again after every call.
 * spk_vq_train_bwd: gout_bdhw = dL/d out, gloss_or_null [1] = dL/d loss (device) -> gx_seq [T,B,D,HW], galpha_out [1],
 *   gcodebook_out [K,D] (one workgroup per code, fixed summation order).  SPK_ERR_UNSUPPORTED for D > 64. */
long long spk_vq_train_ws_bytes(void);
int spk_vq_train_readout(const float* x_seq, const float* coef, const float* alpha, float* xm_out, float* dxa_out, int T, int B,
                         int D, int HW, spk_stream_t stream);
int spk_vq_train_quant(const float* xm, const long long* idx, const float* codebook, float* out_bdhw, float* loss_out, float beta,
                       void* ws, long long N, int D, int HW, spk_stream_t stream);
/* The PSP losses of the same branch, R/snn_model/vae_model.py:79-84: loss_out[0] = mean((psp(q) - psp(x))^2) * (1 + beta) over dense
 * [T][N] fp32 tensors (both filters in registers, nothing stored), and its backward: gq_seq = dL/dq, gx_seq = dL/dx (the commitment
 * factor beta on x) from gloss [1] (device) through the adjoint filters -- one launch each.  ws as for spk_vq_train_quant.
 * spk_psp_loss_bwd: SPK_ERR_UNSUPPORTED for T > 16. */
int spk_psp_loss_fwd(const float* q_seq, const float* x_seq, float* loss_out, float beta, float tau_s, void* ws, int T, long long N,
                     spk_stream_t stream);
int spk_psp_loss_bwd(const float* q_seq, const float* x_seq, const float* gloss, float* gq_seq, float* gx_seq, float beta,
                     float tau_s, int T, long long N, spk_stream_t stream);
/* Reconstruction loss of SNN_VQVAE.forward in training, R/snn_model/vae_model.py:189-196: xr_out [N] = tanh(sum_t y[t] * coef[t]),
 * loss_out[0] = mean((xr - image)^2) over N = B*C*H*W elements (one launch), and gy_seq [T][N] = dL/dy from gloss [1] (one launch). */
int spk_recon_loss_fwd(const float* y_seq, const float* coef, const float* image, float* xr_out, float* loss_out, void* ws, int T,
                       long long N, spk_stream_t stream);
int spk_recon_loss_bwd(const float* xr, const float* image, const float* coef, const float* gloss, float* gy_seq, int T, long long N,
                       spk_stream_t stream);
int spk_vq_train_bwd(const float* gout_bdhw, const float* gloss_or_null, const float* xm, const long long* idx,
                     const float* codebook, const float* dxa, const float* coef, const float* alpha, float beta, float* gx_seq,
                     float* galpha_out, float* gcodebook_out, void* ws, int T, long long N, int D, int HW, int K,
                     spk_stream_t stream);

/* Weight gradient of a 3x3 / stride 1 / pad 1 convolution over a SPIKE input (training step of the denoiser's conv2..conv6,
 * R/snn_model/vq_diffusion.py:166-187 through autograd; the reference runs the library's fp32 kernels):
 * gw[co][ky][kx][ci] = sum_{n,y,x} gy[n,co,y,x] * s[n,ci,y+ky-1,x+kx-1] on the bf16 matrix cores -- the spikes are exact in bf16,
 * the fp32 output gradient is split into three bf16 terms exactly, so only the fp32 accumulation rounds.  gy_cl / spikes_cl:
 * channels-last fp32 [N = T*B][H*W][C]; gw_out fp32 [Cout][3][3][Cin] (= a channels-last [Cout,Cin,3,3] tensor); ws: scratch of
 * spk_conv3x3_wgrad_ws_bytes (split-K partial sums, added in a fixed order: deterministic).  7x7 or 8x8 maps (H == W),
 * Cout % 128 == 0, Cin % 64 == 0; otherwise SPK_ERR_UNSUPPORTED (use the framework's operator).
 * The spike operand may also hold small non-negative integers (spike counts up to 256: exact in bf16) -- the time-collapsed
 * backward of the denoiser's last layer.
 * gb_out_or_null [Cout]: the bias gradient (sum of gy over images and positions), from the same pass over gy. */
long long spk_conv3x3_wgrad_ws_bytes(int N, int Cout, int Cin);
int spk_conv3x3_wgrad_bf16(const float* gy_cl, const float* spikes_cl, float* ws, long long ws_bytes, float* gw_out,
                           float* gb_out_or_null, int N, int H, int W, int Cout, int Cin, spk_stream_t stream);

/* Data gradient of the same convolution (autograd of layer.Conv2d in the training step, R/snn_model/vq_diffusion.py:166-187;
 * cuDNN's data-gradient kernels in the reference): gi[n][y][x][ci] = sum over (co, ky, kx) of gy[n][y+1-ky][x+1-kx][co] *
 * w[co][ky][kx][ci], channels-last fp32 tensors (gy [N][49][Cout], w [Cout][3][3][Cin], gi [N][49][Cin]).  Both operands are
 * split into three bf16 terms exactly and six cross products run on the bf16 matrix cores with fp32 accumulation (what is
 * dropped is below 2^-24 of a product): an fp32 GEMM's accuracy.  ws: spk_conv3x3_dgrad_ws_bytes(Cout, Cin) bytes (the packed
 * weight terms, rewritten by every call).  Deterministic.  7x7 or 8x8 maps (49 -> H * W), Cout % 16 == 0, Cin % 32 == 0; otherwise
 * SPK_ERR_UNSUPPORTED (use the framework's operator). */
long long spk_conv3x3_dgrad_ws_bytes(int Cout, int Cin);
int spk_conv3x3_dgrad_bf16(const float* gy_cl, const float* w_cl, uint8_t* ws, long long ws_bytes, float* gi_out, int N, int H,
                           int W, int Cout, int Cin, spk_stream_t stream);
/* The same data gradient with TWO fp16 terms per operand and three cross products (half the matrix work): every image's gy
 * and every input channel's weights are scaled by a power of two that puts their largest magnitude into [2^14, 2^15), x 2^s =
 * h + m to 2^-23 (h, m: nearest fp16 of the value and of the exact remainder), products (h,h) (h,m) (m,h), fp32 accumulation,
 * exact descaling.  Values more than 28 binades below their image's / channel's maximum lose relative (not absolute)
 * precision.  Same arguments, workspace and shapes as spk_conv3x3_dgrad_bf16. */
int spk_conv3x3_dgrad_f16x2(const float* gy_cl, const float* w_cl, uint8_t* ws, long long ws_bytes, float* gi_out, int N, int H,
                            int W, int Cout, int Cin, spk_stream_t stream);
/* spk_conv3x3_dgrad_f16x2 in two halves.  pack_multi: the weight half (per-channel maxima + the two fp16 term planes) of
 * n <= 8 layers in ONE launch -- inside a training iteration each layer's backward otherwise spends a fill, a maximum and a pack
 * launch on weights that change once per optimizer step.  HOST arrays of n entries; ws[i]: spk_conv3x3_dgrad_ws_bytes(Cout[i],
 * Cin[i]) bytes; N[i]: the image count the data half will be called with (the packed tile width depends on it).  prepacked: the
 * data half on such a workspace, bit-identical to the one-call form.  A workspace packed for another tile width (another N)
 * yields NaN, not a permuted result. */
int spk_conv3x3_dgrad_f16x2_pack_multi(const float* const* w_cl, uint8_t* const* ws, const long long* ws_bytes, const int* N,
                                       const int* Cout, const int* Cin, int n, spk_stream_t stream);
int spk_conv3x3_dgrad_f16x2_prepacked(const float* gy_cl, const uint8_t* ws, long long ws_bytes, float* gi_out, int N, int H,
                                      int W, int Cout, int Cin, spk_stream_t stream);

/* Weight and bias gradient of a 3x3 / stride 1 / pad 1 convolution with FEW input channels (Cin <= 4) and a dense fp32 input: the
 * denoiser's first layer in the training step (cat(x_t, t): two channels, R/snn_model/vq_diffusion.py:161-165,195-201; what
 * loss.backward() computes through layer.Conv2d, R/main.py:226-252).  gy_cl channels-last [N][H*W][Cout], in_nchw [N][Cin][H][W];
 * gw_out [Cout][3][3][Cin] (weight_channels_last = 1, the storage the training path keeps its weights in) or [Cout][Cin][3][3];
 * gb_out optional [Cout].  ws: spk_conv3x3_wgrad_small_ws_bytes(...) bytes; deterministic (fixed-order partial sums, fp64). */
long long spk_conv3x3_wgrad_small_ws_bytes(int N, int H, int W, int Cout, int Cin);
int spk_conv3x3_wgrad_small(const float* gy_cl, const float* in_nchw, float* ws, long long ws_bytes, float* gw_out,
                            float* gb_out_or_null, int N, int H, int W, int Cout, int Cin, int weight_channels_last,
                            spk_stream_t stream);

/* Training convolutions of the spiking VQ-VAE (R/snn_model/vae_model.py:101-159 under loss.backward(), R/main.py:118-146; cuDNN's
 * forward / data-gradient / weight-gradient kernels in the reference): channels-last fp32 tensors ([N][H][W][C]) on the fp32 matrix
 * cores (v_mfma_f32_32x32x2_f32: fp32 products and accumulation, no operand narrowed).  csrc/conv_train.hip.
 *
 * spk_conv_train_gather: out[N][Ho][Wo][Cout] (+ bias) from in[N][Hi][Wi][Cred] and a k x k weight tensor addressed through three
 * element strides, W(tap, c, co) = w[tap * w_tap + c * w_red + co * w_out]:
 *   form 0: out[n, o, co] = sum over (tap, c) of in[n, o * stride + k - pad, c] * W(tap, c, co)       -- layer.Conv2d forward
 *           (SJ/activation_based/layer.py:164-173), the data gradient of layer.ConvTranspose2d;
 *   form 1: out[n, o, co] = sum over the taps with (o + pad - k) % stride == 0 and c of in[n, (o + pad - k) / stride, c] *
 *           W(tap, c, co)  -- layer.ConvTranspose2d forward (layer.py:316-325), the data gradient of layer.Conv2d; sub-pixel
 *           classes are separate tile rows, structural zeros are not multiplied.
 * Matrix path: Cred % 8 == 0, Cred <= 64, 2 <= Cout <= 64, stride <= 4 (form 1: <= 2), the k * k weight taps of one 32-channel column
 * tile within 150 KB of LDS; vector kernels for Cred <= 4 (form 0, Cout % 4 == 0: the first layer on grey or RGB images, the read-out
 * layer's data gradient) and Cout <= 4 (Cred 8 / 16 / 32 / 64; form 1 only with stride 1: the read-out layer).  N * Ho * Wo and N * Hi * Wi * Cred below 2^31.  spk_conv_train_gather_supported answers 1 / 0; an unsupported call returns
 * SPK_ERR_UNSUPPORTED (the host then takes the framework's operator).  Deterministic; capturable in a hipGraph. */
int spk_conv_train_gather_supported(int Cred, int Cout, int k, int stride, int form);
int spk_conv_train_gather(const float* in_cl, const float* w, const float* bias_or_null, float* out_cl, int N, int Hi, int Wi,
                          int Cred, int Ho, int Wo, int Cout, int k, int stride, int pad, int form, long long w_tap,
                          long long w_red, long long w_out, spk_stream_t stream);
/* Weight (and bias) gradient of the same layers: D(tap, cu, cv) = sum over (n, q) of u[n, q * stride - pad + k, cu] * v[n, q, cv],
 * written to gw_out[tap * g_tap + cu * g_u + cv * g_v].  layer.Conv2d: u = the layer's input, v = gy; layer.ConvTranspose2d:
 * u = gy, v = the layer's input (u is the tensor on the finer grid).  bias_from: 0 none, 1: gb_out[cv] = column sums of v, 2:
 * gb_out[cu] = column sums of u.  Every workgroup owns a range of positions; a second launch adds the partial tiles in index order
 * (deterministic).  5 <= Cu <= 64, Cv <= 64 (k * k * ceil(Cu / 32) * ceil(Cv / 32) <= 36 tiles), or Cu <= 4 with Cv in {4, 8, 16, 32, 64}
 * (vector kernel; k <= 3 unless Cu == 1).  ws: spk_conv_train_wgrad_ws_bytes(...) bytes (-1: unsupported shape). */
long long spk_conv_train_wgrad_ws_bytes(int N, int Hv, int Wv, int Cu, int Cv, int k);
int spk_conv_train_wgrad(const float* u_cl, const float* v_cl, float* ws, long long ws_bytes, float* gw_out, float* gb_out_or_null,
                         int N, int Hu, int Wu, int Cu, int Hv, int Wv, int Cv, int k, int stride, int pad, long long g_tap,
                         long long g_u, long long g_v, int bias_from, spk_stream_t stream);

/* ---- sampler ---------------------------------------------------------------------------------------------------- */
/* Images touched by reverse step t.  R/snn_model/vq_diffusion.py:113-124 computes `changes = (u < 1/t) & ~unmasked`
 * BEFORE the denoiser call and only scatters the sample there (:140): for an image without a change at step t the
 * denoiser output is never read.  Writes the ascending list of images with >= 1 change (active_out [B] int32) and its
 * length (n_active_out [2] int32: [0] the length, [1] a work word that must be ZERO before the first call and is left zero);
 * u / Philox arguments exactly as spk_psample_step (same draws; K = the class count of that call: it is the stride of the
 * counter layout, see spk_psample_step).  With sample_steps = 100
 * and 49 positions an image is touched by 39 % of the steps on average: the per-step kernels take the list / count as
 * `active` / `n_dyn` arguments and skip the rest -- the same tokens as the dense loop, fewer evaluations. */
int spk_select_active(const uint8_t* unmasked, int t, const float* u_or_null, unsigned long long philox_seed,
                      unsigned long long philox_offset, const unsigned long long* philox_state_or_null, int* active_out,
                      int* n_active_out, int B, int HW, int K, spk_stream_t stream);
/* Positions of the active images that reverse step t needs from each denoiser layer (same `changes` test as above, per
 * position; R/snn_model/vq_diffusion.py:134-140 reads the denoiser output only there).  need_out: a ZERO-INITIALISED buffer of
 * spk_select_needed_bytes(B, R) bytes (it stays consistent from call to call); for every radius r = 1..R it receives, per
 * active slot, the positions within Chebyshev distance r of a change of image active[slot] -- what a stack of r 3x3
 * convolutions below the logits has to provide (64-byte records: bytes 0..47 ascending positions below 48, padded with the
 * last entry; byte 48 their number n; byte 49 ceil(n / 8); byte 50 whether position 48 is among them) -- and the slots
 * grouped by ceil(n / 8), the form spk_den_conv3x3_mfma_fp6v2_listed walks.  7x7 latents, R <= 8. */
long long spk_select_needed_bytes(int B, int R);
int spk_select_needed(const uint8_t* unmasked, int t, const float* u_or_null, unsigned long long philox_seed,
                      unsigned long long philox_offset, const unsigned long long* philox_state_or_null, const int* active,
                      const int* n_active, uint8_t* need_out, int B, int H, int W, int R, int K, spk_stream_t stream);
/* cat(x, ones_like(x)*t) of DummyModel.forward, R/snn_model/vq_diffusion.py:195-197 -> fp32 [B,2,h,w].
 * active / n_active (both or neither): slot s of the output is image active[s], s < *n_active. */
int spk_den_build_input(const float* x_float_or_null, const long long* x_tokens_or_null, const long long* t_vec_or_null,
                        long long t_scalar, float* out_b2hw, int B, int HW, const int* active_or_null,
                        const int* n_active_or_null, spk_stream_t stream);
/* Loop body of AbsorbingDiffusion.sample after the denoiser call, R/snn_model/vq_diffusion.py:113-124,134-140.
 * logits [B,K,h,w] fp32; x_t int64 [B*HW]; unmasked u8/bool [B*HW]; u [B*HW] / q [B*HW*K] injected noise or NULL
 * (then Philox4x32-10 keyed by seed: position p = image * HW + hw draws u from counter offset + p * K of stream 0 and q_k
 * from counter offset + p * K + k of stream 1 -- ONE rule for both, so a shard that starts at image b0 of a larger job
 * passes offset + b0 * HW * K and draws exactly what the whole job would draw for its images: the sample does not depend
 * on how a batch is split over devices); philox_state optional device {seed, base offset} pair that overrides
 * the seed and is added to the offset (lets a captured hipGraph draw fresh noise on every replay);
 * x0_hat_out optional int64 [B*HW].  active / n_active (both or neither; not with x0_hat_out): logits hold one slot
 * per active image (slot s = image active[s]); noise, x_t and unmasked stay indexed by image.
 * next_input_b2hw_or_null (not with active): also writes cat(x_t, t - 1) fp32 [B,2,h,w] -- what spk_den_build_input would
 * produce for the NEXT reverse step (:195-197 with the updated tokens): one launch less per step. */
int spk_psample_step(const float* logits_bkhw, long long* x_t_inout, uint8_t* unmasked_inout, int t, float temp,
                     const float* u_or_null, const float* q_or_null, unsigned long long philox_seed,
                     unsigned long long philox_offset, const unsigned long long* philox_state_or_null,
                     long long* x0_hat_out_or_null, int B, int HW, int K, const int* active_or_null,
                     const int* n_active_or_null, float* next_input_b2hw_or_null, spk_stream_t stream);

/* The tail of one DENSE reverse step as one launch per image: conv6 on the spike counts + mean over T (as
 * spk_den_conv3x3_counts_mfma; R/snn_model/vq_diffusion.py:185-187,205-206), the token update (as spk_psample_step: :113-124,
 * 134-140, same u / q / Philox arguments and draws) and -- unless x1_s32_out is NULL (last step) -- the first denoiser layer of
 * the NEXT step on cat(x_t, t - 1) (as spk_conv_fused_fwd on a time-invariant input: :161-165,195-201): S32 spikes
 * [B][2][HW][16][16 B] and spike counts u8 [B][2][HW][32].  cnt5 [B][8][HW][32] / cnt1 [B][2][HW][32]: spike counts of conv5 /
 * conv1 of THIS step; wq / scale / bias_d: conv6 packed by spk_den_pack_weight_i8 with its output channels zero-padded to
 * ceil16(K) (any --codebook_size, R/main.py:58: classes >= K are masked in the sampling exactly as spk_psample_step masks them);
 * conv1_w_packed: spk_pack_conv_weight of the first layer ([9][2][64]); bn1_a / bn1_b: its folded BatchNorm.  logits_out optional
 * fp32 [B][K][H][W].  1 <= K <= 512, 7x7 or 8x8 latents, 256 + 64 input channels (the reference's architecture); anything
 * else: SPK_ERR_UNSUPPORTED (use the three launches).
 * The fused first layer is the time-invariant-input form with the module defaults baked in: T = 16 steps and
 * LIFNode(tau = 2, v_threshold = 1, v_reset = 0) (R/snn_model/vq_diffusion.py:161-165); with x1_s32_out set any other T is
 * SPK_ERR_UNSUPPORTED (conv6 alone takes T <= 127).
 * active / n_active (round 6, both or neither): the active-set form of the untouched-image elimination (spk_select_active) -- workgroup s
 * serves slot s of the list: cnt5 / cnt1 / logits_out are indexed by slot (what the active-set denoiser launches produced), x_t / unmasked /
 * the noise by image active[s], so the draws are those of the dense form; x1_s32_out must be NULL there (the next step's first layer is
 * the next step's active set's). */
int spk_den_step_tail(const uint8_t* cnt5, int nch5, const uint8_t* cnt1, int nch1, const int8_t* wq, const double* scale,
                      const double* bias_d, float* logits_out_or_null, long long* x_t_inout, uint8_t* unmasked_inout, int t,
                      float temp, const float* u_or_null, const float* q_or_null, unsigned long long philox_seed,
                      unsigned long long philox_offset, const unsigned long long* philox_state_or_null,
                      const float* conv1_w_packed_or_null, const float* conv1_bias_or_null, const float* bn1_a,
                      const float* bn1_b, uint8_t* x1_s32_out_or_null, uint8_t* cnt1_out_or_null, int T, int B, int H, int W,
                      int K, const int* active_or_null, const int* n_active_or_null, spk_stream_t stream);
/* q_sample of the diffusion training step, R/snn_model/vq_diffusion.py:61-75: mask = u < t[b] / num_timesteps (fp32, as there);
 * x_t = mask ? mask_id : x_0;  x_0_ignore = mask ? x_0 : -1 (the loss's ignore index).  x0 / u / outputs fp32 [B*HW], t int64 [B],
 * mask_out optional u8.  u is the caller's draw (torch.rand_like in the reference's order). */
int spk_q_sample(const float* x0, const long long* t, const float* u, float* x_t_out, float* x0_ignore_out,
                 uint8_t* mask_out_or_null, int B, int HW, int num_timesteps, float mask_id, spk_stream_t stream);
/* The noise of one reverse step as spk_psample_step / spk_select_active / spk_select_needed draw it in Philox mode, written
 * out (parity aid: R/snn_model/vq_diffusion.py:116 `rand_like` -> u, :138 `Categorical.sample()` -> q ~ Exp(1)): u_out [B*HW] =
 * the uniforms of the `changes` test, q_out [B*HW*K] = the exponentials of the categorical race; same (seed, offset,
 * philox_state) arguments as the step they belong to.  The oracle run on these must reproduce the Philox-mode tokens. */
int spk_philox_noise(unsigned long long philox_seed, unsigned long long philox_offset,
                     const unsigned long long* philox_state_or_null, float* u_out_or_null, float* q_out_or_null, int B, int HW,
                     int K, spk_stream_t stream);

/* Content checksum of n device tensors in one launch (host-side cache validation; no reference counterpart: the reference
 * re-reads its weights on every call, this library keeps derived forms of them).  table_dev: device array of n pairs
 * {address, number of 32-bit words}; out1: one device word.  Order-independent 64-bit sum. */
int spk_checksum_multi(const unsigned long long* table_dev, int n, unsigned long long* out1, spk_stream_t stream);

/* ---- spike counts (syops report) -------------------------------------------------------------------------------- */
/* Spikes in a tensor the library emitted, all time steps and time step 0 alone -- the firing rates R/syops/ops.py:14-24
 * (`spike_rate`) and :69-75 (the LIF hook reads output[0]) feed into the ACs / MACs report.  The tensor is read as
 * n_words u32 words; word w belongs to time step (w / inner_words) % T.  kind 0: u8 {0,1} bytes (PTC / CPTC), 1: fp4
 * nibbles (C4 / S32), 2: fp32 ([T][N]).  out3 (device, 3 x u64): spikes, spikes at t = 0, fp32 words equal to 1.0f (kind 2:
 * nonzero == ones  <=>  the tensor is binary). */
int spk_count_spikes(const void* data, long long n_words, long long inner_words, int T, int kind, unsigned long long* out3,
                     spk_stream_t stream);

/* ---- measurement aid ------------------------------------------------------------------------------------------ */
/* Shader clock this device holds under a block-scaled fp6 x fp4 MFMA load (bench.py records it next to every
 * matrix-core number: devices of one pool differ by ~10 %).  nblocks workgroups of 256 threads issue 4*iters MFMAs per
 * wave between two {s_memtime, s_memrealtime} stamps; out [nblocks][4] u64 = {shader cycles, 100 MHz ticks, MFMAs
 * per wave, 0}.  No counterpart in the reference (it has no timing hooks on this path). */
int spk_clock_probe(unsigned long long* out, int nblocks, int iters, spk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPKDIFF_H */
