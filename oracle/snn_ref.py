"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A functional, torch-CPU restatement of the Spiking-Diffusion time-stepped SNN
inference path (spiking VQ-VAE encode/decode, spiking denoiser, absorbing-state
reverse-diffusion sampler).  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import this module, and only as
the checker / the reported CPU baseline.

Citation notation (same as SURVEY.md):
  R/<path>:<lines>  = /root/reference/Spiking-Diffusion-release/<path>
  SJ/<path>:<lines> = member <path> of R/spikingjelly.zip

Third-party arithmetic: convolution, batch-norm, matmul, argmin, softmax and
``exponential_`` live in PyTorch (the reference pins no version; this image has
torch 2.10.0+rocm7.0 here and on the GPU box).  The reference calls them at
SJ/activation_based/layer.py:164-173,316-325,458-465, R/snn_model/vae_model.py:89-94,99
and R/snn_model/vq_diffusion.py:136-138; this oracle calls the same ATen ops in
the same order, so on one machine its results are bit-identical to the
reference modules.

PARITY PINNING: the reference has no tests and no golden vectors.
This oracle is pinned by ``oracle/gen_golden.py``, which imports the real
reference in the build container, runs both on the same weights/inputs/seeds,
asserts bit-equality, and commits the reference's outputs as fixtures under
``tests/golden/`` (``tests/test_oracle_golden.py`` re-checks the oracle against
them).  T != 16 and latent != 7x7 cannot be run by the unmodified reference
(it raises); those shapes are "parity unpinned" and rely on this restatement.

Everything is parameterised by the state_dict ``sd`` (reference key names), the
number of time steps T (taken from the tensors) and the latent side.
"""
from __future__ import annotations

import numpy as np
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, SJ/activation_based/layer.py:423-436


# --------------------------------------------------------------------------- a1
def lif_multi_step(x_seq: torch.Tensor, v=0.0, v_threshold: float = 1.0, v_reset: float = 0.0,
                   tau: float = 2.0):
    """Eval-mode multi-step LIF, hard reset, decay_input=True.

    Follows ``jit_eval_multi_step_forward_hard_reset_decay_input``
    SJ/activation_based/neuron.py:799-811 (dispatch :971-1011); a python-float
    ``v`` is expanded like ``v_float_to_tensor`` :260-263.
    Returns (spike_seq, v_final).
    """
    if not torch.is_tensor(v):
        v = torch.full_like(x_seq[0], float(v))
    spike_seq = torch.zeros_like(x_seq)
    for t in range(x_seq.shape[0]):
        v = v + (x_seq[t] - (v - v_reset)) / tau
        spike = (v >= v_threshold).to(x_seq)
        v = v_reset * spike + (1.0 - spike) * v
        spike_seq[t] = spike
    return spike_seq, v


def lif_multi_step_ex(x_seq: torch.Tensor, v=0.0, v_threshold: float = 1.0, v_reset=0.0, tau: float = 2.0,
                      decay_input: bool = True):
    """The eval-mode multi-step LIF in all four forms of the reference, with the membrane potential after every step:
    ``jit_eval_multi_step_forward_{hard,soft}_reset_{decay,no_decay}_input_with_v_seq``,
    SJ/activation_based/neuron.py:813-900 (dispatch :971-1011; ``v_reset=None`` = soft reset).  The operations are the
    reference's, one for one.  Returns (spike_seq, v_final, v_seq)."""
    if not torch.is_tensor(v):
        v = torch.full_like(x_seq[0], float(v))
    spike_seq = torch.zeros_like(x_seq)
    v_seq = torch.zeros_like(x_seq)
    for t in range(x_seq.shape[0]):
        if v_reset is None:
            if decay_input:
                v = v + (x_seq[t] - v) / tau
            else:
                v = v * (1. - 1. / tau) + x_seq[t]
        else:
            if decay_input:
                v = v + (x_seq[t] - (v - v_reset)) / tau
            else:
                v = v - (v - v_reset) / tau + x_seq[t]
        spike = (v >= v_threshold).to(x_seq)
        if v_reset is None:
            v = v - spike * v_threshold
        else:
            v = v_reset * spike + (1. - spike) * v
        spike_seq[t] = spike
        v_seq[t] = v
    return spike_seq, v, v_seq


class _ATanSpike(torch.autograd.Function):
    """Heaviside forward, arc-tangent surrogate backward: SJ/activation_based/surrogate.py:664-678
    (``atan_backward``: alpha / 2 / (1 + (pi / 2 * alpha * x)^2) * grad_output)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return (x >= 0).to(x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return ctx.alpha / 2 / (1 + (math.pi / 2 * ctx.alpha * x).pow_(2)) * grad_output, None


def lif_multi_step_train(x_seq: torch.Tensor, v=0.0, v_threshold: float = 1.0, v_reset: float = 0.0,
                         tau: float = 2.0, alpha: float = 2.0, detach_reset: bool = False):
    """Training-mode multi-step LIF (torch backend of the reference): per step ``neuronal_charge`` (decay_input;
    SJ/activation_based/neuron.py:739-749: ``v + (x - v) / tau`` when v_reset == 0, else ``v + (x - (v - v_reset)) / tau``),
    ``neuronal_fire`` = surrogate(v - v_threshold), hard ``neuronal_reset`` (:133-135: ``(1 - spike_d) * v + spike_d *
    v_reset``, spike_d detached iff detach_reset).  Differentiable through torch autograd.  Returns (spike_seq, v_final).
    SURVEY.md §8f item 2."""
    if not torch.is_tensor(v):
        v = torch.full_like(x_seq[0], float(v))
    spikes = []
    for t in range(x_seq.shape[0]):
        if v_reset == 0.0:
            v = v + (x_seq[t] - v) / tau
        else:
            v = v + (x_seq[t] - (v - v_reset)) / tau
        spike = _ATanSpike.apply(v - v_threshold, alpha)
        spike_d = spike.detach() if detach_reset else spike
        v = (1.0 - spike_d) * v + spike_d * v_reset
        spikes.append(spike)
    return torch.stack(spikes), v


# --------------------------------------------------------------------------- a2
def seq_to_ann(x_seq: torch.Tensor, fn):
    """SJ/activation_based/functional.py:680-688: fold T into the batch, apply, unfold."""
    y = fn(x_seq.flatten(0, 1))
    return y.view(x_seq.shape[0], x_seq.shape[1], *y.shape[1:])


def seq_conv2d(x_seq, w, b, stride=1, padding=0):
    """layer.Conv2d in 'm' mode, SJ/activation_based/layer.py:164-173."""
    if x_seq.dim() != 5:
        raise ValueError(f"expected x with shape [T, N, C, H, W], but got x with shape {x_seq.shape}!")
    return seq_to_ann(x_seq, lambda y: F.conv2d(y, w, b, stride, padding))


def seq_conv_transpose2d(x_seq, w, b, stride=1, padding=0, output_padding=0):
    """layer.ConvTranspose2d in 'm' mode, SJ/activation_based/layer.py:316-325."""
    if x_seq.dim() != 5:
        raise ValueError(f"expected x with shape [T, N, C, H, W], but got x with shape {x_seq.shape}!")
    return seq_to_ann(x_seq, lambda y: F.conv_transpose2d(y, w, b, stride, padding, output_padding))


def seq_bn_eval(x_seq, sd, prefix):
    """layer.BatchNorm2d eval in 'm' mode, SJ/activation_based/layer.py:458-465."""
    if x_seq.dim() != 5:
        raise ValueError(f"expected x with shape [T, N, C, H, W], but got x with shape {x_seq.shape}!")
    return seq_to_ann(x_seq, lambda y: F.batch_norm(
        y, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS))


def bn_affine_terms(sd, prefix):
    """The fp32 form PyTorch-CPU eval BN evaluates, pinned bit-exactly by fixture F7
    (aten/native/cpu/batch_norm_kernel.cpp collect_linear_and_constant_terms + fmadd):
        a = (1 / sqrt(var + eps)) * gamma ;  b' = fma(-mean, a, beta) ;  y = fma(x, a, b').
    The fused multiply-adds are emulated through float64 (products of two fp32 are exact there)."""
    var, mean = sd[prefix + ".running_var"], sd[prefix + ".running_mean"]
    a = (1.0 / torch.sqrt(var + BN_EPS)) * sd[prefix + ".weight"]
    b = ((-mean).double() * a.double() + sd[prefix + ".bias"].double()).float()
    return a, b


def bn_apply_fma(x_seq, a, b):
    """y = fma(x, a[c], b[c]) over [T,B,C,H,W] (float64 emulation of the fp32 fma)."""
    c = a.numel()
    return (x_seq.double() * a.double().view(1, 1, c, 1, 1) + b.double().view(1, 1, c, 1, 1)).float()


def conv_bn_lif(x_seq, sd, conv_prefix, bn_prefix, stride, padding, transposed=False, output_padding=0, state=None):
    """One (Conv|ConvT) + BN + LIF block; returns (spikes, pre_activation).  LIF state: fresh, unless ``state`` (a dict)
    is given -- then the block's membrane potential is read from / left in ``state[bn_prefix]``, the way a module keeps
    ``v`` between two forwards that no ``reset_net`` separates (SJ/activation_based/base.py:277-343)."""
    if transposed:
        y = seq_conv_transpose2d(x_seq, sd[conv_prefix + ".weight"], sd[conv_prefix + ".bias"],
                                 stride, padding, output_padding)
    else:
        y = seq_conv2d(x_seq, sd[conv_prefix + ".weight"], sd[conv_prefix + ".bias"], stride, padding)
    y = seq_bn_eval(y, sd, bn_prefix)
    if state is None:
        s, _ = lif_multi_step(y)
    else:
        s, state[bn_prefix] = lif_multi_step(y, state.get(bn_prefix, 0.0))
    return s, y


# --------------------------------------------------------------------------- a4
def memout_coef(T: int) -> torch.Tensor:
    """R/snn_model/snn_layers.py:31-34 (n_steps generalised from the literal 16)."""
    return torch.pow(0.8, torch.arange(T - 1, -1, -1))[:, None, None, None, None]


def membrane_output(x_seq, coef=None):
    """MembraneOutputLayer.forward, R/snn_model/snn_layers.py:36-41."""
    if coef is None:
        coef = memout_coef(x_seq.shape[0])
    return torch.sum(x_seq * coef, dim=0)


# --------------------------------------------------------------------------- a3
def encoder_forward(x_seq, sd, return_layers=False, state=None):
    """Encoder.forward, R/snn_model/vae_model.py:101-129 (``state``: carried membrane potentials, see conv_bn_lif)."""
    p = "encoder.snn_convs."
    s1, y1 = conv_bn_lif(x_seq, sd, p + "0", p + "1", 2, 1, state=state)
    s2, y2 = conv_bn_lif(s1, sd, p + "3", p + "4", 2, 1, state=state)
    s3, y3 = conv_bn_lif(s2, sd, p + "6", p + "7", 1, 0, state=state)
    if return_layers:
        return s3, [(s1, y1), (s2, y2), (s3, y3)]
    return s3


# --------------------------------------------------------------------------- a5
def vq_readout(z_seq, sd):
    """x_memout of VectorQuantizer.forward, R/snn_model/vae_model.py:40-46 -> flat [B*h*w, D]."""
    T = z_seq.shape[0]
    alpha = sd["vq_layer.alpha"]
    x_memout = (1 - alpha) * membrane_output(z_seq, sd["vq_layer.memout.coef"]) + alpha * torch.sum(z_seq, dim=0) / T
    x_memout = x_memout.permute(0, 2, 3, 1).contiguous()
    return x_memout.reshape(-1, x_memout.shape[-1]), x_memout.shape


def vq_distances(flat_x, codebook):
    """get_code_indices distance expression, R/snn_model/vae_model.py:87-93."""
    return (torch.sum(flat_x ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2.0 * torch.matmul(flat_x, codebook.t()))


def vq_code_indices(flat_x, codebook):
    """R/snn_model/vae_model.py:94-95 (argmin: first index on ties)."""
    return torch.argmin(vq_distances(flat_x, codebook), dim=1)


def poisson_forward(q_bchw, sd, T):
    """'adaptive spike generator': repeat T, Conv1x1 + BN + LIF. R/snn_model/vae_model.py:34-38,54-57."""
    q = torch.unsqueeze(q_bchw, dim=0).repeat(T, 1, 1, 1, 1)
    p = "vq_layer.poisson."
    s, y = conv_bn_lif(q, sd, p + "0", p + "1", 1, 0)
    return s, y


def vq_forward(z_seq, sd):
    """VectorQuantizer.forward eval branch, R/snn_model/vae_model.py:40-58 -> (spikes, indices)."""
    flat_x, shp = vq_readout(z_seq, sd)
    idx = vq_code_indices(flat_x, sd["vq_layer.embeddings.weight"])
    quantized = F.embedding(idx, sd["vq_layer.embeddings.weight"]).view(shp)   # quantize(), :97-99
    quantized = quantized.permute(0, 3, 1, 2).contiguous()
    e, _ = poisson_forward(quantized, sd, z_seq.shape[0])
    return e, idx


# --------------------------------------------------------------------------- a6
def decoder_forward(e_seq, sd, return_layers=False):
    """Decoder.forward, R/snn_model/vae_model.py:131-159 -> pre-membrane output [T,B,C,H,W]."""
    p = "decoder.snn_convs."
    s1, y1 = conv_bn_lif(e_seq, sd, p + "0", p + "1", 2, 1, True, 1)
    s2, y2 = conv_bn_lif(s1, sd, p + "3", p + "4", 2, 1, True, 1)
    y3 = seq_conv_transpose2d(s2, sd[p + "6.weight"], sd[p + "6.bias"], 1, 1, 0)
    if return_layers:
        return y3, [(s1, y1), (s2, y2)]
    return y3


# --------------------------------------------------------------------------- a7
def snn_vqvae_forward(x_seq, sd):
    """SNN_VQVAE.forward eval branch, R/snn_model/vae_model.py:179-187 -> (e, x_recon, indices)."""
    z = encoder_forward(x_seq, sd)
    e, idx = vq_forward(z, sd)
    x_recon = torch.tanh(membrane_output(decoder_forward(e, sd), sd["memout.coef"]))
    return e, x_recon, idx


def encode_indices(images, sd, T=16, state=None):
    """Body of get_data_for_diff's loop, R/snn_model/vq_diffusion.py:27-34 (one batch, no .cuda()).

    The reference calls ``model(images_spike, images)`` for batch after batch with NO ``reset_net`` in between
    (:23-36; its loaders drop the last ragged batch, R/load_dataset_snn.py:65-66), so every LIF layer starts a batch
    from the membrane potentials the previous batch left.  Pass one ``state`` dict across the batches to reproduce that;
    ``state=None`` encodes from the reset state.  Only the encoder's state can reach the code indices (the spike
    generator's and the decoder's come after the code search)."""
    images = images - 0.5
    x_seq = images.unsqueeze(0).repeat(T, 1, 1, 1, 1)
    z = encoder_forward(x_seq, sd, state=state)
    flat, _ = vq_readout(z, sd)
    idx = vq_code_indices(flat, sd["vq_layer.embeddings.weight"])
    L = images.shape[-1] // 4
    return idx.reshape(images.shape[0], L, L)


def get_data_for_diff(batches, sd, T=16):
    """get_data_for_diff, R/snn_model/vq_diffusion.py:23-36: list of [B,h,w] index tensors, LIF state carried from
    batch to batch as the reference does (no reset inside the loop)."""
    state = {}
    return [encode_indices(images, sd, T, state) for images, _ in batches]


def decode_tokens(tokens_bhw, sd, T=16):
    """Sample->image glue of R/main.py:388-399: tokens [B,h,w] int64 -> pred [B,C,H,W] in (-1,1)."""
    z = F.embedding(tokens_bhw, sd["vq_layer.embeddings.weight"])          # quantize(): [B,h,w,D]
    z = z.permute(0, 3, 1, 2).contiguous()
    e, _ = poisson_forward(z, sd, T)
    return torch.tanh(membrane_output(decoder_forward(e, sd), sd["memout.coef"]))


def to_uint8(pred):
    """R/main.py:401: np.array(np.clip((pred + 0.5).cpu().numpy(), 0., 1.) * 255, dtype=np.uint8)."""
    return np.array(np.clip((pred + 0.5).cpu().numpy(), 0.0, 1.0) * 255, dtype=np.uint8)


# --------------------------------------------------------------------------- a8
def conv_bn_lif_exact(x_seq, sd, conv_prefix, bn_prefix, stride, padding):
    """The block of ``conv_bn_lif`` with the convolution evaluated EXACTLY: fp64 accumulation of the fp32 products (exact to
    2^-53 of the sum), bias added in fp64, ONE rounding to fp32 -- the arithmetic contract of the HIP kernels --
    followed by the reference's fp32 BatchNorm (its fma form, fixture F7) and LIF operations.  The reference's own convolution
    (oneDNN, fp32 accumulation in an unknowable order) approximates this value to a few ulp; where a membrane potential sits
    that close to the threshold the two can decide a spike differently.  Test infrastructure: it tells the two apart."""
    w, b = sd[conv_prefix + ".weight"].double(), sd[conv_prefix + ".bias"].double()
    y = seq_to_ann(x_seq.double(), lambda v: F.conv2d(v, w, b, stride, padding)).float()
    a, bb = bn_affine_terms(sd, bn_prefix)
    y = bn_apply_fma(y, a, bb)
    s, _ = lif_multi_step(y)
    return s, y


def denoiser_forward(x, t, sd, T=16, return_layers=False, exact_conv=False):
    """DummyModel.forward, R/snn_model/vq_diffusion.py:189-208. x [B,1,h,w] float, t [B] long.
    ``exact_conv``: every convolution as the correctly rounded exact dot product (conv_bn_lif_exact) instead of oneDNN's fp32."""
    tt = torch.ones_like(x) * (t.unsqueeze(1).unsqueeze(2).unsqueeze(3))
    x = torch.cat((x, tt), dim=1)
    x = x.unsqueeze(dim=0).repeat(T, 1, 1, 1, 1)
    layers = []
    h = x
    outs = []
    for i in range(1, 6):
        s, y = (conv_bn_lif_exact if exact_conv else conv_bn_lif)(h, sd, f"conv{i}.0", f"conv{i}.1", 1, 1)
        layers.append((s, y))
        outs.append(s)
        h = s
    cat = torch.cat((outs[4], outs[0]), dim=2)
    if exact_conv:
        x6 = seq_to_ann(cat.double(), lambda v: F.conv2d(v, sd["conv6.0.weight"].double(), sd["conv6.0.bias"].double(), 1, 1))
        logits = (torch.sum(x6, dim=0) / T).float()
    else:
        x6 = seq_conv2d(cat, sd["conv6.0.weight"], sd["conv6.0.bias"], 1, 1)
        logits = torch.sum(x6, dim=0) / T
    if return_layers:
        return logits, layers
    return logits


# --------------------------------------------------------------------------- §8f item 2: diffusion training step
def seq_bn_train(x_seq, sd, prefix, momentum=0.1, stats_out=None):
    """layer.BatchNorm2d in 'm' mode, TRAINING: T folded into the batch, ``F.batch_norm(..., training=True)``
    (SJ/activation_based/layer.py:458-465 -> torch.nn.BatchNorm2d.forward).  Batch statistics over (T*B, H, W);
    the running statistics are updated on copies returned through ``stats_out`` (a dict), never in ``sd``."""
    rm = sd[prefix + ".running_mean"].detach().clone()
    rv = sd[prefix + ".running_var"].detach().clone()
    y = seq_to_ann(x_seq, lambda z: F.batch_norm(z, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], True,
                                                 momentum, BN_EPS))
    if stats_out is not None:
        stats_out[prefix + ".running_mean"] = rm
        stats_out[prefix + ".running_var"] = rv
    return y


def denoiser_forward_train(x, t, sd, T=16, alpha=2.0, stats_out=None, return_layers=False):
    """DummyModel.forward in train() mode, R/snn_model/vq_diffusion.py:189-208: same graph as ``denoiser_forward``
    with batch-statistics BN and the surrogate-gradient LIF; differentiable w.r.t. the tensors of ``sd``."""
    tt = torch.ones_like(x) * (t.unsqueeze(1).unsqueeze(2).unsqueeze(3))
    h = torch.cat((x, tt), dim=1).unsqueeze(dim=0).repeat(T, 1, 1, 1, 1)
    outs = []
    for i in range(1, 6):
        y = seq_conv2d(h, sd[f"conv{i}.0.weight"], sd[f"conv{i}.0.bias"], 1, 1)
        y = seq_bn_train(y, sd, f"conv{i}.1", stats_out=stats_out)
        h, _ = lif_multi_step_train(y, alpha=alpha)
        outs.append(h)
    x6 = seq_conv2d(torch.cat((outs[4], outs[0]), dim=2), sd["conv6.0.weight"], sd["conv6.0.bias"], 1, 1)
    logits = torch.sum(x6, dim=0) / T
    if return_layers:
        return logits, outs
    return logits


def sample_time(b, num_timesteps):
    """AbsorbingDiffusion.sample_time, R/snn_model/vq_diffusion.py:56-59 (global generator: one randint)."""
    t = torch.randint(1, num_timesteps + 1, (b,)).long()
    pt = torch.ones_like(t).float() / num_timesteps
    return t, pt


def q_sample(x_0, t, num_timesteps, mask_id, u=None):
    """AbsorbingDiffusion.q_sample, R/snn_model/vq_diffusion.py:61-74: mask each token with probability t/T.
    ``u`` (uniforms shaped like x_0) may be injected, otherwise one ``rand_like`` draw from the global generator."""
    x_t, x_0_ignore = x_0.clone(), x_0.clone()
    t_mask = t.reshape(x_0.shape[0], 1, 1, 1).expand(x_0.shape[0], 1, x_0.shape[2], x_0.shape[3])
    if u is None:
        u = torch.rand_like(x_t.float())
    mask = u < (t_mask.float() / num_timesteps)
    x_t[mask] = mask_id
    x_0_ignore[torch.bitwise_not(mask)] = -1
    return x_t, x_0_ignore, mask


def masked_ce_loss(logits, x_0_ignore, t, num_timesteps, loss_type="reweighted_elbo"):
    """Loss tail of AbsorbingDiffusion._train_loss, R/snn_model/vq_diffusion.py:85-101.  logits [B,K,h,w]."""
    b, K = logits.shape[0], logits.shape[1]
    hw = logits.shape[2] * logits.shape[3]
    ce = F.cross_entropy(logits.reshape(b, K, hw), x_0_ignore.reshape(b, hw).long(), ignore_index=-1,
                         reduction='none').sum(1)
    denom = math.log(2) * x_0_ignore.shape[1:].numel()
    if loss_type == 'elbo':
        pt = torch.ones_like(t).float() / num_timesteps
        loss = ce / t / pt / denom
    elif loss_type == 'reweighted_elbo':
        loss = (1 - (t / num_timesteps)) * ce / denom
    else:
        raise ValueError
    return loss.mean()


def train_loss(x_0, sd, mask_id, num_timesteps=None, T=16, t=None, u=None, stats_out=None):
    """AbsorbingDiffusion._train_loss, R/snn_model/vq_diffusion.py:77-101.  RNG order: sample_time's randint,
    then q_sample's rand_like (global generator) unless ``t`` / ``u`` are injected."""
    if num_timesteps is None:
        num_timesteps = x_0.shape[2] * x_0.shape[3]
    if t is None:
        t, _ = sample_time(x_0.size(0), num_timesteps)
    x_t, x_0_ignore, mask = q_sample(x_0, t, num_timesteps, mask_id, u)
    logits = denoiser_forward_train(x_t, t, sd, T, stats_out=stats_out)
    return masked_ce_loss(logits, x_0_ignore, t, num_timesteps), (t, x_t, x_0_ignore, mask, logits)


def psp_filter(inputs, tau_s=2):
    """PSP.forward, R/snn_model/snn_layers.py:12-26: syn_t = syn_{t-1} + (inputs[t] - syn_{t-1}) / tau_s, stacked over t."""
    syn = 0
    syns = []
    for t in range(inputs.shape[0]):
        syn = syn + (inputs[t, ...] - syn) / tau_s
        syns.append(syn)
    return torch.stack(syns)


def conv_bn_lif_train(x_seq, sd, conv_prefix, bn_prefix, stride, padding, transposed=False, output_padding=0,
                      stats_out=None):
    """One (Conv|ConvT)+BN+LIF block in train() mode: library convolution, batch-statistics BN, surrogate-gradient LIF."""
    w, b = sd[conv_prefix + ".weight"], sd[conv_prefix + ".bias"]
    if transposed:
        y = seq_conv_transpose2d(x_seq, w, b, stride, padding, output_padding)
    else:
        y = seq_conv2d(x_seq, w, b, stride, padding)
    s, _ = lif_multi_step_train(seq_bn_train(y, sd, bn_prefix, stats_out=stats_out))
    return s


def snn_vqvae_train_forward(x_seq, image, sd, data_variance, commitment_cost=0.25, stats_out=None):
    """SNN_VQVAE.forward in train() mode, R/snn_model/vae_model.py:179-196 with VectorQuantizer.forward's training branch
    (:40-47,61-85), Encoder (:101-129) and Decoder (:131-159).  Returns (e_q_loss, recon_loss, real_recon_loss) and
    the code indices; differentiable w.r.t. the floating-point tensors of ``sd``."""
    T = x_seq.shape[0]
    p = "encoder.snn_convs."
    z = conv_bn_lif_train(x_seq, sd, p + "0", p + "1", 2, 1, stats_out=stats_out)
    z = conv_bn_lif_train(z, sd, p + "3", p + "4", 2, 1, stats_out=stats_out)
    z = conv_bn_lif_train(z, sd, p + "6", p + "7", 1, 0, stats_out=stats_out)
    alpha = sd["vq_layer.alpha"]
    x_memout = (1 - alpha) * membrane_output(z, sd["vq_layer.memout.coef"]) + alpha * torch.sum(z, dim=0) / T
    x_memout = x_memout.permute(0, 2, 3, 1).contiguous()
    flat_x = x_memout.reshape(-1, x_memout.shape[-1])
    idx = vq_code_indices(flat_x, sd["vq_layer.embeddings.weight"])
    quantized = F.embedding(idx, sd["vq_layer.embeddings.weight"]).view_as(x_memout)
    loss_1 = F.mse_loss(quantized, x_memout.detach()) + commitment_cost * F.mse_loss(x_memout, quantized.detach())
    quantized = x_memout + (quantized - x_memout).detach()
    quantized = quantized.permute(0, 3, 1, 2).contiguous().unsqueeze(0).repeat(T, 1, 1, 1, 1)
    e = conv_bn_lif_train(quantized, sd, "vq_layer.poisson.0", "vq_layer.poisson.1", 1, 0, stats_out=stats_out)
    q2 = torch.mean((psp_filter(e) - psp_filter(z.detach())) ** 2)
    e2 = torch.mean((psp_filter(e.detach()) - psp_filter(z)) ** 2)
    e_q_loss = loss_1 + (q2 + commitment_cost * e2)
    p = "decoder.snn_convs."
    d = conv_bn_lif_train(e, sd, p + "0", p + "1", 2, 1, True, 1, stats_out=stats_out)
    d = conv_bn_lif_train(d, sd, p + "3", p + "4", 2, 1, True, 1, stats_out=stats_out)
    y3 = seq_conv_transpose2d(d, sd[p + "6.weight"], sd[p + "6.bias"], 1, 1, 0)
    x_recon = torch.tanh(membrane_output(y3, sd["memout.coef"]))
    real_recon_loss = F.mse_loss(x_recon, image)
    return (e_q_loss, real_recon_loss / data_variance, real_recon_loss), idx


# --------------------------------------------------------------------------- a9
def categorical_sample(logits, q=None):
    """``dists.Categorical(logits=l).sample()`` as torch evaluates it on CPU:
    logits - logsumexp -> softmax -> torch.multinomial(probs, 1, True), whose one-draw fast path is
    argmax(probs / q), q ~ Exp(1) drawn row-major over [rows, K] (SURVEY.md §3.2 'RNG facts').
    ``q`` may be injected ([rows, K] fp32); otherwise it is drawn from the global CPU generator."""
    ln = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = F.softmax(ln, dim=-1)
    p2 = probs.reshape(-1, probs.shape[-1])
    if q is None:
        q = torch.empty_like(p2).exponential_(1)
    return torch.argmax(p2 / q, dim=-1).reshape(probs.shape[:-1])


def p_sample_step(x_t, unmasked, logits_bhwk, t, temp=1.0, u=None, q=None):
    """One iteration of the loop body of AbsorbingDiffusion.sample,
    R/snn_model/vq_diffusion.py:111-140, given the denoiser logits [B,h,w,K].
    ``u`` [B,1,h,w] uniforms and ``q`` [B*h*w,K] exponentials may be injected."""
    if u is None:
        u = torch.rand_like(x_t.float())
    t_mask = torch.full_like(u, float(t))             # t.reshape(b,1,1,1).expand(...).float(), :114-116
    changes = u < 1 / t_mask
    changes = torch.bitwise_xor(changes, torch.bitwise_and(changes, unmasked))
    unmasked = torch.bitwise_or(unmasked, changes)
    x_0_hat = categorical_sample(logits_bhwk / temp, q).long().unsqueeze(dim=1)
    x_t = x_t.clone()
    x_t[changes] = x_0_hat[changes]
    return x_t, unmasked


def absorbing_sample(sd, n_samples, mask_id, temp=1.0, sample_steps=49, latent=7, T=16,
                     noise=None, record=None, exact_conv=False):
    """AbsorbingDiffusion.sample, R/snn_model/vq_diffusion.py:103-142 (device literal dropped).

    RNG consumption order per step (probe-verified): B*h*w uniforms, then
    B*h*w*K exponentials, both from the global CPU generator unless ``noise`` (a callable
    step -> (u, q)) injects them.  ``record`` (a list) receives (t, x_t, unmasked, logits).
    ``exact_conv``: the denoiser's convolutions as correctly rounded exact dot products (``denoiser_forward``)."""
    b = int(n_samples)
    x_t = torch.ones(b, 1, latent, latent).long() * mask_id
    unmasked = torch.zeros_like(x_t).bool()
    for t in reversed(range(1, sample_steps + 1)):
        tt = torch.full((b,), t, dtype=torch.long)
        if noise is None:
            u = torch.rand_like(x_t.float())          # drawn BEFORE the denoiser call (:116)
        else:
            u, q_inj = noise(t)
        logits = denoiser_forward(x_t.float(), tt, sd, T, exact_conv=exact_conv).permute(0, 2, 3, 1)
        x_t, unmasked = p_sample_step(x_t, unmasked, logits, t, temp, u, None if noise is None else q_inj)
        if record is not None:
            record.append((t, x_t.clone(), unmasked.clone(), logits.clone()))
    return x_t


def sample_images(sd_vae, sd_den, n_samples, mask_id=128, temp=1.0, sample_steps=100, latent=7, T=16,
                  noise=None):
    """Full BASELINE path: sample tokens, decode (R/main.py:384-401) -> uint8 [B,C,H,W]."""
    tok = absorbing_sample(sd_den, n_samples, mask_id, temp, sample_steps, latent, T, noise)
    return to_uint8(decode_tokens(tok.reshape(n_samples, latent, latent), sd_vae, T)), tok
