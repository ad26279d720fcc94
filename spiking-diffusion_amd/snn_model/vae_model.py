"""``snn_model.vae_model`` of the MI355X build: the spiking VQ-VAE inference path.

Counterpart of R/snn_model/vae_model.py:22-196 -- same class names, constructor signatures, attribute paths
(``model.encoder``, ``model.vq_layer.quantize/poisson``, ``model.decoder``, ``model.memout``) and ``state_dict``
keys, so R/main.py's test/sampling section runs against it unchanged.  The eval branches run on
``libspkdiff.so`` (fused Conv+BN+LIF kernels, VQ argmin kernel).  The training branches (VQ / commitment / PSP /
reconstruction losses, straight-through estimator: vae_model.py:61-85,189-196) run in train()
mode with autograd: native convolutions forward and backward (``csrc/conv_train.hip``, fp32 matrix cores; shapes it does not
cover take the framework's operator), native BatchNorm+LIF block tails (``spk_bn_lif_train_*``), and the VectorQuantizer's
read-out / code search / losses, the PSP losses and the reconstruction loss as fused operators (``csrc/vq_train.hip``).

Re-exported names match what ``from snn_model.vae_model import *`` gives R/main.py (``functional`` in particular,
R/main.py:101-107,317).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from spikingjelly.activation_based import neuron, functional, layer, surrogate, monitor  # noqa: F401
from spikingjelly import visualizing  # noqa: F401

from spkdiff import ops
from spkdiff.fused import FusedSequential, has_hooks, derived_epoch
from spkdiff.ops import IN_PTC, IN_SEQ, IN_TINV

from .snn_layers import *  # noqa: F401,F403
from .snn_layers import MembraneOutputLayer, PSP


def _training_oos(what):
    raise NotImplementedError(f'spkdiff: {what} is a training branch of the reference, outside the inference hot '
                              'path (SURVEY.md §8f); call .eval()')


class VectorQuantizer(nn.Module):
    def __init__(self, embedding_dim, num_embeddings, commitment_cost, n_steps: int = 16):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.commitment_cost = commitment_cost
        self.memout = MembraneOutputLayer(n_steps)
        self.num_step = n_steps
        self.psp = PSP()
        self.alpha = nn.Parameter(torch.tensor(0.5))
        self.embeddings = nn.Embedding(self.num_embeddings, self.embedding_dim)
        self.fused_train = True              # training branch: ops.VQTrainFunction (False: the same algebra op by op through autograd)
        self.poisson = FusedSequential(
            layer.Conv2d(in_channels=embedding_dim, out_channels=embedding_dim, kernel_size=1),
            layer.BatchNorm2d(embedding_dim),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),
        )

    # -- fused pieces, shared with SNN_VQVAE's end-to-end path ------------------------------------------------
    def _quantize_ptc(self, z_ptc, want_xm=False, want_zq=True):
        """z_ptc u8 [B,h,w,T,D] -> (indices int64 [B*h*w], quantized fp32 [B,D,h,w] or None)."""
        idx, zq, xm = ops.vq_readout_argmin(z_ptc, self.memout.coef.flatten(), self.alpha, self.embeddings.weight,
                                            want_zq=want_zq, want_xm=want_xm)
        return (idx, zq, xm) if want_xm else (idx, zq)

    def _spike_generator(self, zq, T, final='f32'):
        """'adaptive spike generator': repeat(T) + poisson, the repeat folded into the kernel (same frame each step)."""
        return self.poisson.run(zq, IN_TINV, final=final, T=T)

    def forward(self, x):
        # x: (T,N,C,H,W) spikes of the encoder
        if self.training:
            return self._train_forward(x)
        T = x.shape[0]
        idx, zq = self._quantize_ptc(ops.spikes_to_ptc(x))
        if has_hooks(self.poisson):                   # hooks on the spike generator's layers: run it child by child
            return self.poisson(torch.unsqueeze(zq, dim=0).repeat(T, 1, 1, 1, 1)), idx
        quantized = self._spike_generator(zq, T)['f32']
        return quantized, idx

    def _train_forward(self, x):
        """Training branch, R/snn_model/vae_model.py:40-47,61-85: read-out, nearest code, VQ and
        commitment losses, straight-through estimator, spike generator, PSP losses.  Native pieces: the membrane read-out
        (spk_memout_fwd), the code search (spk_vq_argmin), the spike generator's BatchNorm+LIF (spk_bn_lif_train_*) and
        the PSP filter (spk_psp); the remaining element-wise algebra and the embedding gradient are torch plumbing."""
        if not torch.is_grad_enabled():
            _training_oos('VectorQuantizer.forward in train() mode without autograd')
        T = x.shape[0]
        if (self.fused_train and x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and T == self.memout.coef.numel()
                and self.embedding_dim <= ops.VQ_TRAIN_MAX_D):       # (spk_vq_train_bwd keeps a code vector in registers: D <= 64)
            # read-out, code search, q / e latent losses, straight-through value: three native launches forward, one backward
            # (ops.VQTrainFunction; the module-by-module algebra below is the same arithmetic through autograd)
            quantized, loss_1 = ops.VQTrainFunction.apply(x, self.memout.coef, self.alpha, self.embeddings.weight,
                                                          self.commitment_cost)
        else:
            x_memout = (1 - self.alpha) * self.memout(x) + self.alpha * torch.sum(x, dim=0) / self.num_step
            x_memout = x_memout.permute(0, 2, 3, 1).contiguous()
            flat_x = x_memout.reshape(-1, self.embedding_dim)
            encoding_indices = self.get_code_indices(flat_x.detach())
            quantized = F.embedding(encoding_indices, self.embeddings.weight).view_as(x_memout)
            q_latent_loss = F.mse_loss(quantized, x_memout.detach())
            e_latent_loss = F.mse_loss(x_memout, quantized.detach())
            loss_1 = q_latent_loss + self.commitment_cost * e_latent_loss
            quantized = x_memout + (quantized - x_memout).detach()           # straight-through estimator
            quantized = quantized.permute(0, 3, 1, 2).contiguous()
        quantized = torch.unsqueeze(quantized, dim=0).repeat(T, 1, 1, 1, 1)
        quantized = self.poisson(quantized)
        if self.fused_train and quantized.is_cuda and quantized.shape == x.shape and T <= 16:
            # both PSP filters, both mean squares and their backward: one launch each way (ops.PSPLossFunction)
            loss_2 = ops.PSPLossFunction.apply(quantized, x, self.commitment_cost, float(self.psp.tau_s))
        else:
            # psp(x.detach()) and psp(x).detach() are the same numbers: each filter runs once
            pq, px = self.psp(quantized), self.psp(x)
            q_latent_loss_2 = torch.mean((pq - px.detach()) ** 2)
            e_latent_loss_2 = torch.mean((pq.detach() - px) ** 2)
            loss_2 = q_latent_loss_2 + self.commitment_cost * e_latent_loss_2
        return quantized, loss_1 + loss_2

    def get_code_indices(self, flat_x):
        """argmin_k ||x - e_k||^2 for rows of flat_x [N, D] (R/snn_model/vae_model.py:87-95)."""
        return ops.vq_argmin(flat_x, self.embeddings.weight)

    def quantize(self, encoding_indices):
        """Returns embedding tensor for a batch of indices."""
        return ops.embedding(encoding_indices, self.embeddings.weight)


class Encoder(nn.Module):
    """Encoder of VQ-VAE"""

    def __init__(self, in_dim=1, latent_dim=16):
        super().__init__()
        self.in_dim = in_dim
        self.latent_dim = latent_dim
        self.snn_convs = FusedSequential(
            layer.Conv2d(in_channels=in_dim, out_channels=32, kernel_size=3, stride=2, padding=1),
            layer.BatchNorm2d(32),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),

            layer.Conv2d(in_channels=32, out_channels=64, kernel_size=3, stride=2, padding=1),
            layer.BatchNorm2d(64),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),

            layer.Conv2d(in_channels=64, out_channels=latent_dim, kernel_size=1, stride=1, padding=0),
            layer.BatchNorm2d(latent_dim),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),
        )

    def forward(self, x):
        # [t, b, c, h, w]
        return self.snn_convs(x)


class Decoder(nn.Module):
    """Decoder of VQ-VAE"""

    def __init__(self, out_dim=1, latent_dim=16):
        super().__init__()
        self.out_dim = out_dim
        self.latent_dim = latent_dim
        self.snn_convs = FusedSequential(
            layer.ConvTranspose2d(in_channels=latent_dim, out_channels=64, kernel_size=3, stride=2, padding=1,
                                  output_padding=1),
            layer.BatchNorm2d(64),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),

            layer.ConvTranspose2d(in_channels=64, out_channels=32, kernel_size=3, stride=2, padding=1,
                                  output_padding=1),
            layer.BatchNorm2d(32),
            neuron.LIFNode(surrogate_function=surrogate.ATan()),

            layer.ConvTranspose2d(in_channels=32, out_channels=out_dim, kernel_size=3, stride=1, padding=1,
                                  output_padding=0),
        )

    def forward(self, x):
        # [t, b, c, h, w]
        return self.snn_convs(x)


SPIKEGEN_BY_TOKEN = True     # decode_tokens: the spike generator as a per-token pattern table (False: embedding + generator + packing launches)


class SNN_VQVAE(nn.Module):
    """VQ-VAE"""

    def __init__(self, in_dim, embedding_dim, num_embeddings, data_variance, commitment_cost=0.25,
                 n_steps: int = 16):
        super().__init__()
        self.in_dim = in_dim
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        self.data_variance = data_variance

        self.encoder = Encoder(in_dim, embedding_dim)
        self.vq_layer = VectorQuantizer(embedding_dim, num_embeddings, commitment_cost, n_steps)
        self.decoder = Decoder(in_dim, embedding_dim)
        self.memout = MembraneOutputLayer(n_steps)

    def forward(self, x, image):
        # x: [t, B, C, H, W]
        if self.training:
            # training branch, R/snn_model/vae_model.py:189-196
            if not torch.is_grad_enabled():
                _training_oos('SNN_VQVAE.forward in train() mode without autograd')
            z = self.encoder(x)
            e, e_q_loss = self.vq_layer(z)
            y = self.decoder(e)
            if self.vq_layer.fused_train and y.is_cuda and y.shape[1:] == image.shape and image.dtype == torch.float32:
                real_recon_loss = ops.ReconLossFunction.apply(y, self.memout.coef, image)      # read-out + tanh + mse: one launch
            else:
                x_recon = torch.tanh(self.memout(y))
                real_recon_loss = F.mse_loss(x_recon, image)
            return e_q_loss, real_recon_loss / self.data_variance, real_recon_loss
        T = x.shape[0]
        enc = self.encoder.snn_convs
        dec = self.decoder.snn_convs
        if enc._fusable(enc._blocks()) and dec._fusable(dec._blocks()) and T <= ops.MAX_T and not has_hooks(self):
            # end-to-end fused: spikes stay u8 PTC between encoder, VQ, spike generator and decoder
            z_ptc = enc.run(x, IN_SEQ, final='ptc')['ptc']
            idx, zq = self.vq_layer._quantize_ptc(z_ptc)
            e = self.vq_layer._spike_generator(zq, T, final='both')
            x_recon = dec.run(e['ptc'], IN_PTC, final='memout', coef=self.memout.coef.flatten(),
                              apply_tanh=True)['f32']
            return e['f32'], x_recon, idx
        z = self.encoder(x)
        e, enco = self.vq_layer(z)
        x_recon = self.decoder(e)
        x_recon = torch.tanh(self.memout(x_recon))
        return e, x_recon, enco

    # ---- convenience entry points of the MI355X build (not in the reference) -----------------------------------
    @torch.no_grad()
    def encode_images(self, images, T=16):
        """images [B,C,H,W] already normalised (images - 0.5) -> code indices [B,h,w]; the T-fold repeat of
        R/main.py:309 / vq_diffusion.py:30 is folded into the first kernel (time-invariant input)."""
        z_ptc = self.encoder.snn_convs.run(images, IN_TINV, final='ptc', T=T, stateful=False)['ptc']
        idx, _ = self.vq_layer._quantize_ptc(z_ptc, want_zq=False)      # (indices only: no [B,D,h,w] gather)
        L = images.shape[-1] // 4
        return idx.reshape(images.shape[0], L, L)

    def _decoder_takes_s32(self, T, h, w):
        """Will the decoder's first layer take nibble-packed spikes (the fp6 transposed-convolution kernel, csrc/vae_fp6.hip)?"""
        blocks = self.decoder.snn_convs._blocks()
        if blocks is None or len(blocks) < 3 or T != 16:
            return False
        conv = blocks[-3][0]
        from spkdiff.fused import conv_geometry, has_hooks
        if has_hooks(conv) or len(blocks) != 3:
            return False
        geo = conv_geometry(conv)
        return ops.vae_fp6_kind(conv.in_channels, conv.out_channels, geo['k'], geo['stride'], geo['pad'], geo['out_pad'],
                                geo['transposed'], T, h, w) == ops.VAE_OUT_S32

    @torch.no_grad()
    def decode_tokens(self, tokens, T=16, want_u8=True):
        """tokens int64 [B,h,w] -> (pred fp32 [B,C,H,W] in (-1,1), uint8 image): the glue of R/main.py:388-401 as
        three launches (embedding gather, spike generator, fused decoder + read-out + tanh + uint8)."""
        B, h, w = tokens.shape
        e_ptc = None
        if SPIKEGEN_BY_TOKEN and self._decoder_takes_s32(T, h, w):
            # embedding + spike generator + nibble packing as a per-token pattern table (csrc/conv_direct.hip, spk_spikegen_tokens_s32)
            e_ptc = self.vq_layer.poisson.tokens_to_s32(tokens, self.vq_layer.embeddings.weight, T=T,
                                                        epoch=derived_epoch(self.vq_layer))
        if e_ptc is None:
            zq = ops.embedding(tokens, self.vq_layer.embeddings.weight, nchw_hw=(h, w))
            e_ptc = self.vq_layer.poisson.run(zq, IN_TINV, final='ptc', T=T, stateful=False)['ptc']
        r = self.decoder.snn_convs.run(e_ptc, IN_PTC, final='memout', coef=self.memout.coef.flatten(), apply_tanh=True,
                                       want_u8=want_u8, stateful=False)
        return r['f32'], r['u8']


def _not_in_scope(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f'spkdiff: {name} is a baseline model outside the named hot path '
                                      '(SURVEY.md §2.1 #3); only SNN_VQVAE is implemented')
    _Stub.__name__ = name
    return _Stub


SNN_VAE = _not_in_scope('SNN_VAE')
VQVAE = _not_in_scope('VQVAE')
SNN_VQVAE_uni = _not_in_scope('SNN_VQVAE_uni')
