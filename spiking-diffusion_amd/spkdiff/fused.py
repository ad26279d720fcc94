"""Layer fusion for the drop-in modules: a ``(Conv2d|ConvTranspose2d) -> BatchNorm2d -> LIFNode`` triple of an
``nn.Sequential`` becomes ONE kernel launch (``spk_conv_fused_fwd``), spikes travel between blocks as u8 "PTC"
tensors, and fp32 [T,B,C,H,W] tensors (the reference's format) are only materialised at the module boundary.

``FusedSequential`` is an ``nn.Sequential`` (same children, same ``state_dict`` keys as the reference's
``snn_convs`` / ``poisson`` / ``convN`` containers: R/snn_model/vae_model.py:34-38,109-124,139-155,
R/snn_model/vq_diffusion.py:161-187).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from spikingjelly.activation_based import layer, neuron, surrogate

from . import ops
from .ops import IN_PTC, IN_SEQ, IN_TINV, MODE_LIF, MODE_MEAN, MODE_MEMOUT, MODE_RAW


def _ver(t):
    return None if t is None else (t.data_ptr(), t._version, str(t.device))


class ConvParams:
    """Prepared weights of one conv layer, rebuilt when the parameters change:
    ``get``  -> fp32 packed [k*k][Cin][Cout] for the direct kernels;
    ``get_i8`` -> int8 digit planes + fp64 scale / bias for the MFMA kernel (built on first use)."""

    def __init__(self):
        self.key = None
        self.w_packed = None
        self.key_i8 = None
        self.i8 = None

    def invalidate(self):
        """Forget every derived form (call after writing weights through ``.data``: that does not bump ``_version``)."""
        self.key = self.key_i8 = None
        self.key_fp6 = self.key_i8g = self.key_fp6v2 = self.key_convT_fp6 = None

    def get(self, conv):
        key = (_ver(conv.weight), _ver(conv.bias))
        if key != self.key:
            transposed = isinstance(conv, nn.ConvTranspose2d)
            self.w_packed = ops.pack_conv_weight(conv.weight, transposed)
            self.key = key
        return self.w_packed

    def get_i8(self, conv, pad_cout=False):
        """int8 digit planes of the denoiser family; pad_cout: output channels zero-padded to a multiple of 16 (logits layer)."""
        key = (_ver(conv.weight), _ver(conv.bias), bool(pad_cout))
        if key != self.key_i8:
            self.i8 = ops.den_pack_weight_i8(conv.weight, conv.bias, pad_cout=pad_cout)
            self.key_i8 = key
        return self.i8

    def get_fp6(self, conv):
        """six fp6 digit planes + fp64 scale / bias for the block-scaled MFMA kernel (built on first use)."""
        key = (_ver(conv.weight), _ver(conv.bias))
        if key != getattr(self, 'key_fp6', None):
            self.fp6 = ops.den_pack_weight_fp6(conv.weight, conv.bias)
            self.key_fp6 = key
        return self.fp6

    def get_vae_fp6(self, conv):
        """digit tiles of the fp6 kernel of the VQ-VAE's stride-2 layers (csrc/vae_fp6.hip), built on first use."""
        key = (_ver(conv.weight), _ver(conv.bias))
        if key != getattr(self, 'key_convT_fp6', None):
            self.convT_fp6 = ops.vae_fp6_pack(conv.weight, conv.bias, isinstance(conv, nn.ConvTranspose2d))
            self.key_convT_fp6 = key
        return self.convT_fp6

    def get_fp6v2(self, conv):
        """digit tiles of the second-generation fp6 kernel (+ scale / bias / L1 norms / an fp32 copy), built on first use."""
        key = (_ver(conv.weight), _ver(conv.bias))
        if key != getattr(self, 'key_fp6v2', None):
            self.fp6v2 = ops.den_pack_weight_fp6v2(conv.weight, conv.bias)
            self.key_fp6v2 = key
        return self.fp6v2

    def get_i8_generic(self, conv):
        """int8 digit planes in the layout of the gather-MFMA kernel (any k, Conv2d or ConvTranspose2d)."""
        key = (_ver(conv.weight), _ver(conv.bias))
        if key != getattr(self, 'key_i8g', None):
            self.i8g = ops.pack_conv_weight_i8(conv.weight, conv.bias, isinstance(conv, nn.ConvTranspose2d))
            self.key_i8g = key
        return self.i8g


def conv_geometry(conv):
    layer._check_plain(conv)
    transposed = isinstance(conv, nn.ConvTranspose2d)
    return dict(k=layer._one(conv.kernel_size, 'kernel_size'), stride=layer._one(conv.stride, 'stride'),
                pad=layer._one(conv.padding, 'padding'), transposed=transposed,
                out_pad=layer._one(conv.output_padding, 'output_padding') if transposed else 0)


def _is_conv(m):
    return isinstance(m, (layer.Conv2d, layer.ConvTranspose2d))


def has_hooks(module):
    """True if a forward (pre-)hook is registered anywhere below ``module``.  Fused launches bypass the children's
    ``__call__``; with hooks present (the reference's syops counter, monitors) containers run child by child instead -- every
    child is still a HIP kernel, and every hook sees the [T,B,C,H,W] tensors it would see in the reference."""
    return any(m._forward_hooks or m._forward_pre_hooks for m in module.modules())


def invalidate_derived(module):
    """Drop every cached derived form of the parameters below ``module``: packed convolution weights (fp32 / int8 /
    fp6 digit planes), folded BatchNorm terms, captured sampler graphs.  The caches are keyed by ``(data_ptr,
    _version)``, which in-place writes through ``.data`` (``p.data.copy_(ema)``) do NOT change -- call this after such a
    write.  ``load_state_dict`` and train()/eval() transitions call it on their own, and the sampler compares a content
    checksum of the denoiser's tensors on every ``sample()`` call (``AbsorbingDiffusion.verify_weights``).

    Every module below ``module`` also gets its ``_derived_epoch`` bumped: a captured graph bakes the ADDRESSES of the
    derived tensors, so whoever captured one keys it on ``derived_epoch(root)`` and re-captures after an invalidation
    (the graph entry itself keeps the tensors it addresses alive: ``derived_refs``)."""
    for m in module.modules():
        pr = getattr(m, '_spk_params', None)
        if pr is not None:
            pr.invalidate()
        if isinstance(m, layer.BatchNorm2d):
            m._affine_cache = None
        g = getattr(m, '_graphs', None)
        if isinstance(g, dict):
            g.clear()
        tab = getattr(m, '_spikegen_tab', None)
        if isinstance(tab, dict):                   # the per-token spike-pattern tables of a spike generator (tokens_to_s32):
            for ent in tab.values():                # contents unknown from here on; the buffer stays (a captured graph may address it)
                ent[1] = ent[2] = None
        object.__setattr__(m, '_derived_epoch', getattr(m, '_derived_epoch', 0) + 1)


def derived_epoch(module):
    """Changes whenever derived forms anywhere below ``module`` were dropped (part of every captured graph's key)."""
    return tuple(getattr(m, '_derived_epoch', 0) for m in module.modules())


def derived_refs(module):
    """References to every derived tensor currently cached below ``module`` (packed weights, BN terms): a captured graph
    stores this list so that the memory its launches address by raw pointer outlives any later invalidation."""
    refs = []
    for m in module.modules():
        pr = getattr(m, '_spk_params', None)
        if pr is not None:
            refs.append(tuple(v for v in vars(pr).values() if v is not None))
        if isinstance(m, layer.BatchNorm2d) and m._affine_cache is not None:
            refs.append(m._affine_cache)
    return refs


class FusedSequential(nn.Sequential):
    """nn.Sequential whose (conv, bn, lif) triples run as fused HIP kernels in eval / multi-step mode."""

    def __init__(self, *args):
        super().__init__(*args)
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: invalidate_derived(module))

    def invalidate(self):
        invalidate_derived(self)

    def train(self, mode: bool = True):
        if mode != self.training:                 # a training phase rewrites the weights (also through .data): rebuild
            invalidate_derived(self)
        return super().train(mode)

    def _blocks(self):
        mods = list(self)
        blocks, i = [], 0
        while i < len(mods):
            if (i + 2 < len(mods) and _is_conv(mods[i]) and isinstance(mods[i + 1], layer.BatchNorm2d)
                    and isinstance(mods[i + 2], neuron.LIFNode)):
                blocks.append((mods[i], mods[i + 1], mods[i + 2]))
                i += 3
            elif _is_conv(mods[i]) and i == len(mods) - 1:
                blocks.append((mods[i], None, None))
                i += 1
            else:
                return None
        return blocks

    def _fusable(self, blocks):
        if blocks is None:
            return False
        for conv, bn, lif in blocks:
            if conv.step_mode != 'm':
                return False
            if lif is not None:
                if lif.step_mode != 'm' or bn.step_mode != 'm' or lif.training or bn.training:
                    return False
                if (lif.tau != 2.0 or lif.v_threshold != 1.0 or lif.v_reset != 0.0 or not lif.decay_input or lif.store_v_seq):
                    return False
        return True

    def _trainable_fused(self, blocks, x):
        """train() mode with autograd: every (conv, bn, lif) triple can run as library conv + the native fused
        BatchNorm(batch statistics)+LIF(surrogate gradient) operator."""
        if blocks is None or not torch.is_grad_enabled() or x.dim() != 5 or x.shape[0] > ops.MAX_T or not x.is_cuda:
            return False
        for conv, bn, lif in blocks:
            if not conv.training or conv.step_mode != 'm':
                return False
            if lif is not None:
                if not (lif.training and bn.training) or lif.step_mode != 'm' or bn.step_mode != 'm':
                    return False
                if (not isinstance(lif.surrogate_function, surrogate.ATan) or lif.v_reset is None or not lif.decay_input
                        or lif.store_v_seq or bn.momentum is None or not bn.track_running_stats):
                    return False
        return True

    # exact MFMA forward for spike-input 3x3 convolutions in training (ops.SpikeConvTrainFunction); False = library forward
    exact_train_forward = True

    def exact_conv_fits(self, blocks, x):
        """True for a single conv-only block whose convolution the exact MFMA training forward takes on spikes x."""
        if blocks is None or len(blocks) != 1 or blocks[0][2] is not None or blocks[0][1] is not None or not self.exact_train_forward:
            return False
        conv = blocks[0][0]
        w = conv.weight
        if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
            w.data = w.data.contiguous(memory_format=torch.channels_last)
        return (isinstance(conv, layer.Conv2d) and conv.groups == 1 and tuple(conv.dilation) == (1, 1)
                and conv.padding_mode == 'zeros' and conv.kernel_size[0] == conv.kernel_size[1]
                and ops.den_fp6_supported(conv.out_channels, conv.in_channels, conv.kernel_size[0], conv.stride[0],
                                          conv.padding[0] if not isinstance(conv.padding, str) else -1, x.shape[0],
                                          x.shape[3], x.shape[4]))

    def train_forward(self, x, binary_input=False, prep=None, want_c4=False):
        """[T,B,C,H,W] -> spikes [T,B,C',H',W'] (or the raw conv output of a conv-only last block), differentiable.
        Convolution: ROCm library operator through torch -- or, when the caller states that x holds spikes
        (``binary_input``) and the shape fits, the exact fp6 MFMA forward with the native (7x7) or library backward; BN + LIF:
        ops.BNLIFTrainFunction (one native operator).  prep: the ops.WeightPrep of this module's (single) convolution for the
        current iteration; want_c4: the last block's spikes are also left as C4 records (attribute ``_spk_c4`` of the result),
        which the next module's exact forward picks up instead of converting the fp32 tensor."""
        blocks = self._blocks()
        c4_in = getattr(x, '_spk_c4', None)
        if c4_in is not None:                          # (packed spikes, version of x they were made from): stale after an in-place write
            c4_in = c4_in[0] if c4_in[1] == x._version else None
        for bi, (conv, bn, lif) in enumerate(blocks):
            w = conv.weight
            if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
                # keep the parameter itself channels-last while training: the library's NHWC kernels then read it (and
                # write its gradient) without a per-call layout copy; values, shape and state_dict keys are unchanged
                w.data = w.data.contiguous(memory_format=torch.channels_last)
            if (binary_input and self.exact_train_forward and isinstance(conv, layer.Conv2d) and conv.groups == 1
                    and tuple(conv.dilation) == (1, 1) and conv.padding_mode == 'zeros'
                    and ops.den_fp6_supported(conv.out_channels, conv.in_channels, conv.kernel_size[0], conv.stride[0],
                                              conv.padding[0] if not isinstance(conv.padding, str) else -1, x.shape[0],
                                              x.shape[3], x.shape[4]) and conv.kernel_size[0] == conv.kernel_size[1]):
                x = ops.SpikeConvTrainFunction.apply(x, conv.weight, conv.bias, prep if len(blocks) == 1 else None,
                                                     c4_in if bi == 0 else None)
            else:
                x = conv(x)
            c4_in = None
            if lif is None:
                continue
            binary_input = True                      # what follows a LIF is a spike train
            v0 = lif.v if torch.is_tensor(lif.v) else None
            if v0 is None and float(lif.v) != float(lif.v_reset):
                v0 = torch.full_like(x[0], float(lif.v))
            emit = bool(want_c4) and bi == len(blocks) - 1 and x.shape[0] == 16
            out = ops.BNLIFTrainFunction.apply(x, bn.weight, bn.bias, v0, bn.running_mean, bn.running_var,
                                               bn.momentum, bn.eps, lif.tau, lif.v_threshold, lif.v_reset,
                                               float(lif.surrogate_function.alpha), lif.detach_reset, emit)
            x, lif.v = out[0], out[1]
            if emit and out[2] is not None:
                x._spk_c4 = (out[2], x._version)
            sink = getattr(self, '_nbt_sink', None)
            if sink is not None:
                sink.append(bn.num_batches_tracked)   # (the caller bumps all counters of the model with one launch)
            else:
                bn.num_batches_tracked.add_(1)
        return x

    def forward(self, x):
        blocks = self._blocks()
        if self._trainable_fused(blocks, x):
            return self.train_forward(x)
        if not self._fusable(blocks) or x.dim() != 5 or x.shape[0] > ops.MAX_T or has_hooks(self):
            for m in self:                      # layer by layer: still HIP kernels, just not fused
                x = m(x)
            return x
        return self.run(x, IN_SEQ, final='f32')['f32']

    @staticmethod
    def _vae_kind(conv, geo, T, H, W):
        if has_hooks(conv):
            return None
        return ops.vae_fp6_kind(conv.in_channels, conv.out_channels, geo['k'], geo['stride'], geo['pad'], geo['out_pad'],
                                geo['transposed'], T, H, W)

    @staticmethod
    def _next_convT_fp6(block, cur, geo, T):
        """Will ``block`` (the one after the layer with geometry ``geo`` applied to the PTC tensor ``cur``) take the fp6
        transposed-convolution kernel?"""
        conv, bn, lif = block
        if bn is None or lif is None or has_hooks(conv) or has_hooks(bn) or has_hooks(lif):
            return False
        g2 = conv_geometry(conv)
        Ho = ops.conv_out_size(cur.shape[1], geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
        Wo = ops.conv_out_size(cur.shape[2], geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
        return ops.convT_fp6_supported(conv.in_channels, conv.out_channels, g2['k'], g2['stride'], g2['pad'], g2['out_pad'],
                                       g2['transposed'], T, Ho, Wo)

    @staticmethod
    def _next_convT_fp6_hw(block, H, W, geo, T):
        """_next_convT_fp6 for a layer whose input map is H x W (S32 input: the map sits in dims 2, 3)."""
        conv, bn, lif = block
        if bn is None or lif is None or has_hooks(conv) or has_hooks(bn) or has_hooks(lif):
            return False
        g2 = conv_geometry(conv)
        Ho = ops.conv_out_size(H, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
        Wo = ops.conv_out_size(W, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
        return ops.convT_fp6_supported(conv.in_channels, conv.out_channels, g2['k'], g2['stride'], g2['pad'], g2['out_pad'],
                                       g2['transposed'], T, Ho, Wo)

    @staticmethod
    def _collapsible(block, coef, T):
        """Can the conv-only read-out ``block`` take time-collapsed spikes (ops.readout_collapsed)?"""
        conv, bn, lif = block
        if bn is not None or lif is not None or coef is None or T != 16 or has_hooks(conv):
            return False
        geo = conv_geometry(conv)
        return (geo['stride'] == 1 and geo['k'] % 2 == 1 and geo['pad'] == geo['k'] // 2 and geo['out_pad'] == 0 and
                conv.in_channels % 8 == 0 and ops.readout_collapsed_supported(conv.in_channels, conv.out_channels, geo['k']))

    def tokens_to_s32(self, tokens, codebook, T=16, epoch=()):
        """The spike generator container (one 1x1 Conv2d + BN + LIF block) applied to the code vectors of ``tokens`` [B,h,w], as
        nibble-packed S32 spikes [B,1,h,w,16,16] by the per-token pattern table (ops.spikegen_tokens_s32): embedding look-up,
        repeat(T), convolution, BN, LIF from the reset state and the PTC -> S32 packing in two launches.  None when the container
        is not that block in fused-eval configuration (the caller then takes the layer-by-layer path).
        epoch: the caller's ``derived_epoch`` of whatever owns ``codebook`` (it lives outside this container): together with this
        container's own epoch it makes ``invalidate_derived`` -- the one signal for writes that bump no ``_version`` (``.data`` copies,
        graph replays of an in-graph optimizer) -- reach the table."""
        blocks = self._blocks()
        if not self._fusable(blocks) or len(blocks) != 1 or T != 16 or not tokens.is_cuda:
            return None
        conv, bn, lif = blocks[0]
        if bn is None or lif is None or has_hooks(conv) or has_hooks(bn) or has_hooks(lif) or isinstance(conv, nn.ConvTranspose2d):
            return None
        geo = conv_geometry(conv)
        if (geo['k'] != 1 or geo['stride'] != 1 or geo['pad'] != 0 or conv.out_channels not in (16, 32) or
                conv.in_channels != codebook.shape[1]):
            return None
        if not hasattr(conv, '_spk_params'):
            object.__setattr__(conv, '_spk_params', ConvParams())
        a, b = bn.affine_terms()
        bias = None if conv.bias is None else conv.bias.detach()
        packed = conv._spk_params.get(conv)
        # the table is kept while the parameter versions that went into it AND the invalidation epochs are unchanged; the buffer
        # belongs to this container (no table is shared between models)
        key = (conv._spk_params.key, bn._affine_cache[0], _ver(codebook), derived_epoch(self), tuple(epoch))
        if not isinstance(getattr(self, '_spikegen_tab', None), dict):
            object.__setattr__(self, '_spikegen_tab', {})
        return ops.spikegen_tokens_s32(tokens, codebook, packed, bias, a, b, T=T, table_key=key, table_slot=self._spikegen_tab)

    def run(self, x, in_kind, final='f32', T=None, in1=None, coef=None, apply_tanh=False, want_u8=False,
            stateful=True, want_pre=False, chunk_out=None, impl='auto', want_counts=False, need_radius=None):
        """Run all blocks fused.

        x: per in_kind (IN_SEQ fp32 [T,B,C,H,W]; IN_TINV fp32 [B,C,H,W] with ``T`` given; IN_PTC u8 [B,H,W,T,C]).
        final: output of the LAST block -- 'f32' (spikes TBCHW, or the raw conv output when the last block has no
        BN/LIF), 'ptc', 'both', 'memout' (read-out of a conv-only last block) or 'mean'.
        stateful: honour and update each LIFNode's ``v`` (module semantics); False = fresh state, nothing written.
        chunk_out: channel chunking of the PTC output of the last block (32 = the CPTC layout the int8 MFMA kernel reads,
        ops.CHUNK_C4 = the nibble-packed fp4 layout of the fp6 MFMA kernel).
        need_radius: see ops.den_conv3x3_mfma_fp6v2 (single-block containers on S32 spikes inside a position-list scope).
        impl: 'auto' uses the MFMA kernel that matches the input layout (CPTC u8 -> int8 planes, C4 -> fp6 planes;
        3x3/s1/p1, T=16), 'direct' never.
        Returns dict(ptc=, f32=, pre=[...], u8=)."""
        blocks = self._blocks()
        if not self._fusable(blocks):
            raise RuntimeError('spkdiff: this container is not in fused-eval configuration (eval(), step_mode "m")')
        if in_kind == IN_SEQ:
            T = x.shape[0]
        elif in_kind == IN_PTC:
            T = x.shape[-2]
        elif T is None:
            raise ValueError('T is required for a time-invariant input')
        cur, kind = x, in_kind
        out = {'ptc': None, 'f32': None, 'u8': None, 'pre': [], 'cnt': None}
        for bi, (conv, bn, lif) in enumerate(blocks):
            with ops.timed(getattr(conv, '_spk_tag', None)):          # bench.py tags layers it wants timed in situ
                last = bi == len(blocks) - 1
                if not hasattr(conv, '_spk_params'):
                    object.__setattr__(conv, '_spk_params', ConvParams())
                geo = conv_geometry(conv)
                bias = None if conv.bias is None else conv.bias.detach()
                src1 = in1 if (last and in1 is not None) else None
                c4 = kind == IN_PTC and cur.dim() == 6 and cur.dtype == ops.C4_DTYPE
                if c4 and cur.shape[-1] == 16 and lif is not None and not stateful and impl != 'direct' and not want_pre and src1 is None:
                    # the VQ-VAE's stride-2 layers on the fp6 MFMA (stateless calls; csrc/vae_fp6.hip)
                    vk = self._vae_kind(conv, geo, T, cur.shape[2], cur.shape[3])
                    tail_ok = final == 'memout' and self._collapsible(blocks[-1], coef, T)
                    if vk == ops.VAE_OUT_COLLAPSED and bi == len(blocks) - 2 and tail_ok:
                        # decoder convT2, handing the read-out layer its time-collapsed spikes
                        a, b = bn.affine_terms()
                        cur = ops.vae_fp6_fwd(cur, conv._spk_params.get_vae_fp6(conv), conv.out_channels, bn_a=a, bn_b=b,
                                              transposed=True, out_kind=vk, coef=coef)
                        kind = 'collapsed'
                        continue
                    if (vk == ops.VAE_OUT_S32 and bi == len(blocks) - 3 and tail_ok and conv.out_channels % 32 == 0 and
                            self._next_convT_fp6_hw(blocks[bi + 1], cur.shape[2], cur.shape[3], geo, T)):
                        # decoder convT1 fed nibble-packed spikes directly (the token-table spike generator)
                        a, b = bn.affine_terms()
                        cur = ops.vae_fp6_fwd(cur, conv._spk_params.get_vae_fp6(conv), conv.out_channels, bn_a=a, bn_b=b,
                                              transposed=True, out_kind=ops.VAE_OUT_S32)
                        kind = IN_PTC
                        continue
                    if vk == ops.VAE_OUT_PTC and not last:
                        a, b = bn.affine_terms()                  # encoder conv2: plain u8 PTC out for the 1x1 layer
                        cur = ops.vae_fp6_fwd(cur, conv._spk_params.get_vae_fp6(conv), conv.out_channels, bn_a=a, bn_b=b,
                                              transposed=False, out_kind=vk)
                        kind = IN_PTC
                        continue
                if c4 and cur.shape[-1] == 16:               # S32 records: the sampler's second-generation fp6 kernel
                    ok = (impl != 'direct' and lif is not None and not want_pre and src1 is None and not geo['transposed'] and
                          not stateful and
                          ops.den_fp6v2_supported(conv.out_channels, conv.in_channels, geo['k'], geo['stride'], geo['pad'], T,
                                                  cur.shape[2], cur.shape[3]) and
                          (not last or (final == 'ptc' and chunk_out == ops.CHUNK_S32)))
                    if not ok:
                        raise NotImplementedError('spkdiff: S32 spikes are only consumed by the fp6v2 MFMA conv (3x3/s1/p1 + BN + '
                                                  'LIF, T=16, 7x7, fresh LIF state, S32 output)')
                    a, b = bn.affine_terms()
                    o = ops.den_conv3x3_mfma_fp6v2(cur, conv._spk_params.get_fp6v2(conv), conv.out_channels, bn_a=a, bn_b=b,
                                                   want_counts=last and want_counts, need_radius=need_radius)
                    if last and want_counts:
                        out['ptc'], out['cnt'] = o
                    elif last:
                        out['ptc'] = o
                    else:
                        cur, kind = o, IN_PTC
                    continue
                if c4:
                    ok = (impl != 'direct' and lif is not None and not want_pre and src1 is None and not geo['transposed'] and
                          ops.den_fp6_supported(conv.out_channels, conv.in_channels, geo['k'], geo['stride'], geo['pad'], T,
                                                cur.shape[2], cur.shape[3]) and
                          (not last or (final == 'ptc' and chunk_out == ops.CHUNK_C4)))
                    if not ok:
                        raise NotImplementedError('spkdiff: fp4-packed (C4) spikes are only consumed by the fp6 MFMA conv '
                                                  '(3x3/s1/p1 + BN + LIF, T=16, C4 output)')
                    a, b = bn.affine_terms()
                    v = None
                    if stateful:
                        shape = (cur.shape[0], conv.out_channels, cur.shape[2], cur.shape[3])
                        if isinstance(lif.v, float):
                            lif.v = torch.full(shape, lif.v, dtype=torch.float32, device=cur.device)
                        elif tuple(lif.v.shape) != shape:
                            raise RuntimeError(f'LIFNode state has shape {tuple(lif.v.shape)} but the input implies '
                                               f'{shape}; call functional.reset_net first')
                        v = lif.v
                    o = ops.den_conv3x3_mfma_fp6(cur, conv._spk_params.get_fp6(conv), conv.out_channels, bn_a=a, bn_b=b, v=v,
                                                 want_counts=last and want_counts)
                    if last and want_counts:
                        out['ptc'], out['cnt'] = o
                    elif last:
                        out['ptc'] = o
                    else:
                        cur, kind = o, IN_PTC
                    continue
                cptc = kind == IN_PTC and cur.dim() == 6 and cur.shape[-1] == 32 and (src1 is None or src1.dim() == 6)
                use_mfma = (impl != 'direct' and cptc and not geo['transposed'] and not want_pre and
                            (lif is not None or final == 'mean') and
                            ops.den_mfma_supported(conv.out_channels, conv.in_channels, geo['k'], geo['stride'], geo['pad'],
                                                   T, cur.shape[2], cur.shape[3]) and
                            (lif is None or not last or (final == 'ptc' and chunk_out == 32)))
                if use_mfma:
                    packed = conv._spk_params.get_i8(conv)
                    if lif is not None:
                        a, b = bn.affine_terms()
                        v = None
                        if stateful:
                            shape = (cur.shape[0], conv.out_channels, cur.shape[2], cur.shape[3])
                            if isinstance(lif.v, float):
                                lif.v = torch.full(shape, lif.v, dtype=torch.float32, device=cur.device)
                            elif tuple(lif.v.shape) != shape:
                                raise RuntimeError(f'LIFNode state has shape {tuple(lif.v.shape)} but the input implies '
                                                   f'{shape}; call functional.reset_net first')
                            v = lif.v
                        o = ops.den_conv3x3_mfma(cur, packed, conv.out_channels, mode=MODE_LIF, in1=src1, bn_a=a, bn_b=b, v=v,
                                                 want_counts=last and want_counts)
                        if last and want_counts:
                            out['ptc'], out['cnt'] = o
                        elif last:
                            out['ptc'] = o
                        else:
                            cur, kind = o, IN_PTC
                    else:
                        out['f32'] = ops.den_conv3x3_mfma(cur, packed, conv.out_channels, mode=MODE_MEAN, in1=src1)
                    continue
                if kind == 'collapsed':                      # (see the producing block below)
                    r = ops.readout_collapsed(cur, conv.weight.detach(), bias, coef, apply_tanh=apply_tanh, want_u8=want_u8,
                                              k=geo['k'], pad=geo['pad'], transposed=geo['transposed'])
                    out['f32'], out['u8'] = r['f32'], r['u8']
                    continue
                # spiking VQ-VAE layers: gather-MFMA kernel (plain PTC input, T = 16)
                plain_ptc = kind == IN_PTC and cur.dim() == 5 and src1 is None
                g_mode = MODE_LIF if lif is not None else (MODE_MEMOUT if final == 'memout' else None)
                use_gather = (impl != 'direct' and plain_ptc and g_mode is not None and not want_pre and not want_counts and
                              not (last and chunk_out) and
                              ops.conv_mfma_supported(conv.in_channels, conv.out_channels, T, g_mode))
                if use_gather:
                    packed = conv._spk_params.get_i8_generic(conv)
                    if lif is not None:
                        a, b = bn.affine_terms()
                        v = None
                        if stateful:
                            Ho = ops.conv_out_size(cur.shape[1], geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                            Wo = ops.conv_out_size(cur.shape[2], geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                            shape = (cur.shape[0], conv.out_channels, Ho, Wo)
                            if isinstance(lif.v, float):
                                lif.v = torch.full(shape, lif.v, dtype=torch.float32, device=cur.device)
                            elif tuple(lif.v.shape) != shape:
                                raise RuntimeError(f'LIFNode state has shape {tuple(lif.v.shape)} but the input implies '
                                                   f'{shape}; call functional.reset_net first')
                            v = lif.v
                        # the next block runs on the fp6 MFMA (decoder convT2, stateless call): it reads nibble-packed spikes;
                        # this one (decoder convT1) does too where its shape has an instance
                        if (not last and not stateful and impl != 'direct' and bi == len(blocks) - 3 and final == 'memout' and
                                conv.out_channels % 32 == 0 and self._collapsible(blocks[-1], coef, T) and
                                self._next_convT_fp6(blocks[bi + 1], cur, geo, T)):
                            if self._vae_kind(conv, geo, T, cur.shape[1], cur.shape[2]) == ops.VAE_OUT_S32:
                                cur = ops.vae_fp6_fwd(ops.ptc_to_s32(cur), conv._spk_params.get_vae_fp6(conv), conv.out_channels,
                                                      bn_a=a, bn_b=b, transposed=True, out_kind=ops.VAE_OUT_S32)
                            else:
                                cur = ops.conv_mfma_fused(cur, packed, conv.out_channels, mode=MODE_LIF, bn_a=a, bn_b=b, v=None,
                                                          out_s32=True, **geo)
                            kind = IN_PTC
                            continue
                        # a linear read-out layer next (conv-only last block + 'memout'): hand it sum_t coef[t] * spikes[t]
                        # instead of the spike frames -- one convolution instead of T, no spike tensor in between
                        if (not last and bi == len(blocks) - 2 and final == 'memout' and impl != 'direct' and
                                self._collapsible(blocks[-1], coef, T)):
                            cur = ops.conv_mfma_fused(cur, packed, conv.out_channels, mode=MODE_LIF, bn_a=a, bn_b=b, v=v,
                                                      collapse_coef=coef, **geo)
                            kind = 'collapsed'
                            continue
                        o = ops.conv_mfma_fused(cur, packed, conv.out_channels, mode=MODE_LIF, bn_a=a, bn_b=b, v=v, **geo)
                        if last:
                            out['ptc'] = o
                            if final in ('f32', 'both'):
                                out['f32'] = ops.ptc_to_spikes(o)
                        else:
                            cur, kind = o, IN_PTC
                    else:
                        r = ops.conv_mfma_fused(cur, packed, conv.out_channels, mode=MODE_MEMOUT, coef=coef,
                                                apply_tanh=apply_tanh, want_u8=want_u8, **geo)
                        out['f32'], out['u8'] = r['f32'], r['u8']
                    continue
                w_packed = conv._spk_params.get(conv)
                if lif is not None:
                    a, b = bn.affine_terms()
                    v = None
                    if stateful:
                        B = cur.shape[0] if kind != IN_SEQ else cur.shape[1]
                        if kind == IN_PTC:
                            H, W = (cur.shape[2], cur.shape[3]) if cur.dim() == 6 else (cur.shape[1], cur.shape[2])
                        else:
                            H, W = cur.shape[-2], cur.shape[-1]
                        Ho = ops.conv_out_size(H, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                        Wo = ops.conv_out_size(W, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                        if isinstance(lif.v, float):
                            lif.v = torch.full((B, conv.out_channels, Ho, Wo), lif.v, dtype=torch.float32,
                                               device=cur.device)
                        elif tuple(lif.v.shape) != (B, conv.out_channels, Ho, Wo):
                            raise RuntimeError(f'LIFNode state has shape {tuple(lif.v.shape)} but the input implies '
                                               f'{(B, conv.out_channels, Ho, Wo)}; call functional.reset_net first')
                        v = lif.v
                    co_chunk = chunk_out if last else None
                    if (not last and not stateful and impl != 'direct' and not want_pre and conv.out_channels % 32 == 0 and
                            blocks[bi + 1][2] is not None):
                        # the next block runs on the fp6 MFMA (encoder conv2, stateless call): it reads nibble-packed spikes
                        if kind == IN_PTC:
                            Hi, Wi = (cur.shape[2], cur.shape[3]) if cur.dim() == 6 else (cur.shape[1], cur.shape[2])
                        else:
                            Hi, Wi = cur.shape[-2], cur.shape[-1]
                        Hn = ops.conv_out_size(Hi, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                        Wn = ops.conv_out_size(Wi, geo['k'], geo['stride'], geo['pad'], geo['transposed'], geo['out_pad'])
                        nconv = blocks[bi + 1][0]
                        if self._vae_kind(nconv, conv_geometry(nconv), T, Hn, Wn) == ops.VAE_OUT_PTC and bi + 1 < len(blocks) - 1:
                            co_chunk = ops.CHUNK_S32
                    r = ops.conv_fused(cur, w_packed, bias, in_kind=kind, T=T, mode=MODE_LIF, in1=src1, bn_a=a, bn_b=b,
                                       v=v, want_ptc=(not last) or final in ('ptc', 'both'),
                                       want_f32=last and final in ('f32', 'both'), want_pre=want_pre,
                                       chunk_out=co_chunk, want_counts=last and want_counts, **geo)
                    if want_pre:
                        out['pre'].append(r['pre'])
                    if last:
                        out['ptc'], out['f32'], out['cnt'] = r['ptc'], r['f32'], r['cnt']
                    else:
                        cur, kind = r['ptc'], IN_PTC
                else:
                    if final == 'memout':
                        r = ops.conv_fused(cur, w_packed, bias, in_kind=kind, T=T, mode=MODE_MEMOUT, in1=src1, coef=coef,
                                           apply_tanh=apply_tanh, want_u8=want_u8, **geo)
                    elif final == 'mean':
                        r = ops.conv_fused(cur, w_packed, bias, in_kind=kind, T=T, mode=MODE_MEAN, in1=src1, **geo)
                    else:
                        r = ops.conv_fused(cur, w_packed, bias, in_kind=kind, T=T, mode=MODE_RAW, in1=src1, **geo)
                    out['f32'], out['u8'] = r['f32'], r['u8']
        return out
