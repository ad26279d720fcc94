"""Spiking neurons of the inference path: ``BaseNode`` state handling and ``LIFNode``.

Surface of SJ/activation_based/neuron.py:23-263 (BaseNode) and :603-1011 (LIFNode); the eval multi-step kernel
``jit_eval_multi_step_forward_hard_reset_decay_input`` (:799-811) is replaced by ``spk_lif_fwd`` (HIP, one pass,
membrane potential in registers across T); in training mode the surrogate-gradient pair ``spk_lif_train_fwd`` /
``spk_lif_train_bwd`` runs behind a ``torch.autograd.Function`` (the reference's CuPy ATGF contract, :954-966).  ``v`` keeps the reference's lifetime: python float after
``reset()``, tensor after the first forward, carried across forwards until the next ``reset()``.
"""
from typing import Callable

import torch

from spkdiff import ops

from . import base, surrogate


class BaseNode(base.MemoryModule):
    def __init__(self, v_threshold: float = 1., v_reset: float = 0., surrogate_function: Callable = surrogate.Sigmoid(),
                 detach_reset: bool = False, step_mode='s', backend='torch', store_v_seq: bool = False):
        for name, val, ok in (('v_reset', v_reset, v_reset is None or isinstance(v_reset, float)),
                              ('v_threshold', v_threshold, isinstance(v_threshold, float)),
                              ('detach_reset', detach_reset, isinstance(detach_reset, bool))):
            assert ok, f'{name}={val!r} has the wrong type'
        super().__init__()
        self.register_memory('v', v_reset if v_reset is not None else 0.)      # membrane potential, float until first use
        self.v_threshold, self.v_reset, self.detach_reset = v_threshold, v_reset, detach_reset
        self.surrogate_function = surrogate_function
        self.step_mode, self.backend, self.store_v_seq = step_mode, backend, store_v_seq

    @property
    def store_v_seq(self):
        return self._store_v_seq

    @store_v_seq.setter
    def store_v_seq(self, value: bool):
        # SJ/activation_based/neuron.py:119-129: the first True registers the `v_seq` memory
        self._store_v_seq = bool(value)
        if value and not hasattr(self, 'v_seq'):
            self.register_memory('v_seq', None)

    def extra_repr(self):
        fields = ('v_threshold', 'v_reset', 'detach_reset', 'step_mode', 'backend')
        return ', '.join(f'{k}={getattr(self, k)}' for k in fields)

    def v_float_to_tensor(self, x: torch.Tensor):
        """SJ/activation_based/neuron.py:260-263: expand the float state to x's shape on first use."""
        if isinstance(self.v, float):
            self.v = torch.full_like(x.data, self.v)


class LIFNode(BaseNode):
    def __init__(self, tau: float = 2., decay_input: bool = True, v_threshold: float = 1.,
                 v_reset: float = 0., surrogate_function: Callable = surrogate.Sigmoid(),
                 detach_reset: bool = False, step_mode='s', backend='torch', store_v_seq: bool = False):
        assert isinstance(tau, float) and tau > 1., f'tau={tau!r} must be a float > 1'
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.tau, self.decay_input = tau, decay_input

    @property
    def supported_backends(self):
        # both names run the same HIP kernel; 'torch' is kept because it is the reference's default string
        return ('torch', 'hip')

    def extra_repr(self):
        return super().extra_repr() + f', tau={self.tau}'

    def _check_supported(self, x):
        if x.dtype != torch.float32:
            raise NotImplementedError(x.dtype)

    def _plain(self):
        """The configuration snn_model uses (hard reset, decay_input, no v_seq): the vectorised spk_lif_fwd kernel."""
        return self.v_reset is not None and self.decay_input and not self.store_v_seq

    def multi_step_forward(self, x_seq: torch.Tensor):
        self._check_supported(x_seq)
        self.v_float_to_tensor(x_seq[0])
        if not self.v.is_contiguous():
            self.v = self.v.contiguous()
        if self.training and not self._plain():
            raise NotImplementedError('spkdiff: the BPTT kernel implements hard reset with decay_input=True (the '
                                      'configuration snn_model trains); the other forms run in eval() mode')
        if self.training:
            # surrogate-gradient BPTT: HIP forward that keeps h, HIP backward; the state stays
            # in the autograd graph across calls like the reference's ``self.v = v_seq[-1]``
            # (SJ/activation_based/neuron.py:954-966)
            if not isinstance(self.surrogate_function, surrogate.ATan):
                raise NotImplementedError('spkdiff: the BPTT kernel implements the ATan surrogate (the one snn_model uses)')
            spike_seq, self.v = ops.LIFTrainFunction.apply(x_seq.contiguous(), self.v, self.tau, self.v_threshold,
                                                           self.v_reset, float(self.surrogate_function.alpha),
                                                           self.detach_reset)
            return spike_seq
        if self._plain():
            return ops.lif_fwd(x_seq, self.v, self.tau, self.v_threshold, self.v_reset)
        # soft reset / decay_input=False / store_v_seq: SJ/activation_based/neuron.py:971-1011 (eval dispatch)
        spike_seq, v_seq = ops.lif_fwd_ex(x_seq, self.v, self.tau, self.v_threshold, self.v_reset,
                                          soft_reset=self.v_reset is None, decay_input=self.decay_input,
                                          want_v_seq=self.store_v_seq)
        if self.store_v_seq:
            self.v_seq = v_seq
        return spike_seq

    def single_step_forward(self, x: torch.Tensor):
        return self.multi_step_forward(x.unsqueeze(0))[0]
