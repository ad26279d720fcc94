"""Step-mode wrappers of the stateless layers used by ``snn_model``: Conv2d, ConvTranspose2d, BatchNorm2d (+Linear).

Surface of SJ/activation_based/layer.py:125-173, :276-325, :423-465, :900-922.  They subclass the ``torch.nn``
layers (parameters, ``state_dict`` keys and constructors are torch's), but ``forward`` runs the HIP kernels of
``libspkdiff.so``; in 'm' mode T is folded into the batch exactly like ``functional.seq_to_ann_forward``.

Training: when a module is in train() mode with autograd enabled, a convolution runs forward and
backward on this library's fp32 matrix-core training kernels (ops.NativeConvTrainFunction, csrc/conv_train.hip).  Shapes
those kernels do not take (and every layer when ops.NATIVE_TRAIN_FORWARD is off: the parity runs) run the exact direct kernel
forward up to ops.EXACT_TRAIN_FORWARD_MACS multiply-accumulates and the framework's operator beyond, with the native backward
where it fits and the framework's otherwise; a stand-alone batch-norm runs as the ROCm library operator through torch.  The
native training operators of this build are the surrogate-gradient LIF and the fused BatchNorm+LIF block tail
(``spkdiff.ops.LIFTrainFunction`` / ``BNLIFTrainFunction``), which ``snn_model`` uses for its Conv-BN-LIF blocks.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from spkdiff import ops

from . import base


def _one(v, what):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise NotImplementedError(f'spkdiff: anisotropic {what}={v} is not used by the inference path')
        v = v[0]
    if isinstance(v, str):
        raise NotImplementedError(f'spkdiff: {what}="{v}" is not used by the inference path')
    return int(v)


def _check_plain(m):
    if m.groups != 1 or _one(m.dilation, 'dilation') != 1 or m.padding_mode != 'zeros':
        raise NotImplementedError('spkdiff: groups/dilation/padding_mode other than the defaults are not implemented')


def _library_path(m, x):
    """Training with autograd: the differentiable library operator (see the module docstring).  Device tensors only --
    there is no CPU path in either mode."""
    if not (m.training and torch.is_grad_enabled()):
        return False
    if not x.is_cuda:
        raise RuntimeError(f"spkdiff: input is on '{x.device}'; there is no CPU path (move module and tensors to a ROCm "
                           "device)")
    return True


def _fold(x, step_mode, nd):
    """Return (x folded to nd dims, unfold shape prefix or None)."""
    if step_mode == 's':
        return x, None
    if x.dim() != nd + 1:
        raise ValueError(f'expected x with shape [T, N, C, H, W], but got x with shape {x.shape}!')
    return x.flatten(0, 1), (x.shape[0], x.shape[1])


def _unfold(y, prefix):
    return y if prefix is None else y.view(prefix + tuple(y.shape[1:]))


class Conv2d(nn.Conv2d, base.StepModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode='zeros', step_mode='s'):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.step_mode = step_mode

    def extra_repr(self):
        return super().extra_repr() + f', step_mode={self.step_mode}'

    def forward(self, x: torch.Tensor):
        _check_plain(self)
        y, prefix = _fold(x, self.step_mode, 4)
        if _library_path(self, y):
            k, st, pd = self.kernel_size[0], _one(self.stride, 'stride'), _one(self.padding, 'padding')
            macs = y.shape[0] * self.out_channels * self.in_channels * k * k * (y.shape[2] // st) * (y.shape[3] // st)
            if ops.NATIVE_TRAIN_FORWARD and ops.conv_train_supported(y.shape, self.weight, st, pd, False, 0, y.requires_grad,
                                                                     forward=True):
                y = ops.NativeConvTrainFunction.apply(y, self.weight, self.bias, st, pd, False, 0)
            elif macs <= ops.EXACT_TRAIN_FORWARD_MACS:
                y = ops.ExactConvTrainFunction.apply(y, self.weight, self.bias, st, pd, False, 0)
            else:
                y = F.conv2d(y, self.weight, self.bias, self.stride, self.padding)
        else:
            y = ops.conv2d(y, self.weight, self.bias, _one(self.stride, 'stride'), _one(self.padding, 'padding'))
        return _unfold(y, prefix)


class ConvTranspose2d(nn.ConvTranspose2d, base.StepModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, groups=1,
                 bias=True, dilation=1, padding_mode='zeros', step_mode='s'):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias,
                         dilation, padding_mode)
        self.step_mode = step_mode

    def extra_repr(self):
        return super().extra_repr() + f', step_mode={self.step_mode}'

    def forward(self, x: torch.Tensor):
        _check_plain(self)
        y, prefix = _fold(x, self.step_mode, 4)
        if _library_path(self, y):
            k, st, pd = self.kernel_size[0], _one(self.stride, 'stride'), _one(self.padding, 'padding')
            op = _one(self.output_padding, 'output_padding')
            macs = y.shape[0] * self.out_channels * self.in_channels * k * k * y.shape[2] * y.shape[3]
            if ops.NATIVE_TRAIN_FORWARD and ops.conv_train_supported(y.shape, self.weight, st, pd, True, op, y.requires_grad,
                                                                     forward=True):
                return _unfold(ops.NativeConvTrainFunction.apply(y, self.weight, self.bias, st, pd, True, op), prefix)
            if macs <= ops.EXACT_TRAIN_FORWARD_MACS:
                return _unfold(ops.ExactConvTrainFunction.apply(y, self.weight, self.bias, st, pd, True, op), prefix)
            return _unfold(F.conv_transpose2d(y, self.weight, self.bias, self.stride, self.padding,
                                              self.output_padding), prefix)
        y = ops.conv_transpose2d(y, self.weight, self.bias, _one(self.stride, 'stride'), _one(self.padding, 'padding'),
                                 _one(self.output_padding, 'output_padding'))
        return _unfold(y, prefix)


class BatchNorm2d(nn.BatchNorm2d, base.StepModule):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, step_mode='s'):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.step_mode = step_mode
        self._affine_cache = None

    def extra_repr(self):
        return super().extra_repr() + f', step_mode={self.step_mode}'

    def affine_terms(self):
        """(a, b) with y = fma(x, a, b); recomputed when any of the four tensors changed (in place or replaced)."""
        if self.training or not self.track_running_stats:
            raise NotImplementedError('spkdiff: BatchNorm2d batch statistics (training) are outside the inference '
                                      'hot path; call .eval()')
        tens = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((None if t is None else (t.data_ptr(), t._version, str(t.device))) for t in tens)
        if self._affine_cache is None or self._affine_cache[0] != key:
            self._affine_cache = (key, ops.bn_prepare(self.weight, self.bias, self.running_mean, self.running_var,
                                                      self.eps))
        return self._affine_cache[1]

    def forward(self, x: torch.Tensor):
        y, prefix = _fold(x, self.step_mode, 4)
        if self.training and torch.is_grad_enabled() and _library_path(self, y):
            # nn.BatchNorm2d.forward in training mode: batch statistics, running statistics and the counter updated
            return _unfold(nn.BatchNorm2d.forward(self, y), prefix)
        a, b = self.affine_terms()
        return _unfold(ops.bn_eval(y, a, b), prefix)


class Linear(nn.Linear, base.StepModule):
    """Declared for ``from ...layer import *`` completeness (used only by the out-of-scope SNN_VAE baseline)."""

    def __init__(self, in_features, out_features, bias=True, step_mode='s'):
        super().__init__(in_features, out_features, bias)
        self.step_mode = step_mode

    def forward(self, x):
        raise NotImplementedError('spkdiff: layer.Linear belongs to the SNN_VAE baseline, outside the hot path')
