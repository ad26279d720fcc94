"""Per-module counting hooks in the reference's convention (R/syops/ops.py): a forward hook adds
``[overall operations, accumulates (ACs), multiply-accumulates (MACs), firing rate in %]`` to ``module.__syops__``.
An operation counts as an accumulate when the module's input is a spike tensor (then weighted by the input firing rate,
R/syops/ops.py:14-24) and as a multiply-accumulate otherwise.  The firing rates are counted on the device
(``spk_count_spikes``); the operation counts follow the reference's arithmetic on the shapes its hooks see, i.e. on the
``[T, B, C, H, W]`` tensors of multi-step mode (R/syops/ops.py:121-158 reads ``input.shape[0]`` as the batch and
``output.shape[2:]`` as the spatial extent; the fixture F11 pins the resulting numbers).
"""
import numpy as np
import torch.nn as nn

from spikingjelly.activation_based import neuron
from spkdiff import ops as _k


def spike_rate(inp):
    """(is_spike, rate): a tensor whose nonzero entries are all exactly 1 is a spike tensor with rate = mean; anything else
    counts with rate 1 (R/syops/ops.py:14-24 tests ``len(unique) <= 2`` within [0, 1]; two-valued tensors other than
    {0, 1} do not occur on this path)."""
    st = _k.count_spikes(inp)
    if st["binary"]:
        return True, st["total"] / st["numel"]
    return False, 1


def _add(module, overall, spike, rate):
    overall = int(overall)
    module.__syops__[0] += overall
    if spike:
        module.__syops__[1] += overall * rate
    else:
        module.__syops__[2] += overall
    module.__syops__[3] += rate * 100


def empty_syops_counter_hook(module, input, output):
    module.__syops__ += np.array([0.0, 0.0, 0.0, 0.0])


def conv_syops_counter_hook(conv_module, input, output):
    x = input[0]
    spike, rate = spike_rate(x)
    lead = x.shape[0]                                    # what the reference reads as the batch size
    positions = lead * int(np.prod(list(output.shape[2:])))
    per_position = int(np.prod(list(conv_module.kernel_size))) * conv_module.in_channels * (conv_module.out_channels // conv_module.groups)
    overall = per_position * positions + (conv_module.out_channels * positions if conv_module.bias is not None else 0)
    _add(conv_module, overall, spike, rate)


def bn_syops_counter_hook(module, input, output):
    x = input[0]
    spike, rate = spike_rate(x)
    _add(module, int(np.prod(x.shape)) * (2 if module.affine else 1), spike, rate)


def LIF_syops_counter_hook(module, input, output):
    n = input[0].numel()
    module.__syops__[0] += int(n)
    _, rate = spike_rate(output[0])                      # the first time step's spikes, as the reference reads them
    module.__syops__[1] += int(n)
    module.__syops__[3] += rate * 100


IF_syops_counter_hook = LIF_syops_counter_hook


def relu_syops_counter_hook(module, input, output):
    spike, rate = spike_rate(output[0])
    _add(module, output.numel(), spike, rate)


def linear_syops_counter_hook(module, input, output):
    x = input[0]
    spike, rate = spike_rate(x)
    last = output.shape[-1]
    _add(module, int(np.prod(x.shape) * last + (last if module.bias is not None else 0)), spike, rate)


def pool_syops_counter_hook(module, input, output):
    x = input[0]
    spike, rate = spike_rate(x)
    _add(module, int(np.prod(x.shape)), spike, rate)


def upsample_syops_counter_hook(module, input, output):
    first = output[0]
    n = first.shape[0]
    for v in first.shape[1:]:
        n *= v
    spike, rate = spike_rate(first)
    _add(module, n, spike, rate)


CUSTOM_MODULES_MAPPING = {}

# exact types, like the reference (R/syops/engine.py:332-335): a subclass -- the spikingjelly layer.* wrappers -- is not
# matched unless the caller registers it through custom_modules_hooks
MODULES_MAPPING = {
    nn.Conv1d: conv_syops_counter_hook, nn.Conv2d: conv_syops_counter_hook, nn.Conv3d: conv_syops_counter_hook,
    nn.ConvTranspose1d: conv_syops_counter_hook, nn.ConvTranspose2d: conv_syops_counter_hook,
    nn.ConvTranspose3d: conv_syops_counter_hook,
    nn.ReLU: relu_syops_counter_hook, nn.PReLU: relu_syops_counter_hook, nn.ELU: relu_syops_counter_hook,
    nn.LeakyReLU: relu_syops_counter_hook, nn.ReLU6: relu_syops_counter_hook, nn.GELU: relu_syops_counter_hook,
    nn.MaxPool1d: pool_syops_counter_hook, nn.AvgPool1d: pool_syops_counter_hook, nn.AvgPool2d: pool_syops_counter_hook,
    nn.MaxPool2d: pool_syops_counter_hook, nn.MaxPool3d: pool_syops_counter_hook, nn.AvgPool3d: pool_syops_counter_hook,
    nn.AdaptiveMaxPool1d: pool_syops_counter_hook, nn.AdaptiveAvgPool1d: pool_syops_counter_hook,
    nn.AdaptiveMaxPool2d: pool_syops_counter_hook, nn.AdaptiveAvgPool2d: pool_syops_counter_hook,
    nn.AdaptiveMaxPool3d: pool_syops_counter_hook, nn.AdaptiveAvgPool3d: pool_syops_counter_hook,
    nn.BatchNorm1d: bn_syops_counter_hook, nn.BatchNorm2d: bn_syops_counter_hook, nn.BatchNorm3d: bn_syops_counter_hook,
    nn.InstanceNorm1d: bn_syops_counter_hook, nn.InstanceNorm2d: bn_syops_counter_hook,
    nn.InstanceNorm3d: bn_syops_counter_hook, nn.GroupNorm: bn_syops_counter_hook,
    neuron.LIFNode: LIF_syops_counter_hook,
    nn.Linear: linear_syops_counter_hook,
    nn.Upsample: upsample_syops_counter_hook,
}
