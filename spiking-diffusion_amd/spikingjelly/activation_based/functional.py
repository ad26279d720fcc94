"""Network-level helpers of the neuron surface: SJ/activation_based/functional.py:13-40 (reset_net),
:42-107 (set_step_mode), :109-149 (set_backend), :653-688 (seq_to_ann_forward)."""
import logging

import torch.nn as nn

from . import base


def reset_net(net: nn.Module):
    """Call ``reset()`` on every sub-module that has one (restores each memory to its reset value)."""
    for m in net.modules():
        if hasattr(m, 'reset'):
            if not isinstance(m, base.MemoryModule):
                logging.warning(f'Trying to call `reset()` of {m}, which is not spikingjelly.activation_based.base'
                                f'.MemoryModule')
            m.reset()


def set_step_mode(net: nn.Module, step_mode: str):
    """Set ``step_mode`` on every sub-module that has the attribute."""
    for m in net.modules():
        if hasattr(m, 'step_mode'):
            if not isinstance(m, base.StepModule):
                logging.warning(f'Trying to set the step mode for {m}, which is not spikingjelly.activation_based'
                                f'.base.StepModule')
            m.step_mode = step_mode


def set_backend(net: nn.Module, backend: str, instance=(nn.Module,)):
    """The reference's operator-plugin switch: select ``backend`` on every module that supports it."""
    for m in net.modules():
        if isinstance(m, instance) and hasattr(m, 'backend'):
            if not isinstance(m, base.MemoryModule):
                logging.warning(f'Trying to set the backend for {m}, which is not spikingjelly.activation_based.base'
                                f'.MemoryModule')
            if backend in m.supported_backends:
                m.backend = backend
            else:
                logging.warning(f'{m} does not supports for backend={backend}. It will still use backend={m.backend}.')


def seq_to_ann_forward(x_seq, stateless_module):
    """Fold [T, N, ...] into [T*N, ...], apply the stateless module(s), unfold."""
    y = x_seq.flatten(0, 1)
    if isinstance(stateless_module, (list, tuple, nn.Sequential)):
        for m in stateless_module:
            y = m(y)
    else:
        y = stateless_module(y)
    return y.view((x_seq.shape[0], x_seq.shape[1]) + tuple(y.shape[1:]))


def multi_step_forward(x_seq, single_step_module):
    import torch
    outs = []
    for t in range(x_seq.shape[0]):
        y = x_seq[t]
        if isinstance(single_step_module, (list, tuple, nn.Sequential)):
            for m in single_step_module:
                y = m(y)
        else:
            y = single_step_module(y)
        outs.append(y)
    return torch.stack(outs)
