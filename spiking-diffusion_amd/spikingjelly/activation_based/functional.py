"""Network-level helpers of the neuron surface.

Behaviour taken from SJ/activation_based/functional.py:13-40 (``reset_net``), :42-107 (``set_step_mode``),
:109-149 (``set_backend``), :525-565 (``multi_step_forward``) and :653-688 (``seq_to_ann_forward``); written for the
small module tree of this build (no step-mode containers exist here, so no sub-tree is ever exempt).
"""
import logging

import torch
import torch.nn as nn

from . import base

_log = logging.getLogger(__name__)


def _stateful_or_warn(m, what):
    if not isinstance(m, base.MemoryModule):
        _log.warning("%s on %s, which is not a MemoryModule of this package", what, type(m).__name__)


def reset_net(net: nn.Module):
    """Restore every stateful sub-module to its reset state (``v`` becomes the python float it was built with)."""
    for m in net.modules():
        reset = getattr(m, 'reset', None)
        if callable(reset):
            _stateful_or_warn(m, "reset()")
            reset()


def set_step_mode(net: nn.Module, step_mode: str):
    """Switch every step-aware sub-module to single-step ('s') or multi-step ('m') operation."""
    for m in net.modules():
        if not hasattr(m, 'step_mode'):
            continue
        if not isinstance(m, base.StepModule):
            _log.warning("step_mode set on %s, which is not a StepModule of this package", type(m).__name__)
        m.step_mode = step_mode


def set_backend(net: nn.Module, backend: str, instance=(nn.Module,)):
    """Select ``backend`` on every sub-module of type ``instance`` that lists it in ``supported_backends``."""
    for m in net.modules():
        if not (isinstance(m, instance) and hasattr(m, 'backend')):
            continue
        _stateful_or_warn(m, "backend switch")
        if backend in m.supported_backends:
            m.backend = backend
        else:
            _log.warning("%s keeps backend=%s: %s is not supported", type(m).__name__, m.backend, backend)


def _apply_chain(modules, y):
    if isinstance(modules, (list, tuple, nn.Sequential)):
        for m in modules:
            y = m(y)
        return y
    return modules(y)


def seq_to_ann_forward(x_seq, stateless_module):
    """[T, N, ...] -> fold T into the batch, apply the stateless module(s), unfold to [T, N, ...]."""
    T, N = x_seq.shape[0], x_seq.shape[1]
    y = _apply_chain(stateless_module, x_seq.flatten(0, 1))
    return y.view((T, N) + tuple(y.shape[1:]))


def multi_step_forward(x_seq, single_step_module):
    """Apply single-step module(s) to every time slice of x_seq [T, N, ...] and stack the results."""
    return torch.stack([_apply_chain(single_step_module, x_seq[t]) for t in range(x_seq.shape[0])])
