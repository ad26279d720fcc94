"""The counting pass (R/syops/engine.py): register a hook on every supported module, run the model, average.

The model runs through this build's own modules -- with hooks registered the fused containers execute child by child
(``spkdiff.fused.has_hooks``), so each hook sees the same ``[T, B, C, H, W]`` tensors as in the reference and every layer is
still a HIP kernel.  Per-module results live in ``module.__syops__`` while a pass is active (the reference's contract for
user-supplied hooks).
"""
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from spikingjelly.activation_based import functional

from . import ops as _ops
from .utils import syops_to_string, params_to_string


def is_supported_instance(module):
    return type(module) in _ops.MODULES_MAPPING or type(module) in _ops.CUSTOM_MODULES_MAPPING


def get_model_parameters_number(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def _accumulated(module):
    """[overall, ACs, MACs, rate] below ``module``: a supported module reports its own, a container the sum of its children."""
    if is_supported_instance(module):
        return module.__syops__
    total = np.array([0.0, 0.0, 0.0, 0.0])
    for child in module.children():
        total += _accumulated(child)
    return total


def _accumulated_params(module):
    if is_supported_instance(module):
        return module.__params__
    return sum(_accumulated_params(c) for c in module.children())


def _batch_counter_hook(module, input, output):
    batch_size = 1
    if len(input) > 0:
        batch_size = len(input[0])
    else:
        print('Warning! No positional inputs found for a module, assuming batch size is 1.')
    module.__batch_counter__ += batch_size
    module.__times_counter__ += 1


class _CountingPass:
    """Hooks + counters for one model, removed again on exit."""

    def __init__(self, model, ost, verbose, ignore_list):
        self.model, self.handles, self.touched = model, [], []
        model.__batch_counter__ = 0
        model.__times_counter__ = 0
        self.handles.append(model.register_forward_hook(_batch_counter_hook))
        seen = set()
        for m in model.modules():
            if type(m) in ignore_list:
                seen.add(type(m))
                if is_supported_instance(m):
                    m.__params__ = 0
                    m.__syops__ = np.array([0.0, 0.0, 0.0, 0.0])
                    self.touched.append(m)
            elif is_supported_instance(m):
                m.__syops__ = np.array([0.0, 0.0, 0.0, 0.0])
                m.__params__ = get_model_parameters_number(m)
                hook = _ops.CUSTOM_MODULES_MAPPING.get(type(m)) or _ops.MODULES_MAPPING[type(m)]
                self.handles.append(m.register_forward_hook(hook))
                self.touched.append(m)
                seen.add(type(m))
            else:
                if verbose and type(m) not in (nn.Sequential, nn.ModuleList) and type(m) not in seen:
                    print('Warning: module ' + type(m).__name__ + ' is treated as a zero-op.', file=ost)
                seen.add(type(m))

    def average_cost(self):
        total = _accumulated(self.model)
        return np.array([v / self.model.__batch_counter__ for v in total]), get_model_parameters_number(self.model)

    def close(self):
        for h in self.handles:
            h.remove()
        for m in self.touched:
            for attr in ('__syops__', '__params__'):
                if hasattr(m, attr):
                    delattr(m, attr)
        for attr in ('__batch_counter__', '__times_counter__'):
            if hasattr(self.model, attr):
                delattr(self.model, attr)


def print_model_with_syops(model, total_syops, total_params, syops_units='GMac', param_units='M', precision=3, ost=sys.stdout):
    """One line per supported module: parameters, ACs, MACs (absolute and share of the model) and firing rate."""
    total_syops = [max(v, 1) for v in total_syops[:3]]
    total_params = max(total_params, 1)
    for name, m in model.named_modules():
        if not is_supported_instance(m):
            continue
        cost = np.array(_accumulated(m), dtype=np.float64)
        cost[:3] /= model.__batch_counter__
        cost[3] /= model.__times_counter__
        n_par = _accumulated_params(m)
        print(', '.join([f'{name} ({type(m).__name__}): ' + params_to_string(n_par, units=param_units, precision=precision),
                         '{:.3%} Params'.format(n_par / total_params),
                         syops_to_string(cost[1], units=syops_units, precision=precision),
                         '{:.3%} ACs'.format(cost[1] / total_syops[1]),
                         syops_to_string(cost[2], units=syops_units, precision=precision),
                         '{:.3%} MACs'.format(cost[2] / total_syops[2]),
                         '{:.3%} Spike Rate'.format(cost[3] / 100.)]), file=ost)


def get_syops_pytorch(model, input_res, dataloader=None, print_per_layer_stat=True, input_constructor=None, ost=sys.stdout,
                      verbose=False, ignore_modules=[], custom_modules_hooks={}, output_precision=3, syops_units='GMac',
                      param_units='M'):
    _ops.CUSTOM_MODULES_MAPPING = dict(custom_modules_hooks)
    model.eval()
    run = _CountingPass(model, ost, verbose, list(ignore_modules))
    try:
        dev = next(model.parameters()).device
        if dataloader is not None:
            for batch, _ in dataloader:
                batch = batch.float().to(dev)
                with torch.no_grad():
                    model(batch)
                functional.reset_net(model)
        elif input_constructor:
            with torch.no_grad():
                model(**input_constructor(input_res))
            functional.reset_net(model)
        else:
            with torch.no_grad():
                model(torch.empty((1, *input_res), dtype=next(model.parameters()).dtype, device=dev))
            functional.reset_net(model)
        syops_count, params_count = run.average_cost()
        if print_per_layer_stat:
            print_model_with_syops(model, syops_count, params_count, ost=ost, syops_units=syops_units,
                                   param_units=param_units, precision=output_precision)
        per_module = {name: np.array(m.__syops__, dtype=np.float64) for name, m in model.named_modules()
                      if is_supported_instance(m)}
    finally:
        run.close()
        _ops.CUSTOM_MODULES_MAPPING = {}
    get_syops_pytorch.last_per_module = per_module
    return syops_count, params_count
