"""``get_model_complexity_info`` with the reference's signature and return values (R/syops/flops_counter.py:16-65)."""
import sys

import torch.nn as nn

from .engine import get_syops_pytorch
from .utils import syops_to_string, params_to_string


def get_model_complexity_info(model, input_res, dataloader=None, print_per_layer_stat=True, as_strings=True,
                              input_constructor=None, ost=sys.stdout, verbose=False, ignore_modules=[],
                              custom_modules_hooks={}, backend='pytorch', syops_units=None, param_units=None,
                              output_precision=2):
    assert type(input_res) is tuple
    assert len(input_res) >= 1
    assert isinstance(model, nn.Module)
    if backend != 'pytorch':
        raise ValueError('Wrong backend name')
    syops_count, params_count = get_syops_pytorch(model, input_res, dataloader, print_per_layer_stat, input_constructor, ost,
                                                  verbose, ignore_modules, custom_modules_hooks,
                                                  output_precision=output_precision, syops_units=syops_units,
                                                  param_units=param_units)
    if as_strings:
        return ([syops_to_string(syops_count[i], units=syops_units, precision=output_precision) for i in range(3)],
                params_to_string(params_count, units=param_units, precision=output_precision))
    return syops_count, params_count
