"""Surrogate functions: forward only (Heaviside).  SJ/activation_based/surrogate.py:13-51,118-155,664-760.

Training (the surrogate gradient) is outside the hot path; the objects exist so that
``neuron.LIFNode(surrogate_function=surrogate.ATan())`` constructs exactly as in the reference.
"""
import torch.nn as nn


class SurrogateFunctionBase(nn.Module):
    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.spiking = spiking
        self.alpha = alpha

    def set_spiking_mode(self, spiking: bool):
        self.spiking = spiking

    def extra_repr(self):
        return f'alpha={self.alpha}, spiking={self.spiking}'

    def forward(self, x):
        raise NotImplementedError(
            "spkdiff: surrogate functions are evaluated inside the fused HIP LIF kernels (x >= 0); "
            "calling them as stand-alone modules (training path) is out of scope")


class ATan(SurrogateFunctionBase):
    def __init__(self, alpha=2.0, spiking=True):
        super().__init__(alpha, spiking)


class Sigmoid(SurrogateFunctionBase):
    def __init__(self, alpha=4.0, spiking=True):
        super().__init__(alpha, spiking)
