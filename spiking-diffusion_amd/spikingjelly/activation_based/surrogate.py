"""Surrogate-function objects: constructor surface only.

In inference the forward of every surrogate is the Heaviside step ``x >= 0`` (SJ/activation_based/surrogate.py:13-51),
which the fused HIP LIF kernels evaluate in place; the ATan backward (:664-678) is evaluated inside the BPTT kernel
``spk_lif_train_bwd`` when a ``LIFNode`` runs in training mode.  The classes exist so that ``neuron.LIFNode(surrogate_function=surrogate.ATan())`` is built
exactly as the reference builds it, and keep the reference's attributes (``alpha``, ``spiking``).
"""
import torch.nn as nn


class SurrogateFunctionBase(nn.Module):
    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.alpha, self.spiking = alpha, spiking

    def set_spiking_mode(self, spiking: bool):
        self.spiking = spiking

    def extra_repr(self):
        return 'alpha={}, spiking={}'.format(self.alpha, self.spiking)

    def forward(self, x):
        raise NotImplementedError('spkdiff: the spike non-linearity runs inside the HIP LIF kernels; stand-alone '
                                  'surrogate evaluation (training path) is not part of this build')


def _make(name, default_alpha):
    def __init__(self, alpha=default_alpha, spiking=True):
        SurrogateFunctionBase.__init__(self, alpha, spiking)
    return type(name, (SurrogateFunctionBase,), {'__init__': __init__, '__doc__': f'{name} surrogate (alpha={default_alpha})'})


ATan = _make('ATan', 2.0)
Sigmoid = _make('Sigmoid', 4.0)
