"""``snn_model.snn_layers`` of the MI355X build -- R/snn_model/snn_layers.py (41 lines) counterpart.

* ``MembraneOutputLayer``: sum_t x[t] * 0.8**(T-1-t); the ``coef`` buffer (shape (T,1,1,1,1), in ``state_dict``)
  is built exactly like R/snn_model/snn_layers.py:31-34, with ``n_steps`` a constructor argument (default 16,
  the reference's literal) so that BASELINE config 1 (T=4) is expressible.  Forward = ``spk_memout_fwd`` (HIP).
* ``PSP``: the post-synaptic-potential filter of the VQ-VAE training losses (R/snn_model/snn_layers.py:6-26):
  ``spk_psp`` forward, its adjoint as the backward.
"""
import torch
import torch.nn as nn

from spkdiff import ops

__all__ = ['PSP', 'MembraneOutputLayer', 'torch', 'nn']


class PSP(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.tau_s = 2

    def forward(self, inputs):
        """inputs: (T, N, ...) -> syns (T, N, ...): syn_t = syn_{t-1} + (inputs[t] - syn_{t-1}) / tau_s."""
        return ops.PSPFunction.apply(inputs, float(self.tau_s))


class MembraneOutputLayer(nn.Module):
    def __init__(self, n_steps: int = 16) -> None:
        super().__init__()
        arr = torch.arange(n_steps - 1, -1, -1)
        self.register_buffer("coef", torch.pow(0.8, arr)[:, None, None, None, None])  # (T,1,1,1,1)

    def forward(self, x):
        """x : (T,N,C,H,W) -> (N,C,H,W)"""
        if torch.is_grad_enabled() and x.requires_grad:
            return ops.MemoutFunction.apply(x, self.coef)
        return ops.memout(x, self.coef)
