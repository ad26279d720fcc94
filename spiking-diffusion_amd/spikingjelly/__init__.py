"""Minimal, from-scratch ``spikingjelly`` surface for the Spiking-Diffusion inference path (MI355X build).

Only what ``snn_model`` imports is provided: ``activation_based.{base, neuron, layer, functional,
surrogate, monitor}`` and ``visualizing``.  Every compute op is executed by ``libspkdiff.so`` (hand-written HIP).
"""
__version__ = "0.0.0.0.14+spkdiff"
