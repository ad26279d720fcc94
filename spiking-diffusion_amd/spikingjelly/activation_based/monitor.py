"""Placeholder for ``spikingjelly.activation_based.monitor`` (imported, never used, by R/snn_model/vae_model.py:16).
The reference module drags in tensorboard (SJ/activation_based/monitor.py:7); nothing on the hot path needs it."""
