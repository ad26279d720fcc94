"""Placeholder for ``spikingjelly.visualizing`` (imported, never used, by R/snn_model/vae_model.py:17)."""
