"""Drop-in ``snn_model`` package of the MI355X build (same module/class names as R/snn_model)."""
