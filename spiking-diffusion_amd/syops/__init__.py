"""``syops`` of the MI355X build: the reference's synaptic-operation / energy counter (R/syops, called from
R/main.py:325-338) behind the same entry point, with the firing rates counted on the device by ``spk_count_spikes``."""
from .flops_counter import get_model_complexity_info  # noqa: F401
