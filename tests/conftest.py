"""pytest configuration: registers the ``gpu`` marker and puts the drop-in tree on sys.path.

``spiking-diffusion_amd/`` is a *directory of top-level packages* (``snn_model``,
``spikingjelly``, ``spkdiff``) laid out like the reference's release directory, so the
reference's ``main.py`` placed beside them imports them unchanged.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spiking-diffusion_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variants: launch forms that exist only in the `make variants` library (tests/variants/; "
                                       "SPKDIFF_LIB=.../libspkdiff_variants.so on a GPU box)")
    config.addinivalue_line("markers", "slow: minutes of host-side oracle work; skipped unless SPKDIFF_RUN_SLOW=1 "
                                       "(run once per round through gpurun, log kept under profiles/)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SPKDIFF_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow: set SPKDIFF_RUN_SLOW=1 (tools/full_size_oracle.sh)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One JSON line with every parity figure the session measured (tests/parity_report.py)."""
    import json
    try:
        from parity_report import RESULTS
    except Exception:
        return
    if RESULTS:
        terminalreporter.write_line("PARITY_REPORT " + json.dumps(RESULTS, sort_keys=True))
