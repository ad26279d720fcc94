#!/bin/bash
# The measured-and-dropped launch forms (csrc/variants/fp6v2_forms.inc, 4 / 12 waves) against the default form, on a GPU box:
# (re)builds the variants library and runs tests/variants against it.  Not part of the driver's suite.
set -e
cd "$(dirname "$0")/.."
V=spiking-diffusion_amd/spkdiff/variants/libspkdiff_variants.so
make -C spiking-diffusion_amd/csrc -j8 variants >/dev/null 2>&1      # (always: a library left from another tree state has other signatures)
SPKDIFF_LIB=$PWD/$V python -m pytest tests/variants -m variants -q "$@"
