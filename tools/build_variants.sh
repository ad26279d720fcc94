#!/bin/bash
# Build tuning variants of libspkdiff.so (tools/fp6_variants.py times them):  tools/build_variants.sh name "-DFLAG=..." ...
set -e
cd "$(dirname "$0")/../spiking-diffusion_amd/csrc"
mkdir -p ../spkdiff/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $flags --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -c den_mfma_fp6.hip -o /tmp/den_mfma_fp6_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../spkdiff/variants/libspkdiff_$name.so $(ls *.o | grep -v '^den_mfma_fp6.o$') /tmp/den_mfma_fp6_$name.o
  echo built $name
done
