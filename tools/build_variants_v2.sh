#!/bin/bash
# Build tuning variants of libspkdiff.so that differ in den_mfma_fp6v2.hip only:  tools/build_variants_v2.sh name "-DFLAG=..." ...
set -e
cd "$(dirname "$0")/../spiking-diffusion_amd/csrc"
mkdir -p ../spkdiff/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc $flags --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -c den_mfma_fp6v2.hip -o /tmp/den_mfma_fp6v2_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../spkdiff/variants/libspkdiff_$name.so $(ls *.o | grep -v '^den_mfma_fp6v2.o$') /tmp/den_mfma_fp6v2_$name.o
  echo built $name
done
