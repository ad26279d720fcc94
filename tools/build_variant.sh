#!/bin/bash
# usage: tools/build_variant.sh <file.hip> "<extra flags>" <out.so>  -- rebuild ONE object with extra -D flags and link a variant library
set -e
cd $(dirname $0)/../spiking-diffusion_amd/csrc
f=$1; flags=$2; out=$3
/opt/rocm/bin/hipcc $flags --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -c $f -o /tmp/variant_$$.o 2>/dev/null
objs=$(ls *.o | grep -v "^${f%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs /tmp/variant_$$.o
rm -f /tmp/variant_$$.o
