#!/bin/bash
# the stress record of profiles/<tag>_stress.txt (one box): usage tools/stress.sh [tag=r5]
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
TAG=${1:-r6}
cd $R
run() { echo "== $*"; "$@" 2>/dev/null | tail -6; }
{
run python tools/fp6v2_stress.py 200 32
run python tools/vae_fp6_stress.py 100 8
run python tools/modes_stress.py 30 256
run python tools/r6_stress.py 120
# the measured-and-dropped launch forms: only in the `make variants` library (tools/variants_check.sh builds it)
V=$R/spiking-diffusion_amd/spkdiff/variants/libspkdiff_variants.so
make -C $R/spiking-diffusion_amd/csrc -j8 variants >/dev/null 2>&1
if [ -f "$V" ]; then
  SPKDIFF_LIB=$V SPKDIFF_V2_LAG=1 run python tools/fp6v2_stress.py 40 32
  SPKDIFF_LIB=$V SPKDIFF_V2_WAVES=12 run python tools/fp6v2_stress.py 40 32
  SPKDIFF_LIB=$V SPKDIFF_V2_DUO=1 run python tools/fp6v2_stress.py 40 32
  SPKDIFF_LIB=$V SPKDIFF_V2_DEFER=1 run python tools/fp6v2_stress.py 40 32
fi
run python tools/backward_stress.py 90 7
run python tools/tinv_stress.py 400 1
} > gpurun_out/${TAG}_stress_final.txt 2>&1
cat gpurun_out/${TAG}_stress_final.txt
