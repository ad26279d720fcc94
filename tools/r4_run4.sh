#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r4_gputest4.log
grep -v PARITY gpurun_out/r4_gputest4.log | tail -12 | cut -c1-300
{ for pass in 1 2; do for l in lib_r4c lib_r4c_oldtr lib_r4c_nonmax lib_r4c_oldtr_nonmax; do echo "== pass $pass $l"; SPKDIFF_LIB=$R/$V/$l.so python tools/fp6v2_time.py $R/$V/$l.so; SPKDIFF_LIB=$R/$V/$l.so python tools/listed_time.py 256 3 dense; done; done; } > gpurun_out/r4_ab3.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_ab3.log | cut -c1-300 | tail -20
