#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r4_gputest3.log
grep -v PARITY gpurun_out/r4_gputest3.log | tail -6
{ for pass in 1 2; do for l in lib_vgpr_merge lib_r4b lib_r4b_noepi; do echo "== pass $pass $l"; SPKDIFF_LIB=$R/$V/$l.so python tools/fp6v2_time.py $R/$V/$l.so; SPKDIFF_LIB=$R/$V/$l.so python tools/listed_time.py 256 3 dense; done; done; } > gpurun_out/r4_ab2.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_ab2.log | tail -20
bash tools/full_size_oracle.sh | tail -4 | cut -c1-1500
