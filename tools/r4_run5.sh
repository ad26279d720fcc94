#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
cd $R
python -m pytest tests -m gpu -q -x -k "fp6v2 or f5_denoiser or timed_configuration or f13 or step_tail or full_sample or wide_dynamic" 2>&1 | tail -6 > gpurun_out/r4_gputest5.log
grep -v PARITY gpurun_out/r4_gputest5.log | tail -4 | cut -c1-300
{ for pass in 1 2 3; do for l in lib_r4d lib_r4d_nomask; do echo "== pass $pass $l"; SPKDIFF_LIB=$R/$V/$l.so python tools/fp6v2_time.py $R/$V/$l.so; SPKDIFF_LIB=$R/$V/$l.so python tools/listed_time.py 256 3 dense; done; done; } > gpurun_out/r4_ab4.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_ab4.log | cut -c1-300 | tail -14
