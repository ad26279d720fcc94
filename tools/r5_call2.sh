#!/bin/bash
# round 5: the duo form of the fp6v2 kernel -- correctness, then same-box A/B of its phase-control modes against the one-workgroup form
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "duo or fp6v2_kernel_bit_equal or wide_dynamic" 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-400 | tee gpurun_out/r5_call3_pytest.txt
timeout 1500 python tools/ab.py --passes 2 --what layers,dense base:SPKDIFF_V2_DUO=0 duo:SPKDIFF_V2_DUO=1 duo_s4:LIB=spiking-diffusion_amd/spkdiff/variants/duo_s4.so duo_nosteal:LIB=spiking-diffusion_amd/spkdiff/variants/duo_nosteal.so duo_noprio:LIB=spiking-diffusion_amd/spkdiff/variants/duo_noprio.so duo_head95:SPKDIFF_V2_DUO=95 2>&1 | tee gpurun_out/r5_call3_ab.txt
for m in 1 95; do
  echo "== phase picture, v2_duo=$m"
  SPKDIFF_V2_DUO=$m SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/duo_dbg.so timeout 300 python tools/duo_phase.py 512 256 2>&1 | grep -v amdgpu.ids
  SPKDIFF_V2_DUO=$m SPKDIFF_LIB=$R/spiking-diffusion_amd/spkdiff/variants/duo_dbg.so timeout 300 python tools/duo_phase.py 128 64 2>&1 | grep -v amdgpu.ids | head -2
done 2>&1 | tee gpurun_out/r5_call3_phase.txt
