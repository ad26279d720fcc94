#!/bin/bash
# round 5: the deferred-scan form of the fp6v2 kernel -- correctness, then same-box A/B against the scan between two K loops
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deferred or duo or fp6v2 or wide_dynamic or f5_denoiser or timed_configuration or f15 or f13" 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-600 | tee gpurun_out/r5_call4_pytest.txt
timeout 1500 python tools/ab.py --passes 3 --what layers,dense base:SPKDIFF_V2_DEFER=0 defer:SPKDIFF_V2_DEFER=1 2>&1 | tee gpurun_out/r5_call4_ab.txt
