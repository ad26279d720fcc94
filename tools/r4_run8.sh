#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r4_gputest8.log
grep -v PARITY gpurun_out/r4_gputest8.log | tail -8 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4
{
echo "== tools/fp6v2_stress.py 200 32"; python tools/fp6v2_stress.py 200 32 2>&1 | tail -1
echo "== tools/vae_fp6_stress.py 100 8"; python tools/vae_fp6_stress.py 100 8 2>&1 | tail -1
echo "== tools/modes_stress.py 30 256"; python tools/modes_stress.py 30 256 2>&1 | tail -1
echo "== SPKDIFF_V2_LAG=1 tools/fp6v2_stress.py 40 32"; SPKDIFF_V2_LAG=1 python tools/fp6v2_stress.py 40 32 2>&1 | tail -1
echo "== SPKDIFF_V2_WAVES=12 tools/fp6v2_stress.py 40 32"; SPKDIFF_V2_WAVES=12 python tools/fp6v2_stress.py 40 32 2>&1 | tail -1
echo "== tools/backward_stress.py 60 7"; python tools/backward_stress.py 60 7 2>&1 | tail -6
} > gpurun_out/r4_stress.txt 2>&1
cat gpurun_out/r4_stress.txt | cut -c1-200
