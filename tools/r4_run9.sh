#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
python -m pytest tests -m gpu -q -x -k "train or wgrad or dgrad or bn_lif or f9 or f10 or packing or small_input or graphed or codebook" 2>&1 | tail -8 > gpurun_out/r4_gputest9.log
grep -v PARITY gpurun_out/r4_gputest9.log | tail -6 | cut -c1-300
bash tools/train_prof.sh 2>&1 | tail -30
python bench.py --workload train --steps 30 --warmup 5 --no-cpu-baseline | cut -c1-330
python tools/wgrad_time.py 2>&1 | grep -v amdgpu | head -3; python tools/dgrad_time.py 2>&1 | grep -v amdgpu | head -3
