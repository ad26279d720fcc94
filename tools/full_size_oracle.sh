#!/bin/bash
# Once per round: the bench line's own job, B = 256 x 100 reverse steps, against the CPU oracle.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
mkdir -p $R/gpurun_out
cd $R && SPKDIFF_RUN_SLOW=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "bench_line_job_full_size" 2>&1 | grep -v "amdgpu.ids" | tee $R/gpurun_out/full_size_oracle.txt
