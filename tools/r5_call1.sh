#!/bin/bash
# round 5, first GPU call: whole GPU suite on the new tree, the F15 fixture of the bench line's own job, the start-of-round line
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R && mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "not f15" 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/r5_call1_pytest.txt
python oracle/gen_f15_bench_job.py 2>&1 | grep -v amdgpu.ids | tail -20 > gpurun_out/r5_call1_f15.txt
python bench.py 2>gpurun_out/r5_bench_start.err | tail -1 > gpurun_out/r5_bench_start.json
tail -3 gpurun_out/r5_call1_pytest.txt; cat gpurun_out/r5_call1_f15.txt; cut -c1-600 gpurun_out/r5_bench_start.json
