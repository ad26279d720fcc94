#!/bin/bash
# kernel trace of R/main.py's own call shape (bench.py --workload-main-py-shape: B = 16, 49 reverse steps + module-API decode)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_mainpy_trace${1:-}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/bench.py --workload-main-py-shape --steps 20 --no-cpu-baseline > $O/line.json 2> $O/err.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
head -40 "$f"
