#!/bin/bash
# Kernel trace of the dense reverse process (tools/listed_time.py 256 2 dense) with per-layer averages; optional env prefix
# usage: tools/profile_dense.sh <tag>      (SPKDIFF_NO_TAIL=1 tools/profile_dense.sh notail)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
tag=${1:-dense}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/listed_time.py 256 2 dense > $O/log 2>&1
grep "dense" $O/log
python $R/tools/trace_layers.py $(ls -t $O/trace/*/*_kernel_trace.csv | head -1) | tee $O/layers.txt
cp $(ls -t $O/trace/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
