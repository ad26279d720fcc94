#!/bin/bash
# kernel trace of the reverse process with image elimination + position lists (tools/listed_time.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2lists
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/listed_time.py 256 2 > $O/listed_time.log 2>&1
tail -5 $O/listed_time.log
