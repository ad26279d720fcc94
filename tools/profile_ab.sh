#!/bin/bash
# A/B kernel traces of tools/listed_time.py with two libraries on one box: usage tools/profile_ab.sh <libA.so> [<libB.so>]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  O=$R/gpurun_out/ab_$(basename $lib .so)
  rm -rf $O; mkdir -p $O
  SPKDIFF_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/listed_time.py 256 2 > $O/log 2>&1
  echo "== $lib"; grep "dense" $O/log
  python $R/tools/trace_layers.py $(ls -t $O/trace/*/*_kernel_trace.csv | head -1) | grep "^main\|counts\|tinv"
done
