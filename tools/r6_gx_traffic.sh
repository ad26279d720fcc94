#!/bin/bash
# L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the den.conv5 and den.conv4 launches for libspkdiff variants that
# differ in how many channel groups an XCD keeps (SPK_V2_GX_KB): tools/r6_gx_traffic.sh <lib.so> ...   
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_gx
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  for shape in "256 512" "512 256"; do
    s=${shape// /x}
    for c in FETCH_SIZE WRITE_SIZE; do
      SPKDIFF_LIB=$R/$lib rocprofv3 --pmc $c --output-format csv -d $O/${n}__${s}__$c -- python $R/tools/fp6v2_one.py $shape 9 > $O/${n}__${s}__$c.log 2>&1
    done
  done
done
python3 $R/tools/r6_gx_traffic_summary.py $O
