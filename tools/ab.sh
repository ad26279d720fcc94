#!/bin/bash
# A/B on ONE box: for every library variant given (paths relative to the repo root), the per-layer fp6v2 launch times with the
# bit-equality check against the six-plane kernel (tools/fp6v2_time.py) and the dense reverse process (tools/listed_time.py).
# usage: tools/ab.sh <lib.so> [<lib.so> ...]     (two passes, so that drift of the box shows)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
for pass in 1 2; do
  for lib in "$@"; do
    echo "== pass $pass $lib"
    SPKDIFF_LIB=$R/$lib python $R/tools/fp6v2_time.py $R/$lib
    SPKDIFF_LIB=$R/$lib python $R/tools/listed_time.py 256 3 dense
  done
done
