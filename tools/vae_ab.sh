#!/bin/bash
# same-box A/B of library variants on the encode -> decode path (BASELINE configs[2], B = 1024): end-to-end rate (alternating passes),
# the vae_fp6 launches under rocprofv3 --kernel-trace --stats, and the vae_fp6-against-int8 equality tests per variant.
# usage: [VAE_AB_KERNELS=vae_fp6,tinv_lif,...] tools/vae_ab.sh <label>=<lib.so | default> ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/vae_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ps in 1 2; do
  for spec in "$@"; do
    l=${spec%%=*}; lib=${spec#*=}
    if [ "$lib" = default ]; then unset SPKDIFF_LIB; else export SPKDIFF_LIB=$R/$lib; fi
    echo "== pass $ps $l: $(python $R/tools/vae_bench.py 1024 2>/dev/null | head -1)"
  done
done
for spec in "$@"; do
  l=${spec%%=*}; lib=${spec#*=}
  if [ "$lib" = default ]; then unset SPKDIFF_LIB; else export SPKDIFF_LIB=$R/$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$l -- python $R/tools/vae_bench.py 1024 > $O/$l.log 2>&1
  echo "== $l kernels (avg us):"
  python - $O/$l <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    import os
    if any(k in r["Name"] for k in os.environ.get("VAE_AB_KERNELS", "vae_fp6").split(",")):
        print("   %-60s calls %4s avg %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  (cd $R && python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "vae_fp6 or f3_ or f4_ or f2_" 2>&1 | grep -v PARITY | tail -1)
done
