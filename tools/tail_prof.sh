#!/bin/bash
# kernel durations (rocprofv3 trace) of tools/tail_time.py for each variant library: usage tools/tail_prof.sh <lib.so> ...
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  O=$R/gpurun_out/tailprof_$(basename $lib .so)
  rm -rf $O; mkdir -p $O
  SPKDIFF_LIB=$R/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/tail_time.py --child > $O/log 2>&1
  echo "== $lib"
  python - "$(ls -t $O/trace/*/*_kernel_stats.csv | head -1)" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("step_tail", "counts_mfma", "psample", "tinv_lif")):
        print(f"   {n[:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f}")
PY
  rm -rf $O/trace
done
