#!/bin/bash
# kernel durations (rocprofv3 trace) of tools/wgrad_time.py, per launch shape: usage tools/wgrad_prof.sh
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/wgprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/wgrad_time.py > $O/log 2>&1
python - "$(ls -t $O/trace/*/*_kernel_trace.csv | head -1)" <<'PY'
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.OrderedDict()
for m, r in zip(rows[0::2], rows[1::2]):              # (multiply launch, reduce launch) pairs; the reduce grid identifies the layer
    assert "reduce" in r["Kernel_Name"] and "reduce" not in m["Kernel_Name"]
    t = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
    d.setdefault((m.get("Grid_Size_X", m.get("Grid_Size")), r.get("Grid_Size_X", r.get("Grid_Size"))), []).append((t(m), t(r)))
for k, v in d.items():
    med = lambda i: sorted(x[i] for x in v)[len(v) // 2]
    print(f"grid {k[0]:>7s} / reduce grid {k[1]:>8s}: {len(v):3d} calls, multiply {med(0):7.1f} us, reduce {med(1):6.1f} us")
PY
rm -rf $O/trace
