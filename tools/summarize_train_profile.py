"""Steady-state per-step kernel table of a rocprofv3 --kernel-trace CSV of `bench.py --workload train` (markdown).

usage: python tools/summarize_train_profile.py <kernel_trace.csv> <timed steps>
The first calls contain the library's convolution algorithm search; only the last <timed steps> iterations are kept
(an iteration starts at its first bn_stats launch: 5 per iteration)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bn_stats" in r["Kernel_Name"]]
sel = rows[idx[-5 * steps]:]
d = defaultdict(list)
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    d[n.split("(")[0][:72]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
wall = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print(f"last {steps} iterations: {tot / steps:.0f} us of kernels per iteration, {wall / steps:.0f} us wall, "
      f"{len(sel) / steps:.0f} launches per iteration\n")
print("| kernel | launches / iter | us / iter | % | avg us |")
print("|---|---|---|---|---|")
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"| `{n}` | {len(v) / steps:.1f} | {sum(v) / steps:.1f} | {100 * sum(v) / tot:.1f} | {sum(v) / len(v):.1f} |")
