"""Per-kernel / per-layer averages of a rocprofv3 kernel trace of the sampler (fp6v2 main / listed kernels and their tails
by launch order).  usage: python tools/trace_layers.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
k = 0
last = ("?", 0)
for r in rows:
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "conv3x3_fp6v2_listed" in n:
        last = ("listed", 2 + k % 4); acc[last].append(d); k += 1
    elif ("conv3x3_fp6v2_kernel" in n or "conv3x3_fp6v2_lag_kernel" in n):
        last = ("main", 2 + k % 4); acc[last].append(d); k += 1
    elif "fp6v2_fixup" in n: acc[(last[0] + "-fixup", last[1])].append(d)
    elif "fp6v2_tail" in n: acc[(last[0] + "-tail", last[1])].append(d)
    elif "fp6v2_lastpos" in n: acc[(last[0] + "-lastpos", last[1])].append(d)
    else:
        nm = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("::")[-1][:48]
        acc[(nm, 0)].append(d)
for key in sorted(acc, key=lambda x: (x[0], x[1])):
    v = acc[key]
    print(f"{key[0]:34s} {key[1]:2d} n={len(v):5d} avg {sum(v) / len(v):7.1f} min {min(v):7.1f} max {max(v):7.1f}")
