"""Median per-launch FETCH_SIZE / WRITE_SIZE of the fp6v2 main and tail launches per (library, shape) under gpurun_out/r6_gx.
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 reports half the bytes of 16-B-per-lane streaming reads); both are KiB."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(root + "/*__*__*_SIZE")):
    lib, shape, ctr = os.path.basename(d).split("__")
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            kind = "main" if "conv3x3_fp6v2" in k else ("tail" if "fp6v2_tail_kernel" in k else None)
            if kind and row["Counter_Name"] == ctr:
                acc[(kind, row["Dispatch_Id"])] += float(row["Counter_Value"])
        per = collections.defaultdict(list)
        for (kind, _), v in acc.items():
            per[kind].append(v)
        for kind, v in per.items():
            v.sort()
            mb = v[len(v) // 2] * 1024 / 1e6 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
            res[f"{lib} {shape}"][f"{ctr}_{kind}_MB"] = round(mb, 1)
for k, d in sorted(res.items()):
    d["total_MB"] = round(sum(d.values()), 1)
    print(k, json.dumps(d))
