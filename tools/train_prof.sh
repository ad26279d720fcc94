R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/trainprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload train --eager-train --steps 10 --warmup 3 --no-cpu-baseline > $O/log 2>&1
tail -1 $O/log | cut -c1-300
python - "$(ls -t $O/trace/*/*_kernel_trace.csv | head -1)" $O/train_steady_state.md <<'PY'
# steady state only: the launches after the library's last solver-search kernel (naive_conv_*: MIOpen's find step in warm-up),
# cut into iterations at the loss kernel (one masked_ce_kernel launch per iteration)
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max((i for i, r in enumerate(rows) if "naive_conv" in r["Kernel_Name"]), default=-1)
rows = rows[last + 1:]
# an iteration = from one masked_ce_kernel launch (exactly one per iteration, in the loss) to the next
groups = [i for i, r in enumerate(rows) if "masked_ce_kernel" in r["Kernel_Name"]]
start, n_it = groups[0], len(groups) - 1
seg = rows[start:groups[-1]]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:64]
d = collections.OrderedDict()
for r in seg:
    d.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
out = [f"{n_it} steady-state iterations: {tot / n_it / 1e3:.3f} ms of kernels per iteration, {span / n_it / 1e3:.3f} ms wall, {len(seg) / n_it:.0f} launches", "",
       "| kernel | launches / iteration | avg us | ms / iteration | % |", "|---|---|---|---|---|"]
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:24]:
    out.append(f"| `{k}` | {len(v) / n_it:.1f} | {sum(v) / len(v):.1f} | {sum(v) / n_it / 1e3:.3f} | {sum(v) / tot * 100:.1f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
# the launch sequence of ONE iteration (the last whole one), in order
a, b = groups[-2] - start, groups[-1] - start
with open(sys.argv[2].replace(".md", "_sequence.txt"), "w") as f:
    for r in seg[a:b]:
        f.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {short(r['Kernel_Name'])}\n")
print("\n".join(out))
PY
rm -rf $O/trace
