# kernel tables of the VQ-VAE training iteration and of the 8x8 (CIFAR-shaped) diffusion training iteration
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for what in vqvae diff8x8; do
O=$R/gpurun_out/trainother/$what; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/train_other_prof.py $what 12 > $O/log 2>&1
tail -2 $O/log
python - "$(ls -t $O/trace/*/*_kernel_trace.csv | head -1)" $O/table.md $what <<'PY'
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# iterations: cut at a kernel that runs exactly once per iteration (the loss kernel / the codebook's gradient)
key = "masked_ce_kernel" if sys.argv[3] == "diff8x8" else "vq_train_bwd_kernel"
cuts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
n_it = 8
seg = rows[cuts[-n_it - 1]:cuts[-1]]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:80]
d = collections.OrderedDict()
for r in seg:
    d.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
out = [f"{sys.argv[3]}: {n_it} steady-state iterations, {tot / n_it / 1e3:.3f} ms of kernels per iteration, {len(seg) / n_it:.0f} launches", "",
       "| kernel | launches / iteration | avg us | ms / iteration | % |", "|---|---|---|---|---|"]
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:30]:
    out.append(f"| `{k}` | {len(v) / n_it:.1f} | {sum(v) / len(v):.1f} | {sum(v) / n_it / 1e3:.3f} | {sum(v) / tot * 100:.1f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:24]))
PY
done
