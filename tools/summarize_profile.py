"""Summarise a rocprofv3 --kernel-trace CSV into a per-kernel / per-denoiser-layer table (markdown on stdout).

usage: python tools/summarize_profile.py <kernel_trace.csv> [<pmc_fetch counter csv> <pmc_write counter csv>]
The MFMA kernel (fp6: conv3x3_fp6_kernel, int8: conv3x3_mfma_kernel<.., LIF, ..>) is one symbol for conv2..conv5; the
layers are told apart by dispatch order (4 consecutive dispatches per denoiser call)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = defaultdict(list)
k = 0
layer = 0
for r in rows:
    name = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "conv3x3_mfma_kernel<" in name:
        targs = name.split("conv3x3_mfma_kernel<")[1].split(">")[0].split(",")
        if int(targs[2]) == 0:          # template <NT, NPA, MODE, DBG>: MODE 0 = LIF (conv2..conv5, in launch order)
            name = f"conv3x3_mfma_kernel<LIF> den.conv{2 + k % 4}"
            k += 1
        else:
            name = "conv3x3_mfma_kernel<MEAN>"
    elif "conv3x3_fp6_kernel" in name:      # one symbol for conv2..conv5, in launch order
        name = f"conv3x3_fp6_kernel den.conv{2 + k % 4}"
        k += 1
    elif ("conv3x3_fp6v2_kernel" in name or "conv3x3_fp6v2_lag_kernel" in name):    # second-generation kernel: same order; its two tail launches follow each one
        name = f"conv3x3_fp6v2_kernel den.conv{2 + k % 4}"
        layer = 2 + k % 4
        k += 1
    elif "fp6v2_tail_kernel" in name:     # <H, W, PART>: 0 repair + last position, 1 repair only, 2 last position only
        part = name.split("fp6v2_tail_kernel<")[1].split(">")[0].split(",")[-1].strip()
        name = f"fp6v2_tail_kernel<{ {'0': 'repair+last position', '1': 'repair', '2': 'last position'}.get(part, part) }> den.conv{layer}"
    elif "fp6v2_fixup_kernel" in name:
        name = f"fp6v2_fixup_kernel den.conv{layer}"
    elif "fp6v2_lastpos_kernel" in name:
        name = f"fp6v2_lastpos_kernel den.conv{layer}"
    elif "conv3x3_counts_mfma_kernel" in name:
        name = "conv3x3_counts_mfma_kernel den.conv6 (time-collapsed)"
    dur[name.replace("(anonymous namespace)::", "")[:90]].append(d)
tot = sum(sum(v) for v in dur.values())
print("| kernel | calls | avg us | min us | max us | total ms | % |")
print("|---|---|---|---|---|---|---|")
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"| `{n}` | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.1f} |")
for path in sys.argv[2:]:
    cr = list(csv.DictReader(open(path)))
    acc = defaultdict(list)
    for r in cr:
        acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print()
    for (n, c), v in acc.items():
        print(f"- `{n}` {c}: mean {sum(v)/len(v):.4g} over {len(v)} dispatches")
