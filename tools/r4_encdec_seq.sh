# the launch sequence of one encode->decode iteration (kernel + memory-copy trace), to see what the copy launches are
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/encdec_seq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python $R/tools/vae_bench.py 1024 > $O/log 2>&1
python - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
k = glob.glob(O + "/trace/*/*_kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(k))]
m = glob.glob(O + "/trace/*/*_memory_copy_trace.csv")
if m:
    for r in csv.DictReader(open(m[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?"))))
rows.sort()
# last iteration: from the last tinv_lif_kernel<3 (enc1) on
starts = [i for i, r in enumerate(rows) if "tinv_lif_kernel<3" in r[2]]
seg = rows[starts[-1] - 3:]
with open(O + "/sequence.txt", "w") as f:
    for s, e, n in seg:
        f.write(f"{(e - s) / 1e3:8.1f} us  {n}\n")
print(open(O + "/sequence.txt").read())
PY
