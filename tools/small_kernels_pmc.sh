R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tinv_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python $R/tools/vae_bench.py 1024 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- python $R/tools/vae_bench.py 1024 > $O/b.log 2>&1
python - $O <<'PY'
import csv, glob, sys, collections
for sub in "ab":
    f = glob.glob(sys.argv[1] + "/" + sub + "/*/*counter_collection.csv")
    if not f: print("no file", sub); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        key = "tinv<3,1>" if "tinv_lif_kernel<3, 1>" in n else "tinv<0,0>" if "tinv_lif_kernel<0, 0>" in n else "vq16" if "vq16" in n else "readout" if "readout_collapsed" in n else "gather2" if "conv_mfma_gather2" in n else None
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: sorted(v)[len(v)//2] for c, v in d.items()})
PY
