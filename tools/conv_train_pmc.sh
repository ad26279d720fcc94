# SQ counters of the native training convolutions on the largest layer (dec.convT2): two --pmc passes, no trace flags (usage: through gpurun)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/ct_pmc; rm -rf $O; mkdir -p $O
export CT_ONLY=dec.convT2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -- python $R/tools/conv_train_time.py 512 nolib > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- python $R/tools/conv_train_time.py 512 nolib > $O/sq2.log 2>&1
python - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
tot = collections.defaultdict(dict)
for d in ("sq1", "sq2"):
    f = glob.glob(f"{O}/{d}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "conv_train" in k and "reduce" not in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        tot[k].update({c: sorted(x)[len(x) // 2] for c, x in v.items()})
for k, c in tot.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"{k}: kernel {cyc:.0f} cycles/XCD, MFMA {c['SQ_INSTS_MFMA']:.0f} ({c['SQ_INSTS_MFMA'] / 1024:.0f} per SIMD), matrix pipe busy "
          f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:.2f}, VALU {c['SQ_INSTS_VALU'] / 1e6:.2f} M ({c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.1f} per MFMA), "
          f"SALU {c['SQ_INSTS_SALU'] / 1e6:.2f} M, LDS {c['SQ_INSTS_LDS'] / 1e6:.2f} M, VMEM {c['SQ_INSTS_VMEM'] / 1e6:.2f} M, wait_any / wave_cycles "
          f"{c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
PY
