#!/bin/bash
# SQ counters of the weight-gradient launch at one layer shape: usage tools/dgrad_pmc.sh [Cout=512 Cin=256]
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/dgpmc; rm -rf $O; mkdir -p $O
export DGRAD_ONLY="${1:-512},${2:-256}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $O/a -- python $R/tools/dgrad_time.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -- python $R/tools/dgrad_time.py > $O/b.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --output-format csv -d $O/c -- python $R/tools/dgrad_time.py > $O/c.log 2>&1
python - $O <<'PY'
import collections, csv, glob, sys
for d in "abc":
    fs = glob.glob(f"{sys.argv[1]}/{d}/*/*counter_collection.csv")
    if not fs:
        print(d, "no output:", open(f"{sys.argv[1]}/{d}.log").read()[-400:]); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "dgrad3x3" in r["Kernel_Name"]:
            form = "f16x2 " if "true" in r["Kernel_Name"].split("(")[0] or "Lb1" in r["Kernel_Name"] else "bf16x3"
            acc[(form, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]} {k[1]:32s} {sorted(v)[len(v) // 2]:16.0f}  ({len(v)} launches)")
PY
