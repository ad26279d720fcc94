#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_convt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a -- python $R/tools/convt_time.py 1024 3 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU --output-format csv -d $O/b -- python $R/tools/convt_time.py 1024 3 > $O/b.log 2>&1
python - <<'PY'
import csv, glob, collections, os
for d in ("a", "b"):
    f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_convt", d, "*", "*_counter_collection.csv"))
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "convT_s2_fp6_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, sorted(v)[len(v) // 2], len(v))
PY
