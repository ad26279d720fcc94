#!/bin/bash
# Round profile pass (run through gpurun from the repo root): kernel trace of the bench command + PMC passes on the
# dominant kernel (den.conv4 shape, fp6 MFMA kernel).  Outputs under gpurun_out/prof6/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_prof.json 2> $O/bench_under_prof.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/fp6_one.py 512 256 13 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/fp6_one.py 512 256 13 > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F6F4 GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python $R/tools/fp6_one.py 512 256 13 > $O/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/sq2 -- python $R/tools/fp6_one.py 512 256 13 > $O/sq2.log 2>&1
ls -R $O | grep -c csv
cat $O/bench_under_prof.json | head -c 400
