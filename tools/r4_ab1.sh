#!/bin/bash
# round 4, experiment batch 1 (one box): accumulators in VGPRs, merged tail launch for full batches, flag counts of a
# three-digit tier (bound x 32)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
V=spiking-diffusion_amd/spkdiff/variants
mkdir -p $R/gpurun_out
{
  bash $R/tools/ab.sh $V/lib_base.so $V/lib_vgpr.so $V/lib_merge.so $V/lib_vgpr_merge.so
  for l in lib_base lib_spare32; do echo "== flag stats $l"; SPKDIFF_LIB=$R/$V/$l.so python $R/tools/flag_stats.py; done
} > $R/gpurun_out/r4_ab1.log 2>&1
tail -50 $R/gpurun_out/r4_ab1.log
