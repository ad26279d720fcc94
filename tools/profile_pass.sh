#!/bin/bash
# The round's profile pass (run through gpurun from the repo root; usage: tools/profile_pass.sh <tag>, e.g. r5): kernel trace of the
# bench command + PMC passes on the dominant kernel (den.conv4 shape, fp6v2 kernel; round 5 adds the den.conv5 shape and the 8x8
# conv4 shape of BASELINE configs[3]), on the decoder convT2 launch of the encode->decode workload, on the LIF scan and on the fused
# step tail.  Each --pmc pass is its own run, never combined with a trace flag.  Outputs under gpurun_out/<tag>prof/;
# tools/refresh_profiles.py <tag> copies them into profiles/ and derives the JSON summaries.
R=$GRAFT_REPO_ROOT
TAG=${1:-r6}
O=$R/gpurun_out/${TAG}prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_prof.json 2> $O/bench_under_prof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/v2_$c -- python $R/tools/fp6v2_one.py 512 256 13 > $O/v2_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/v2c5_$c -- python $R/tools/fp6v2_one.py 256 512 13 > $O/v2c5_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/v2cifar_$c -- python $R/tools/fp6v2_one.py 512 256 9 cifar > $O/v2cifar_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/encdec_$c -- python $R/tools/convt_time.py 1024 5 > $O/encdec_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/lif_$c -- python $R/tools/lif_bench.py > $O/lif_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F6F4 GRBM_GUI_ACTIVE --output-format csv -d $O/v2_sq -- python $R/tools/fp6v2_one.py 512 256 13 > $O/v2_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/v2_sq2 -- python $R/tools/fp6v2_one.py 512 256 13 > $O/v2_sq2.log 2>&1
ls -R $O | grep -c csv
head -c 300 $O/bench_under_prof.json
# a second trace of the headline measurement alone (per-layer table of profiles/README.md)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --dense-only > $O/bench_headline_under_prof.json 2> $O/bench_headline_under_prof.err
# trace of the encode->decode workload (configs[2]) and of the reverse process with elimination + position lists
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_encdec -- python $R/bench.py --workload encdec --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_encdec_under_prof.json 2> $O/bench_encdec_under_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_lists -- python $R/tools/listed_time.py 256 2 > $O/listed_time.log 2>&1
# round 3+: vector-instruction count of the VQ-VAE's dominant launch (the `bound: valu` roofline of the encdec object) and the
# fused step tail's traffic / instruction mix
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/encdec_sq -- python $R/tools/convt_time.py 1024 5 > $O/encdec_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/tail_$c -- python $R/tools/tail_time.py --child > $O/tail_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/tail_sq -- python $R/tools/tail_time.py --child > $O/tail_sq.log 2>&1
ls -R $O | grep -c csv
# round 6: R/main.py's own call shape (B = 16, 49 steps + module-API decode)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_mainpy -- python $R/bench.py --workload-main-py-shape --steps 20 --no-cpu-baseline > $O/bench_mainpy_under_prof.json 2> $O/bench_mainpy_under_prof.err
