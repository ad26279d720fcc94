#!/bin/bash
# kernel trace of the encode->decode workload (BASELINE configs[2])
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2encdec
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --workload encdec --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python $R/tools/trace_layers.py $(ls -t $O/trace/*/*_kernel_trace.csv | head -1)
