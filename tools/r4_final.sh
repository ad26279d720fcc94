#!/bin/bash
# round 4, closing pass on one box: the whole GPU suite, the stress tools, the default bench line, the training line
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
O=gpurun_out/r4_final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1
grep -v PARITY_REPORT $O/gputest.log | tail -4 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
{ python tools/fp6v2_stress.py; python tools/vae_fp6_stress.py; python tools/modes_stress.py; python tools/backward_stress.py; } > $O/stress.log 2>&1
tail -12 $O/stress.log | cut -c1-250
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json
python bench.py --workload train --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train.json
cut -c1-250 $O/bench_train.json
