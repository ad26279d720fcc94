#!/bin/bash
# closing pass of a round on one box (usage tools/final_pass.sh [tag=r5]): the whole GPU suite, smoke, the DRIVER's bench command
# (--steps 20 --warmup 5: the job tests/golden/f15_bench_job_tokens.npz holds), the training line
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
TAG=${1:-r6}
O=gpurun_out/${TAG}_final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1
grep -v PARITY_REPORT $O/gputest.log | tail -4 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cut -c1-400 $O/bench_default.json
python bench.py --workload train --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train.json
cut -c1-250 $O/bench_train.json
python bench.py --workload train_vqvae --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_train_vqvae.json
cut -c1-250 $O/bench_train_vqvae.json
python tools/small_batch_time.py > $O/small_batch.txt 2>&1; cat $O/small_batch.txt
