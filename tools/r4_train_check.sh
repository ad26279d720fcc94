# round 4: the training iteration after the weight-prep / ticketed BN-LIF / packed-spike hand-over changes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traincheck; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_prep or ticketed or bn_lif or train or wgrad or dgrad or data_gradient or packing" > $O/tests.log 2>&1
tail -5 $O/tests.log
for i in 1 2 3; do python bench.py --workload train --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200; done > $O/bench_train.log
cat $O/bench_train.log
bash tools/train_prof.sh > $O/prof.log 2>&1
cp $R/gpurun_out/trainprof/train_steady_state.md $O/ 2>/dev/null
head -12 $O/train_steady_state.md
