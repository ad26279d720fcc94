#!/bin/bash
# One job, split two ways, on a 1-GPU box (SPKDIFF_BENCH_SHARE_GPU=1: every rank on device 0, gloo): the global token checksum of
# a 512-image job must be the same number as 1 x 512, 2 x 256 and 4 x 128 ranks (noise_layout 'global').
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
cd $R
for n in 1 2 4; do
  SPKDIFF_BENCH_SHARE_GPU=1 python bench.py --gpus $n --global-batch 512 --steps 2 --warmup 1 --dense-only --no-extras > gpurun_out/r4_split_$n.json 2> gpurun_out/r4_split_$n.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r4_split_$n.json"))
print("ranks", d["n_gpus"], "seen", d["ranks_seen"], "global batch", d["config"]["global_batch"], "images/s", round(d["value"], 1),
      "global_token_checksum", d["global_token_checksum"], "cpu_baseline", "cpu_baseline" in d and round(d["cpu_baseline"]["value"], 2))
PY
done
