"""The path's one collective on a HIP device: multi-GPU hardware is not available to the
suite, so the least that can be proven is that the RCCL path loads and moves bytes.  A child process opens a one-rank ``nccl``
process group (``torch.distributed``'s "nccl" backend IS RCCL on ROCm), runs ``spkdiff.dist.sample_images_sharded`` /
``gather_images`` (equal and padded branch) / ``global_token_checksum`` (device int64 all-reduce) and the key broadcast on real
sampler output, and destroys the group.  The RCCL version banner goes into the session's PARITY_REPORT line."""
import json
import os
import re
import subprocess
import sys

import pytest

from parity_report import record

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_rccl_one_rank_collectives_on_the_device():
    env = dict(os.environ)
    env["NCCL_DEBUG"] = "VERSION"                             # RCCL prints its version banner at communicator creation
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(29541 + os.getpid() % 200)
    p = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_one_rank_worker.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    text = p.stdout + "\n" + p.stderr
    assert p.returncode == 0, text[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RCCL_ONE_RANK ")]
    assert line, text[-4000:]
    res = json.loads(line[-1][len("RCCL_ONE_RANK "):])
    banner = [ln.strip() for ln in text.splitlines() if re.search(r"(RCCL|NCCL) version", ln)]
    record("rccl_one_rank", banner=(banner[0] if banner else None), **res)
    assert res["backend"] == "nccl" and res["world"] == 1
    for k in ("gather_equal", "gather_padded", "checksum_allreduce", "key_broadcast", "tokens_equal_unsharded", "destroyed"):
        assert res[k] is True, (k, res)


def _bench_child(extra, env_extra=None, timeout=900):
    root = os.path.dirname(HERE)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SPKDIFF_BENCH_WORKER"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    # a FRESH child process (never a re-exec of this one, which has initialised the GPU): bench.py's own launcher starts the ranks
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=root)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_two_rank_bench_on_one_device_gives_the_one_rank_job():
    """the REAL bench.py with two ranks -- its launcher (python -m torch.distributed.run as a child), the shard
    offsets, the image gather, `global_token_checksum` and the job-key check meeting each other on a device, not on stubs.  The
    suite's boxes have one GPU, so both ranks share it and the collectives go over gloo (SPKDIFF_BENCH_SHARE_GPU=1; RCCL refuses
    two ranks on one device; the RCCL transport itself is test_rccl_one_rank_collectives_on_the_device).  The same job -- seed,
    steps, warm-up, GLOBAL batch -- as ONE rank must give the same global_token_checksum: the sample does not depend on the split
    (R/snn_model/vq_diffusion.py:103-142 per image)."""
    common = ["--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--dense-only"]
    p2, d2 = _bench_child(["--gpus", "2", "--batch", "48"] + common, {"SPKDIFF_BENCH_SHARE_GPU": "1"})
    assert p2.returncode == 0 and d2 is not None, (p2.stdout[-1500:], p2.stderr[-3000:])
    assert len([ln for ln in p2.stdout.splitlines() if ln.strip()]) == 1, "exactly one line on stdout"
    assert d2["n_gpus"] == 2 and d2["ranks_seen"] == 2 and d2["steps"] == 1 and d2["warmup"] == 1
    assert d2["config"]["global_batch"] == 96 and d2["scaling"] == "weak"
    assert d2["job_key_shared"] is True and d2["rank_token_checksums_distinct"] is True
    assert "ONE device" in d2["config"]["sharding"]
    p1, d1 = _bench_child(["--gpus", "1", "--batch", "96"] + common)
    assert p1.returncode == 0 and d1 is not None, (p1.stdout[-1500:], p1.stderr[-3000:])
    assert d1["ranks_seen"] == 1 and d1["config"]["global_batch"] == 96
    record("bench_two_ranks_one_device", ranks_seen=d2["ranks_seen"], global_batch=96, checksum_two_ranks=d2["global_token_checksum"],
           checksum_one_rank=d1["global_token_checksum"], job_key_shared=d2["job_key_shared"],
           rank_token_checksums_distinct=d2["rank_token_checksums_distinct"])
    assert d2["global_token_checksum"] == d1["global_token_checksum"], "the job's tokens depend on how the batch was split over ranks"
