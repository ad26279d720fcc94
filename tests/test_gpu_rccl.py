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
