"""CPU tests of bench.py's own rank launcher (VERDICT r1 'What's weak' #2): a plain ``python bench.py --gpus N`` must
start N ranks as a child ``torch.distributed.run``, relay rank 0's ONE JSON line and the child's exit code; a WORLD_SIZE
that disagrees with ``--gpus`` is a hard error.  The ranks run tests/_bench_stub_worker.py (gloo, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_bench_stub_worker.py")


def _run(extra_args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["SPKDIFF_BENCH_WORKER"] = STUB
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_plain_start_with_two_gpus_launches_two_ranks_and_relays_one_line():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--global-batch", "7"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                       # exactly one line on stdout: the result
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["job_key_shared"] is True, "'global' noise layout: every rank works under the job's ONE Philox key (rank 0's draw)"
    assert d["keys_distinct"] is True, "'rank' layout: ranks seeded alike must still draw distinct Philox keys"
    assert d["images"] == 7 and abs(d["max_t"] - 0.2) < 1e-12          # ragged shards gathered; max over ranks


def test_eight_ranks_one_line_and_exit_codes():
    """`bench.py --gpus 8` (the driver's largest run) under the stub worker: eight gloo ranks, ragged global batch, the
    rank-0-only tail against seven waiting ranks, ONE line with ranks_seen == 8, exit code 0; a failing rank -> non-zero."""
    p = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--global-batch", "8195"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["images"] == 8195 and d["keys_distinct"] is True and d["job_key_shared"] is True
    assert abs(d["max_t"] - 0.8) < 1e-12
    p = _run(["--gpus", "8"], {"STUB_FAIL": "1"}, timeout=600)
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_rank_failure_is_reported_as_failure():
    p = _run(["--gpus", "2"], {"STUB_FAIL": "1"})
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_world_size_mismatch_is_a_hard_error():
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr and p.stdout.strip() == ""


def test_refuses_more_gpus_than_the_node_has():
    # without the stub the launcher counts devices first (no GPU is initialised by counting): 0 here
    import pytest
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node really has 8 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SPKDIFF_BENCH_WORKER"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "exposes" in p.stderr and p.stdout.strip() == ""


def test_bench_oracle_fixture_check_logic():
    """bench.py's ``oracle_fixture`` object, host logic only: tokens equal to the committed F15 fixture give
    ok = True with the recorded fragile image reported; a batch whose key is not the fixture's is "not comparable"; a token changed in
    an image OUTSIDE the fragile set makes the check fail; the fragile image may carry the exact-convolution oracle's tokens."""
    import argparse
    import importlib.util
    import numpy as np
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    z = np.load(bench.F15_FIXTURE, allow_pickle=False)
    seed, warmup, steps, B, sample_steps, T, _ = (int(v) for v in z["config"])
    args = argparse.Namespace(sample_steps=sample_steps, T=T)
    first = torch.from_numpy(z["first_tokens_oracle"].astype(np.int64)).reshape(B, 1, 7, 7)
    last = torch.from_numpy(z["last_tokens_oracle"].astype(np.int64)).reshape(B, 1, 7, 7).clone()
    frag = [int(b) for b in z["last_fragile_images"]]
    for j, b in enumerate(frag):                                   # what the HIP path produces there: the exact-convolution tokens
        last[b] = torch.from_numpy(z["last_fragile_tokens_exact"][j].astype(np.int64)).reshape(1, 7, 7)
    keys = (int(z["first_key"][0]), int(z["last_key"][0]))
    r = bench.oracle_fixture_check(args, B, {"first": (first, keys[0]), "last": (last, keys[1])})
    assert r["applicable"] and r["ok"], r
    assert r["first_timed_batch"]["images_equal"] == f"{B}/{B}"
    assert r["last_timed_batch"]["images_equal"] == f"{B - len(frag)}/{B}"
    assert all(f["equals_exact_convolution_oracle"] for f in r["last_timed_batch"]["fragile"])
    # another key (another seed / --steps / --warmup): nothing to compare with
    r2 = bench.oracle_fixture_check(args, B, {"first": (first, keys[0] + 1), "last": (last, keys[1] + 1)})
    assert not r2["applicable"] and not r2["first_timed_batch"]["comparable"]
    # a wrong token outside the fragile set
    bad = first.clone()
    victim = next(i for i in range(B) if i not in frag)
    bad[victim, 0, 3, 3] = (bad[victim, 0, 3, 3] + 1) % 128
    r3 = bench.oracle_fixture_check(args, B, {"first": (bad, keys[0]), "last": (last, keys[1])})
    assert r3["applicable"] and not r3["ok"] and r3["first_timed_batch"]["images_differing_outside_the_fragile_set"] == [victim]
    # another batch size than the fixture's
    assert not bench.oracle_fixture_check(args, 128, {"first": (first[:128], keys[0]), "last": (last[:128], keys[1])})["applicable"]
