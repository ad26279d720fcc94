"""CPU test of the N>1 path: two processes, gloo backend, 127.0.0.1 rendezvous.  The sampling batch is
sharded by rank, each rank produces its shard (a deterministic stand-in for the GPU generator: the compute path needs a
ROCm device), and ONE all-gather returns every image on every rank in sample order -- equal and ragged shards."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_images(lo, hi):
    idx = torch.arange(lo, hi, dtype=torch.int64).view(-1, 1, 1, 1)
    yy = torch.arange(28).view(1, 1, 28, 1)
    xx = torch.arange(28).view(1, 1, 1, 28)
    return ((idx * 7 + yy * 3 + xx) % 251).to(torch.uint8)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from spkdiff import dist as sdist
    import torch.distributed as dist
    r, lr, w = sdist.init("gloo")
    assert (r, w) == (rank, world)
    calls = []

    def gen(lo, hi):
        calls.append((lo, hi))
        return _fake_images(lo, hi)

    out = sdist.sample_images_sharded(gen, total)
    ok = out.dtype == torch.uint8 and torch.equal(out, _fake_images(0, total)) and len(calls) == 1
    lo, hi = sdist.shard_range(total, rank, world)
    ok = ok and calls[0] == (lo, hi)
    # timing reduction used by bench.py: max over ranks
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == float(world)
    # 'global' noise layout (host logic; the draws themselves need a device): differently seeded ranks end up with rank 0's
    # key (sync_key), every rank's counter offsets are those of its images in the whole job, and the per-shard token checksums
    # add up to the whole job's
    from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel
    ab = AbsorbingDiffusion(DummyModel(1, 128), mask_id=128)
    # a sampler that never declared a shard takes NO collective inside a process group (rank 0 alone may call it: a preview
    # during DDP training must not hang) and folds the rank into its key (ranks seeded alike still draw distinct images)
    import warnings
    torch.manual_seed(5)
    with warnings.catch_warnings(record=True) as wrn:
        warnings.simplefilter("always")
        if rank == 0:
            ab._philox_key()                                 # only rank 0: would block forever if it were a broadcast
        torch.manual_seed(5)
        k_un = ab._philox_key()
    ok = ok and any("set_shard" in str(x.message) for x in wrn)
    kun = [None] * world
    dist.all_gather_object(kun, k_un)
    ok = ok and len(set(kun)) == world
    # sample_images_sharded(..., sampler=ab) declares the shard itself
    sdist.sample_images_sharded(lambda a, b: _fake_images(a, b), total, sampler=ab)
    ok = ok and ab._shard_set and ab.global_first == lo and ab.n_samples == hi - lo
    torch.manual_seed(100 + rank)
    key = ab._philox_key()
    keys = [None] * world
    dist.all_gather_object(keys, key)
    torch.manual_seed(100)
    ok = ok and len(set(keys)) == 1 and keys[0] == int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64))
    ok = ok and ab._step_offset(3, hi - lo, 7, 7, 128) == 3 * (1 << 40) + lo * 49 * 128
    ab.noise_layout = 'rank'
    ok = ok and ab._step_offset(3, hi - lo, 7, 7, 128) == 3 * (hi - lo) * 49 * 128
    torch.manual_seed(100)
    k_rank = ab._philox_key()
    allk = [None] * world
    dist.all_gather_object(allk, k_rank)
    ok = ok and len(set(allk)) == world                     # rank folded into the key: distinct streams
    toks = torch.randint(0, 128, (total, 1, 7, 7), generator=torch.Generator().manual_seed(9))
    whole = sdist.token_checksum(toks, 0) & 0x7FFFFFFFFFFFFFFF
    ok = ok and sdist.global_token_checksum(toks[lo:hi], lo) == whole
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("total", [8, 7])
def test_sharded_sampling_two_ranks_gloo(total):
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spiking-diffusion_amd")
    os.environ["PYTHONPATH"] = pkg + os.pathsep + os.environ.get("PYTHONPATH", "")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_shard_ranges_cover_everything():
    from spkdiff import dist as sdist
    for total in (1, 7, 8, 8192, 8191):
        for world in (1, 2, 4, 8):
            spans = [sdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.env_world()[2] >= 1
    img = torch.zeros(3, 1, 28, 28, dtype=torch.uint8)
    assert sdist.gather_images(img) is img          # no process group: identity
