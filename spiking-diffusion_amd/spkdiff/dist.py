"""Multi-GPU sampling: one process per GPU, the sampling batch is sharded by rank, every rank runs the whole
reverse diffusion + decode locally, and ONE all-gather of the finished uint8 images is the only collective
(RCCL over xGMI when the backend is "nccl").  Samples are independent -- eval BN uses running
statistics, LIF / VQ are per position -- so there is no other exchange step on this path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def init(backend: str | None = None):
    """Initialise torch.distributed from the environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE). Idempotent."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of ``total`` samples for ``rank``; the first ``total % world`` ranks get one more."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_images(local_u8: torch.Tensor, total: int | None = None) -> torch.Tensor:
    """All-gather uint8 images [b_local, C, H, W] from every rank into [total, C, H, W] in rank order.

    Equal shards use a single ``all_gather_into_tensor``; ragged shards are padded to the largest shard so that
    it is still one collective.  With world_size == 1 (or no process group) it returns the input."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_u8
    world = dist.get_world_size()
    if total is None:
        total = local_u8.shape[0] * world
    sizes = [shard_range(total, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    send = local_u8.contiguous()
    if send.shape[0] != bmax:
        pad = torch.zeros((bmax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat((send, pad), 0)
    if send.is_cuda and dist.get_backend() == 'gloo':
        # gloo (CPU tests; several ranks sharing ONE device as a testing aid: RCCL refuses that) stages device tensors itself, and with work
        # still pending on the stream that staging took SECONDS per call when two processes shared a device (measured: 6.1 s per bench step
        # against 0.2 s): hand it host tensors -- the images are 784 bytes each
        host = torch.empty((world * bmax,) + tuple(send.shape[1:]), dtype=send.dtype)
        dist.all_gather_into_tensor(host, send.cpu())
        out = host.to(send.device)
    else:
        out = torch.empty((world * bmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(out, send)
    if all(hi - lo == bmax for lo, hi in sizes):
        return out
    parts = [out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def sample_images_sharded(generate_local, total: int, sampler=None):
    """``generate_local(lo, hi)`` -> uint8 [hi-lo, C, H, W] on this rank's device; returns all ``total`` images
    on every rank (rank order == sample order).

    ``sampler``: the ``AbsorbingDiffusion`` that ``generate_local`` samples from.  When given, ``sampler.set_shard(lo, hi - lo)``
    is called here, which is what makes the ranks one job ('global' noise layout: shared key, counters on the GLOBAL image index --
    the images do not depend on the split).  Without it ``generate_local`` MUST call ``set_shard(lo, hi - lo)`` itself: a sampler
    that never did draws per-rank noise (and warns)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = shard_range(total, rank, world)
    if sampler is not None:
        sampler.set_shard(lo, hi - lo)
    return gather_images(generate_local(lo, hi), total)


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around arithmetic)."""
    x = (x ^ ((x >> 30) & 0x3FFFFFFFF)) * -4658895280553007687        # 0xBF58476D1CE4E5B9
    x = (x ^ ((x >> 27) & 0x1FFFFFFFFF)) * -7723592293110705685       # 0x94D049BB133111EB
    return x ^ ((x >> 31) & 0x1FFFFFFFF)


def token_checksum(tokens: torch.Tensor, first: int = 0) -> int:
    """Checksum of the tokens of images [first, first + n) of a job, as a sum over images of a hash of (GLOBAL image index,
    tokens): the partial sums of any split of the job add up (mod 2^64) to the same number, so an N-rank run, a 1-rank run
    and the oracle can be compared by one integer.  ``tokens``: integer tensor [n, ...] on any device."""
    t = tokens.reshape(tokens.shape[0], -1).to(torch.int64)
    n, m = t.shape
    w = _mix64(torch.arange(1, m + 1, dtype=torch.int64, device=t.device)) | 1
    row = ((t + 1) * w).sum(1)
    idx = torch.arange(first, first + n, dtype=torch.int64, device=t.device)
    return int(_mix64(row ^ _mix64(idx + 0x1234567)).sum().item())


def global_token_checksum(tokens_local: torch.Tensor, first: int) -> int:
    """``token_checksum`` of the whole job: every rank contributes the partial sum of its shard (one int64 all-reduce, outside
    any timed region).  Reported as a non-negative 63-bit integer."""
    part = token_checksum(tokens_local, first)
    if dist.is_available() and dist.is_initialized():        # (also at one rank: the collective path is the tested path)
        dev = tokens_local.device if dist.get_backend() == "nccl" else "cpu"
        v = torch.tensor([part], dtype=torch.int64, device=dev)
        dist.all_reduce(v)          # int64 sums wrap: the sum mod 2^64 does not depend on the split
        part = int(v.item())
    return part & 0x7FFFFFFFFFFFFFFF
