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
    out = torch.empty((world * bmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send)
    if all(hi - lo == bmax for lo, hi in sizes):
        return out
    parts = [out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def sample_images_sharded(generate_local, total: int):
    """``generate_local(lo, hi)`` -> uint8 [hi-lo, C, H, W] on this rank's device; returns all ``total`` images
    on every rank (rank order == sample order)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    lo, hi = shard_range(total, rank, world)
    return gather_images(generate_local(lo, hi), total)
