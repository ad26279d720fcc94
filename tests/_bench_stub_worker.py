"""Stand-in rank program for tests/test_bench_launcher.py: what bench.py's launcher starts instead of bench.py when
SPKDIFF_BENCH_WORKER points here.  No GPU: gloo process group, the same fence / max-over-ranks reduction bench.py
uses, per-rank Philox keys through the product's own key derivation, and rank 0 prints ONE JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]


def main():
    from spkdiff import dist as sdist
    import bench
    args = bench.parse_args(sys.argv[1:])
    if os.environ.get("STUB_FAIL") == "1":
        raise SystemExit(7)
    rank, local_rank, world = sdist.init("gloo")
    assert world == args.gpus, (world, args.gpus)
    torch.manual_seed(42)                                  # every rank seeded ALIKE: the rank must still separate the streams
    from snn_model.vq_diffusion import AbsorbingDiffusion

    class _Den(torch.nn.Module):
        num_embeddings = 128
    ab = AbsorbingDiffusion(_Den(), mask_id=128)
    # 'global' noise layout (default): ONE key per job -- rank 0's draw, broadcast (ranks seeded differently here on purpose);
    # 'rank' layout: the rank folded into the key, distinct streams for ranks seeded alike
    lo0, hi0 = sdist.shard_range(args.global_batch or args.batch or 256 * world, rank, world)
    ab.set_shard(lo0, hi0 - lo0)                            # a shard of one job: the key broadcast is opt-in through set_shard
    torch.manual_seed(1000 + rank)
    key = torch.tensor([ab._philox_key()], dtype=torch.int64)
    keys = [torch.zeros_like(key) for _ in range(world)]
    dist.all_gather(keys, key)
    ab.noise_layout = 'rank'
    torch.manual_seed(7)
    rkey = torch.tensor([ab._philox_key()], dtype=torch.int64)
    rkeys = [torch.zeros_like(rkey) for _ in range(world)]
    dist.all_gather(rkeys, rkey)
    lo, hi = sdist.shard_range(args.global_batch or args.batch or 256 * world, rank, world)
    imgs = sdist.gather_images(torch.full((hi - lo, 1, 2, 2), rank, dtype=torch.uint8), args.global_batch)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print("noise before the line")
        print(json.dumps({"metric": "stub", "n_gpus": world, "ranks_seen": dist.get_world_size(),
                          "keys_distinct": len({int(k) for k in rkeys}) == world,
                          "job_key_shared": len({int(k) for k in keys}) == 1, "images": int(imgs.shape[0]),
                          "max_t": float(t), "steps": args.steps, "warmup": args.warmup}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
