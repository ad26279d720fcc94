"""Where does a bench step spend its time when several ranks share ONE device over gloo (testing aid)?  Per rank: ms of sample(),
decode_tokens() and the image gather, each fenced by a device synchronise.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dist_phase_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
import torch.distributed as dist
sys.argv = ["bench.py"]
import bench
from spkdiff import dist as sdist
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
model, den, ab = bench.build_models(dev, 16)
B = 256
ab.set_shard(rank * B, B)
ab.sync_key = False
ab.skip_untouched = False
torch.manual_seed(42)
for _ in range(2):
    ab.sample(temp=1.0, sample_steps=100)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
for i in range(3):
    t0 = time.perf_counter()
    tok = ab.sample(temp=1.0, sample_steps=100)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _, u8 = model.decode_tokens(tok.reshape(B, 7, 7), T=16)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out = sdist.gather_images(u8, B * world if world > 1 else None)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rank {rank} step {i}: sample {1e3 * (t1 - t0):8.1f} ms  decode {1e3 * (t2 - t1):7.2f} ms  gather {1e3 * (t3 - t2):8.2f} ms  images {tuple(out.shape)}", flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
