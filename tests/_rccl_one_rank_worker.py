"""Child process of tests/test_gpu_rccl.py: a fresh interpreter opens a ONE-rank ``nccl`` (= RCCL on ROCm) process group
before anything else in the process has one, runs the path's only collective (``spkdiff.dist``) on real
sampler output on the HIP device, destroys the group and prints one JSON line.  Started with ``subprocess`` (never a re-exec
of a process that has touched the GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "spiking-diffusion_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    from spkdiff import dist as sdist
    from spkdiff import synth
    from snn_model.vae_model import SNN_VQVAE, functional
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion

    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(),
           "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())}

    cfg = synth.MNIST
    model = SNN_VQVAE(1, 16, 128, torch.tensor(1.0))
    functional.set_step_mode(net=model, step_mode='m')
    model.load_state_dict(synth.synth_vqvae_state(cfg))
    model = model.to(dev).eval()
    den = DummyModel(1, 128).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(cfg))
    den.eval()
    ab = AbsorbingDiffusion(den, mask_id=128)
    B, steps = 8, 5
    toks = {}

    def gen(lo, hi):
        assert ab.global_first == lo and ab.n_samples == hi - lo          # sample_images_sharded(sampler=ab) declared the shard
        torch.manual_seed(7)
        tok = ab.sample(temp=1.0, sample_steps=steps)
        toks["t"] = tok
        _, u8 = model.decode_tokens(tok.reshape(hi - lo, 7, 7))
        toks["u8"] = u8
        return u8

    # (1) the sharded sampler: ONE all_gather_into_tensor of uint8 images through RCCL on the device
    imgs = sdist.sample_images_sharded(gen, B, sampler=ab)
    torch.cuda.synchronize()
    out["gather_equal"] = bool(imgs.is_cuda and imgs.dtype == torch.uint8 and tuple(imgs.shape) == (B, 1, 28, 28)
                               and torch.equal(imgs, toks["u8"]))
    # (2) the padded branch: this rank holds fewer images than the largest shard of `total` -> zero-padded to it, one collective
    part = toks["u8"][:5].contiguous()
    padded = sdist.gather_images(part, total=B)
    torch.cuda.synchronize()
    out["gather_padded"] = bool(tuple(padded.shape) == (B, 1, 28, 28) and torch.equal(padded[:5], part)
                                and int(padded[5:].to(torch.int64).abs().sum()) == 0)
    # (3) the checksum's int64 all-reduce on the device == the local checksum at one rank
    cs = sdist.global_token_checksum(toks["t"], 0)
    out["checksum_allreduce"] = bool(cs == (sdist.token_checksum(toks["t"], 0) & 0x7FFFFFFFFFFFFFFF))
    # (4) the key broadcast (device int64 broadcast from rank 0) as _philox_key issues it at world_size > 1
    k = torch.tensor([123456789012345], dtype=torch.int64, device=dev)
    dist.broadcast(k, 0)
    out["key_broadcast"] = bool(int(k.item()) == 123456789012345)
    # (5) the same tokens without any process group semantics involved (a second sampler, no shard): the collective changed nothing
    ab2 = AbsorbingDiffusion(den, mask_id=128)
    ab2.n_samples = B
    torch.manual_seed(7)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t2 = ab2.sample(temp=1.0, sample_steps=steps)
    out["tokens_equal_unsharded"] = bool(torch.equal(t2, toks["t"]))
    dist.barrier()
    dist.destroy_process_group()
    out["destroyed"] = not dist.is_initialized()
    print("RCCL_ONE_RANK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
