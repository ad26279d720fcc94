"""The two training iterations that are NOT bench.py's --workload train line, eager, for a kernel trace:
  vqvae      R/main.py:100-146  SNN_VQVAE step (loss_eq + loss_rec, AdamW, reset_net), MNIST shape, batch 32
  diff8x8    R/main.py:226-252  diffusion step on CIFAR-shaped 8x8 token maps, batch 32
usage: python tools/train_other_prof.py <vqvae|diff8x8> [iters=12]   (a marker kernel -- spk_clock_probe is not used; iterations
are cut at the optimizer's first launch by the summarising script)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth
from snn_model.vae_model import SNN_VQVAE
from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = 32
if what == "vqvae":
    cfg = synth.MNIST
    model = SNN_VQVAE(1, cfg.latent_dim, cfg.num_embeddings, 0.08).to(dev)
    functional.set_step_mode(net=model, step_mode='m')
    model.load_state_dict(synth.cached_state('vqvae', cfg))
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    images = (synth.stroke_images(B, 5) - 0.5).to(dev)
    spike = images.unsqueeze(0).repeat(16, 1, 1, 1, 1)

    def step():
        loss_eq, loss_rec, real = model(spike, images)
        opt.zero_grad(); (loss_eq + loss_rec).backward(); opt.step(); functional.reset_net(model)
else:
    cfg = synth.CIFAR
    den = DummyModel(1, cfg.num_embeddings).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.cached_state('denoiser', cfg))
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=cfg.num_embeddings)
    ab.shape = (8, 8)
    opt = torch.optim.AdamW(den.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    x0 = torch.randint(0, cfg.num_embeddings, (B, 1, 8, 8), generator=torch.Generator().manual_seed(1)).float().to(dev)

    def step():
        loss = ab.train_iter(x0)['loss']
        opt.zero_grad(); loss.backward(); opt.step(); functional.reset_net(net=den)
import time
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print(f"{what}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per iteration (eager, host-side clock)")
