"""One CIFAR-shaped (32x32x3) VQ-VAE training iteration loop for a kernel trace: are all convolutions native there too?
usage: python tools/train_vqvae_cifar_prof.py [iters=8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth
from snn_model.vae_model import SNN_VQVAE
from snn_model.vq_diffusion import functional
dev = torch.device("cuda", 0)
cfg = synth.CIFAR
model = SNN_VQVAE(3, cfg.latent_dim, cfg.num_embeddings, 0.06).to(dev)
functional.set_step_mode(net=model, step_mode='m')
model.load_state_dict(synth.cached_state('vqvae', cfg))
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.001)
img = (torch.rand(32, 3, 32, 32, generator=torch.Generator().manual_seed(1)) - 0.5).to(dev)
spike = img.unsqueeze(0).repeat(16, 1, 1, 1, 1)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    a, b, c = model(spike, img)
    opt.zero_grad(); (a + b).backward(); opt.step(); functional.reset_net(model)
torch.cuda.synchronize()
print("loss_eq %.4f loss_rec %.4f mse %.5f" % (float(a), float(b), float(c)))
