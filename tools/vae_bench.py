"""Encode->decode throughput of the spiking VQ-VAE (BASELINE config 3: FMNIST-shaped, B=1024, T=16), per kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth, ops
from snn_model.vae_model import SNN_VQVAE, functional

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
model = SNN_VQVAE(1, 16, 128, torch.tensor(1.0))
functional.set_step_mode(net=model, step_mode='m')
model.load_state_dict(synth.synth_vqvae_state(synth.MNIST))
model = model.cuda().eval()
img = (torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(42)) - 0.5).to(dev)

def run():
    idx = model.encode_images(img, 16)
    pred, u8 = model.decode_tokens(idx, 16)
    return idx, u8

for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B}: encode+decode {dt*1e3:.2f} ms -> {B/dt:.0f} images/s")
# per stage
def timeit(f):
    f(); torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
print("encode only: %.2f ms" % timeit(lambda: model.encode_images(img, 16)))
idx = model.encode_images(img, 16)
print("decode only: %.2f ms" % timeit(lambda: model.decode_tokens(idx, 16)))
