import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import synth
from spkdiff.train import GraphedTrainStep
from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
def build():
    den = DummyModel(1, 128, n_steps=16).cuda(0)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=128)
    opt = torch.optim.AdamW(den.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001, capturable=True)
    return den, ab, opt
x0 = torch.randint(0, 128, (B, 1, 7, 7), generator=torch.Generator().manual_seed(42)).float().cuda(0)
den, ab, opt = build()
def eager():
    loss = ab.train_iter(x0)['loss']
    opt.zero_grad(); loss.backward(); opt.step(); functional.reset_net(net=den)
    return loss
for _ in range(5): eager()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): l = eager()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
print(f"eager   {te * 1e3:.2f} ms / iteration, loss {float(l):.4f}")
den2, ab2, opt2 = build()
g = GraphedTrainStep(ab2, opt2, x0)
for _ in range(5): g(x0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): l2 = g(x0)
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
print(f"graphed {tg * 1e3:.2f} ms / iteration, loss {float(l2):.4f}  ({B / tg:.0f} token maps/s vs {B / te:.0f})")
