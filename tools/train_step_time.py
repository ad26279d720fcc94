"""Time the diffusion training step (train_iter + backward + AdamW step + reset_net, R/main.py:243-252) on one GPU.

usage: python tools/train_step_time.py [B] [steps] [--modular]
--modular runs the blocks module by module (library convolution and BatchNorm + the LIF-only HIP pair) instead of the fused
graph; --library-forward keeps the fused block tails but uses the library's forward convolutions."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth, fused
from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 32
steps = int(args[1]) if len(args) > 1 else 10
if "--modular" in sys.argv:
    def _module_by_module(self, x, binary_input=False):
        for m in self:
            x = m(x)
        return x
    fused.FusedSequential.train_forward = _module_by_module
if "--library-forward" in sys.argv:
    fused.FusedSequential.exact_train_forward = False
dev = torch.device("cuda")
den = DummyModel(1, 128).cuda(0)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
den.train()
ab = AbsorbingDiffusion(den, mask_id=128)
opt = torch.optim.AdamW(den.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
g = torch.Generator().manual_seed(1)
x0 = torch.randint(0, 128, (B, 1, 7, 7), generator=g).float().to(dev)


def step():
    loss = ab.train_iter(x0)['loss']
    opt.zero_grad(); loss.backward(); opt.step(); functional.reset_net(net=den)
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"B={B} {'modular' if '--modular' in sys.argv else 'fused'}: {dt*1e3:.2f} ms/step, {B/dt:.1f} img/s, loss {float(l.detach()):.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**20:.0f} MiB")
