"""Time spk_convt_fp6_collapsed_fwd alone (decoder convT2 shape: 64 -> 32 channels, 14x14 -> 28x28, B=1024).
usage: python tools/convt_time.py [B=1024] [reps=20]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import ops
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator().manual_seed(1)
w = ((torch.rand(64, 32, 3, 3, generator=g) - 0.5) * 0.3).to(dev)
bias = ((torch.rand(32, generator=g) - 0.5) * 0.1).to(dev)
a = (torch.rand(32, generator=g) * 2 + 0.5).to(dev); b = (torch.rand(32, generator=g) - 0.8).to(dev)
coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
s32 = ops.spikes_to_s32((torch.rand(16, B, 64, 14, 14, generator=g) < 0.05).float().to(dev))
pk = ops.vae_fp6_pack(w, bias, True)
for _ in range(3):
    out = ops.convT_fp6_collapsed(s32, pk, 32, bn_a=a, bn_b=b, coef=coef)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = ops.convT_fp6_collapsed(s32, pk, 32, bn_a=a, bn_b=b, coef=coef)
e1.record(); torch.cuda.synchronize()
print(f"convT fp6 B={B}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (3 launches), checksum {float(out.sum()):.3f}, "
      f"firing {float((out > 0).float().mean()):.3f}")
