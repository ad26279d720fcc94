"""A few launches of spk_den_conv3x3_mfma_fp6 at one shape (rocprofv3 counter passes): fp6_one.py Cout Cin [n]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
Cout, Cin = int(sys.argv[1]), int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda"); B, H, W = 256, 7, 7
torch.manual_seed(0)
w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
packed = ops.den_pack_weight_fp6(w, torch.zeros(Cout, device=dev))
x = ops.spikes_to_c4((torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float())
a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
for _ in range(n):
    y = ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b, want_counts=True)
torch.cuda.synchronize()
print("done", int(y[0].view(torch.uint8).sum()))
