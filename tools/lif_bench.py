"""Stand-alone launches of the fused LIF scan (spk_lif_fwd) at BASELINE config-3 size, for rocprofv3 counter passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
T, N = 16, 1024 * 32 * 28 * 28
x = torch.randn(T, N, device="cuda") * 1.5
v = torch.zeros(N, device="cuda")
for _ in range(5):
    v.zero_()
    ops.lif_fwd(x, v)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 8 * T * N + 8 * N)
