"""Time spk_vq_readout_argmin alone at the encode->decode size (B=1024, 7x7, T=16, D=16, K=128): vq_time.py [B]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
z = (torch.rand(B, 7, 7, 16, 16, generator=g) < 0.2).to(torch.uint8).to(dev)
coef = torch.rand(16, generator=g).to(dev); alpha = torch.tensor([0.3], device=dev)
cb = torch.randn(128, 16, generator=g).to(dev)
for zq, xm in ((True, False), (False, False), (True, True)):
    for _ in range(3):
        ops.vq_readout_argmin(z, coef, alpha, cb, want_zq=zq, want_xm=xm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.vq_readout_argmin(z, coef, alpha, cb, want_zq=zq, want_xm=xm)
    e1.record(); torch.cuda.synchronize()
    print(f"want_zq={zq} want_xm={xm}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
