"""Read the in-kernel cycle counters of a -DSPK_FP6_DBG=64 build (DMA wait / barrier / K loop per chunk, per wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda"); B, H, W = 256, 7, 7
for Cout, Cin in ((512, 256), (256, 512)):
    w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
    packed = ops.den_pack_weight_fp6(w, torch.zeros(Cout, device=dev))
    x = ops.spikes_to_c4((torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float())
    a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
    for _ in range(3):
        y = ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b)
    torch.cuda.synchronize()
    t = y.view(torch.uint8).reshape(-1)[:16 * 32].view(torch.int64).reshape(16, 4).cpu()
    for i in range(16):
        d = t[i].tolist(); n = max(d[3], 1)
        print(f"Cout={Cout} Cin={Cin} block {i//4} wave {i%4}: chunks {d[3]}  dma-wait {d[0]/n:7.1f}  barrier {d[1]/n:7.1f}  k-loop {d[2]/n:7.1f} cycles/chunk")
