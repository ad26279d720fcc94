"""In-kernel shader clock of the fp6v2 main kernel (needs a -DSPK_V2_DBG=192 build: stamps + no fixup): after 2 s of
back-to-back launches at the conv4 shape, d(s_memtime) / d(s_memrealtime) * 100 MHz per workgroup, median over workgroups;
also cycles per MFMA per SIMD (a workgroup issues items x chunks x 138 MFMAs per wave)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda"); B, H, W = 256, 7, 7
torch.manual_seed(0)
for name, Cout, Cin in (("conv4", 512, 256), ("conv5", 256, 512)):
    w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
    bias = (torch.rand(Cout, device=dev) - 0.5) * 0.1
    x = ops.spikes_to_s32((torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float())
    a = torch.rand(Cout, device=dev) * 8 + 2; b = torch.rand(Cout, device=dev) * 0.8
    p2 = ops.den_pack_weight_fp6v2(w, bias)
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            ops.den_conv3x3_mfma_fp6v2(x, p2, Cout, bn_a=a, bn_b=b)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.den_conv3x3_mfma_fp6v2(x, p2, Cout, bn_a=a, bn_b=b); e1.record()
    torch.cuda.synchronize()
    ws = list(ops._FLAG_DEFAULT.values())[-1]
    st = ws[2:2 + 4 * 256].view(torch.int64).view(256, 2).double().cpu()
    ghz = (st[:, 0] / st[:, 1]) * 0.1
    mf = (B * (Cout // 32) / 256) * (Cin // 32) * 138
    print(f"{name}: launch {e0.elapsed_time(e1) * 1e3:.1f} us (incl. tail launches); in-kernel clock median {float(ghz.median()):.3f} GHz "
          f"(min {float(ghz.min()):.3f}, max {float(ghz.max()):.3f}); workgroup span median {float(st[:, 1].median()) / 100:.1f} us = "
          f"{float(st[:, 0].median()) / mf:.1f} cycles per MFMA (33 = matrix-pipe bound)", flush=True)
    for v in ops._FLAG_DEFAULT.values():
        v[:2].zero_()
