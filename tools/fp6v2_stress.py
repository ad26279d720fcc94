"""Stress the certified four-digit kernel against the exact six-plane kernel on many random layers: fp6v2_stress.py [seeds=20] [B=32].
Every spike and spike count must be equal (the certification bound carries no spare factor since round 2: a bound that were too
tight would show up here as a rare mismatch).  Regimes: weight amplitude, BatchNorm scale (sign and size), bias, firing rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda")
total = mism = cmism = 0
for seed in range(seeds):
    g = torch.Generator().manual_seed(90000 + seed)
    for hw in (7, 8):
        Cout, Cin = ((128, 64), (256, 128), (512, 256), (256, 512))[int(torch.randint(0, 4, (1,), generator=g))]
        wamp = float(10 ** (torch.rand(1, generator=g) * 2.0 - 2.0))            # 0.01 .. 1
        aamp = float(10 ** (torch.rand(1, generator=g) * 2.3 - 0.7))            # 0.2 .. 40
        rate = float(10 ** (torch.rand(1, generator=g) * 1.5 - 2.0))            # 0.01 .. 0.3
        w = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * wamp
        w[:, :, 1, 1] *= 3.0
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.4
        a = (torch.rand(Cout, generator=g) - 0.3) * aamp
        b = (torch.rand(Cout, generator=g) - 0.4) * 2.0
        # pull the pre-activations of a part of the channels towards the threshold: b ~ 1 - E[z] would need the data; spread b
        spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < rate).float()
        wd, biasd, ad, bd, sd = w.to(dev), bias.to(dev), a.to(dev), b.to(dev), spikes.to(dev)
        o2, c2 = ops.den_conv3x3_mfma_fp6v2(ops.spikes_to_s32(sd), ops.den_pack_weight_fp6v2(wd, biasd), Cout, bn_a=ad, bn_b=bd,
                                            want_counts=True)
        o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(wd, biasd), Cout, bn_a=ad, bn_b=bd,
                                          want_counts=True)
        s2, s1 = ops.s32_to_spikes(o2), ops.c4_to_spikes(o1)
        total += s1.numel()
        mism += int((s1 != s2).sum())
        cmism += int((c1 != c2).sum())
print(f"neuron-steps {total:.3e}  spike mismatches {mism}  count mismatches {cmism}")
assert mism == 0 and cmism == 0
