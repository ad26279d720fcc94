"""Stress the certified VQ-VAE kernels (spk_vae_fp6_fwd) against the exact int8 gather kernel on many random layers:
vae_fp6_stress.py [seeds=40] [B=8].  Outputs must be bit-equal (collapsed fp32, S32 spikes, u8 spikes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")
coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
total = mism = 0
for seed in range(seeds):
    g = torch.Generator().manual_seed(70000 + seed)
    for layer, hw, Cout in (("dec2", 14, 32), ("dec1", 7, 64), ("enc2", 14, 64), ("dec2", 16, 32), ("dec1", 8, 64), ("enc2", 16, 64)):
        transposed = layer != "enc2"
        Cin = {"dec2": 64, "dec1": 16, "enc2": 32}[layer]
        kind = {"dec2": ops.VAE_OUT_COLLAPSED, "dec1": ops.VAE_OUT_S32, "enc2": ops.VAE_OUT_PTC}[layer]
        geo = dict(k=3, stride=2, pad=1, transposed=transposed, out_pad=1 if transposed else 0)
        wamp = float(10 ** (torch.rand(1, generator=g) * 2.0 - 2.0))
        aamp = float(10 ** (torch.rand(1, generator=g) * 2.3 - 0.7))
        rate = float(10 ** (torch.rand(1, generator=g) * 1.5 - 2.0))
        w = (torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * wamp
        w[:, :, 1, 1] *= 3.0
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.4
        a = ((torch.rand(Cout, generator=g) - 0.3) * aamp).to(dev)
        b = ((torch.rand(Cout, generator=g) - 0.4) * 2.0).to(dev)
        spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < rate).float().to(dev)
        wd, bd = w.to(dev), bias.to(dev)
        ptc = ops.spikes_to_ptc(spikes)
        pk8 = ops.pack_conv_weight_i8(wd, bd, transposed)
        got = ops.vae_fp6_fwd(ops.ptc_to_s32(ptc), ops.vae_fp6_pack(wd, bd, transposed), Cout, bn_a=a, bn_b=b, transposed=transposed,
                              out_kind=kind, coef=coef if layer == "dec2" else None)
        if layer == "dec2":
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, collapse_coef=coef, **geo)
        else:
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)
            if layer == "dec1":
                got, want = ops.s32_to_spikes(got), ops.ptc_to_spikes(want)
        total += want.numel() * (16 if layer == "dec2" else 1)
        mism += int((want != got).sum())
print(f"neuron-steps {total:.3e}  mismatches {mism}")
assert mism == 0
