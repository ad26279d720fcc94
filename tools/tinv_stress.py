"""Stress of the staged time-invariant first-layer kernel (csrc/conv_direct.hip, tinv_lif_staged_kernel: per-position work once per position,
table look-up for the sixteen LIF steps) against the step-by-step kernel: a stateless call takes the staged kernel, the same call with a
carried membrane state of zeros takes tinv_lif_kernel and runs the sixteen steps.  Random shapes (3x3 with 1 / 2 / 3 input channels and stride
1 / 2, the 1x1 16-channel spike generator), random weights with BatchNorm terms that put the membrane potentials around the threshold.
usage: python tools/tinv_stress.py [cases=200] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import ops


def one_case(g, dev):
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    if r(0, 3) == 0:
        k, C0, stride, pad = 1, 16, 1, 0
    else:
        k, C0, stride, pad = 3, r(1, 3), r(1, 2), 1
    Cout = (16, 32, 64)[r(0, 2)]
    B, H, W = r(1, 9), r(3, 30), r(3, 30)
    x = (torch.rand(B, C0, H, W, generator=g) - 0.3).to(dev)
    w = ((torch.rand(Cout, C0, k, k, generator=g) - 0.5) * (2.0 / (k * k * C0) ** 0.5)).to(dev)
    bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.2).to(dev)
    a = (torch.rand(Cout, generator=g) * 3 + 0.5).to(dev); b = (torch.rand(Cout, generator=g) * 1.5).to(dev)
    packed = ops.pack_conv_weight(w, False)
    kw = dict(in_kind=ops.IN_TINV, T=16, mode=ops.MODE_LIF, k=k, stride=stride, pad=pad, bn_a=a, bn_b=b, want_ptc=True, chunk_out=Cout)
    got = ops.conv_fused(x, packed, bias, **kw)["ptc"]
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, False, 0), ops.conv_out_size(W, k, stride, pad, False, 0)
    v = torch.zeros(B, Cout, Ho, Wo, device=dev)
    ref = ops.conv_fused(x, packed, bias, v=v, **kw)["ptc"]
    return int((got != ref).sum()), got.numel(), float(got.float().mean())


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda")
    bad = tot = 0; rate = 0.0
    for i in range(n):
        m, e, f = one_case(g, dev)
        bad += m; tot += e; rate += f
    print(f"cases {n}  spikes compared {tot:.3e}  mean firing rate {rate / n:.3f}  mismatches {bad}")
    sys.exit(1 if bad else 0)
