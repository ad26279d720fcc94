"""Random-shape stress of the native backward convolutions against fp64: spk_conv3x3_wgrad_bf16 (spikes and spike counts),
spk_conv3x3_dgrad_bf16, spk_conv3x3_dgrad_f16x2.  usage: python tools/backward_stress.py [cases=40] [seed=0]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
CL = torch.channels_last
worst = {"wgrad": 0.0, "wgrad_counts": 0.0, "dgrad_bf16x3": 0.0, "dgrad_f16x2": 0.0}
ratio = dict.fromkeys(worst, 0.0)


def bwd(gy, x, w, which):
    return torch.ops.aten.convolution_backward(gy, x, w, [w.shape[0]], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, which)


for case in range(cases):
    N = rng.choice([1, 3, 8, 9, 31, 64, 100, 257, 512])
    Cout = 16 * rng.randint(1, 32)
    Cin = 32 * rng.randint(1, 16)
    if case % 3 == 0:                                  # (shapes the weight-gradient kernel takes: every third case)
        Cout, Cin = 128 * rng.randint(1, 4), 64 * rng.randint(1, 8)
    HH = rng.choice([7, 8])                            # MNIST-shaped and CIFAR-shaped maps
    g = torch.Generator().manual_seed(case)
    mag = 10.0 ** rng.uniform(-8, 2)
    gy = torch.randn(N, Cout, HH, HH, generator=g) * torch.rand(Cout, generator=g).view(1, -1, 1, 1) * mag
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (10.0 ** rng.uniform(-4, 1))
    s = (torch.rand(N, Cin, HH, HH, generator=g) < rng.uniform(0.01, 0.5)).float()
    cnt = torch.randint(0, 17, (N, Cin, HH, HH), generator=g).float()
    gy_d, w_d = gy.to(dev).contiguous(memory_format=CL), w.to(dev)
    want_gi = bwd(gy.double(), s.double(), w.double(), [True, False, False])[0]
    lib_gi = bwd(gy_d, s.to(dev).contiguous(memory_format=CL), w_d.contiguous(memory_format=CL), [True, False, False])[0].cpu().double()
    e_lib = float((lib_gi - want_gi).norm() / want_gi.norm())
    for form in ("bf16x3", "f16x2"):
        got = ops.conv3x3_dgrad(gy_d, w_d, Cin, form=form).cpu().double()
        e = float((got - want_gi).norm() / want_gi.norm())
        worst["dgrad_" + form] = max(worst["dgrad_" + form], e)
        ratio["dgrad_" + form] = max(ratio["dgrad_" + form], e / max(e_lib, 1e-12))
        assert e <= 2e-6, (form, N, Cout, Cin, e)
    if Cout % 128 == 0 and Cin % 64 == 0:
        for name, inp in (("wgrad", s), ("wgrad_counts", cnt)):
            want_gw = bwd(gy.double(), inp.double(), w.double(), [False, True, False])[1]
            lib_gw = bwd(gy_d, inp.to(dev).contiguous(memory_format=CL), w_d.contiguous(memory_format=CL), [False, True, False])[1].cpu().double()
            got = ops.conv3x3_wgrad(gy_d, inp.to(dev).contiguous(memory_format=CL), Cout, Cin).cpu().double()
            e, el = float((got - want_gw).norm() / want_gw.norm()), float((lib_gw - want_gw).norm() / want_gw.norm())
            worst[name] = max(worst[name], e)
            ratio[name] = max(ratio[name], e / max(el, 1e-12))
            assert e <= 2e-6, (name, N, Cout, Cin, e)
print(f"{cases} random cases (N 1..512, Cout 16..512, Cin 32..512, 7x7 and 8x8 maps, magnitudes 1e-8..1e2): worst relative L2 error against fp64")
for k in worst:
    print(f"  {k:14s} {worst[k]:.2e}   (at most {ratio[k]:.2f}x the framework operator's error on the same case)")
