"""Random shapes for the native training convolutions (csrc/conv_train.hip): layer kind, channel counts, kernel size, stride, padding, output
padding, map size, batch, weight layout, dense / binary input -- forward, data gradient, weight and bias gradient against torch's fp64
operators on the CPU.  Shapes the kernels do not take are counted and skipped.  usage: python tools/conv_train_stress.py [cases=300] [seed=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
import torch.nn.functional as F
from spkdiff import ops
dev = torch.device("cuda")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


done = skipped = 0
worst = {"y": 0.0, "gi": 0.0, "gw": 0.0, "gb": 0.0}
kinds = {}
while done < ncase:
    tr = ri(0, 1) == 1
    k = ri(1, 4)
    st = ri(1, 2) if tr else ri(1, 3)
    pd = ri(0, min(2, k - 1)) if k > 1 else 0
    op = ri(0, st - 1) if tr else 0
    cin = [1, 2, 3, 4, 8, 16, 24, 32, 40, 48, 64][ri(0, 10)]
    cout = [1, 2, 3, 4, 8, 16, 20, 32, 33, 48, 64][ri(0, 10)]
    H, W, N = ri(max(k, 2), 17), ri(max(k, 2), 17), ri(1, 9)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(*wshape, generator=g) * 0.2
    Ho = (H - 1) * st - 2 * pd + k + op if tr else (H + 2 * pd - k) // st + 1
    Wo = (W - 1) * st - 2 * pd + k + op if tr else (W + 2 * pd - k) // st + 1
    if Ho < 1 or Wo < 1:
        continue
    need_gi = ri(0, 3) > 0
    wd = w.to(dev)
    if ri(0, 1):
        wd = wd.contiguous(memory_format=torch.channels_last)
    if not ops.conv_train_supported((N, cin, H, W), wd, st, pd, tr, op, need_gi, forward=True):
        skipped += 1
        continue
    x = (torch.rand(N, cin, H, W, generator=g) < 0.2).float() if ri(0, 1) else torch.randn(N, cin, H, W, generator=g)
    b = torch.randn(cout, generator=g) * 0.1 if ri(0, 3) else None
    xo, wo = x.double().requires_grad_(need_gi), w.double().requires_grad_(True)
    bo = None if b is None else b.double().requires_grad_(True)
    yo = F.conv_transpose2d(xo, wo, bo, st, pd, op) if tr else F.conv2d(xo, wo, bo, st, pd)
    gy = torch.randn(yo.shape, generator=g)
    (yo * gy.double()).sum().backward()
    xd = x.to(dev)
    if ri(0, 1):
        xd = xd.contiguous(memory_format=torch.channels_last)
    xd = xd.requires_grad_(need_gi)
    wd = wd.requires_grad_(True)
    bd = None if b is None else b.to(dev).requires_grad_(True)
    tag = ("convT" if tr else "conv") + f" {cin}->{cout} k{k} s{st} p{pd} op{op} {H}x{W} N{N}"
    try:
        y = ops.NativeConvTrainFunction.apply(xd, wd, bd, st, pd, tr, op)
        (y * gy.to(dev)).sum().backward()
    except Exception:
        print("FAILED CASE:", tag, flush=True)
        raise
    errs = {"y": rel(y.detach().cpu(), yo.detach()), "gw": rel(wd.grad.cpu(), wo.grad)}
    if need_gi:
        errs["gi"] = rel(xd.grad.cpu(), xo.grad)
    if b is not None:
        errs["gb"] = rel(bd.grad.cpu(), bo.grad)
    for kk, v in errs.items():
        worst[kk] = max(worst[kk], v)
        assert v <= (3e-5 if kk == "gb" else 5e-6), (tag, kk, v)    # (gb: an fp32 sum of random signs against its own small total)
    kinds[("T" if tr else "C") + ("1in" if cin == 1 else "1out" if cout == 1 else "mm")] = kinds.get(("T" if tr else "C") + ("1in" if cin == 1 else "1out" if cout == 1 else "mm"), 0) + 1
    done += 1
print(f"conv_train_stress: {done} random cases within 5e-6 of fp64 (bias gradient 3e-5) (worst relative L2: {worst}); {skipped} drawn shapes not taken by the native "
      f"kernels (framework's operator); by kind {kinds}")
