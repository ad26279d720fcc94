"""Time the native training convolutions (csrc/conv_train.hip) of the VQ-VAE's six layers -- forward, data gradient, weight
gradient -- next to the framework's operators at the training shapes (batch 32 x T 16 = 512 images).
usage: python tools/conv_train_time.py [N=512] [nolib]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
import torch.nn.functional as F
from spkdiff import ops
dev = torch.device("cuda"); N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
nolib = "nolib" in sys.argv
torch.manual_seed(0)
# name, Cin, Cout, k, stride, pad, transposed, out_pad, H
LAYERS = (("enc.conv1", 1, 32, 3, 2, 1, False, 0, 28), ("enc.conv2", 32, 64, 3, 2, 1, False, 0, 14), ("enc.conv3", 64, 16, 1, 1, 0, False, 0, 7),
          ("dec.convT1", 16, 64, 3, 2, 1, True, 1, 7), ("dec.convT2", 64, 32, 3, 2, 1, True, 1, 14), ("dec.convT3", 32, 1, 3, 1, 1, True, 0, 28))
only = os.environ.get("CT_ONLY")


def timed(fn):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[7] * 1e3


tot_n = tot_l = 0.0
for name, cin, cout, k, st, pd, tr, op, H in LAYERS:
    if only and only != name:
        continue
    x = (torch.rand(N, cin, H, H, device=dev) < 0.1).float().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(cout, device=dev)
    y = ops.conv_train_forward(x, w, b, st, pd, tr, op)
    gy = torch.randn_like(y)
    need_gi = cin > 1
    macs = y.numel() / cout * cout * cin * k * k / (st * st if tr else 1)
    row = [f"{name:11s} {cin:2d}->{cout:2d} {2 * macs / 1e9:5.2f} GF"]
    fwd_l = (lambda: F.conv_transpose2d(x, w, b, st, pd, op)) if tr else (lambda: F.conv2d(x, w, b, st, pd))
    bwd_l = lambda gi, gw: torch.ops.aten.convolution_backward(gy, x, w, [cout], [st, st], [pd, pd], [1, 1], tr, [op, op], 1, [gi, gw, gw])
    for label, nat, lib_ in (("fwd", lambda: ops.conv_train_forward(x, w, b, st, pd, tr, op), fwd_l),
                             ("dgrad", (lambda: ops.conv_train_backward(gy, x, w, st, pd, tr, op, (True, False, False))) if need_gi else None,
                              lambda: bwd_l(True, False)),
                             ("wgrad", lambda: ops.conv_train_backward(gy, x, w, st, pd, tr, op, (False, True, True)), lambda: bwd_l(False, True))):
        if nat is None:
            continue
        tn = timed(nat); tot_n += tn
        if nolib:
            row.append(f"{label} {tn:6.1f}")
        else:
            tl = timed(lib_); tot_l += tl
            row.append(f"{label} {tn:6.1f} / {tl:6.1f}")
    print(" | ".join(row), flush=True)
print(f"total native {tot_n:.0f} us" + ("" if nolib else f", framework {tot_l:.0f} us (HIP events around each call: includes its bias / layout launches)"))
