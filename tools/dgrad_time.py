"""Time spk_conv3x3_dgrad_bf16 next to the framework's data-gradient operator (aten.convolution_backward, input gradient only) at the denoiser's shapes
(B = 32 token maps x T = 16 = 512 images), channels-last, and for conv6 (320 input channels) split into channel slices.
usage: python tools/dgrad_time.py [N=512]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda"); N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
CL = torch.channels_last


def timed(fn):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[7] * 1e3


def dgrad(gy, x, w):
    return torch.ops.aten.convolution_backward(gy, x, w, [w.shape[0]], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]


LAYERS = (("conv2", 128, 64), ("conv3", 256, 128), ("conv4", 512, 256), ("conv5", 256, 512), ("conv6", 128, 320))
if os.environ.get("DGRAD_ONLY"):                     # "Cout,Cin": one shape, native kernel only (the counter passes)
    LAYERS = (("only",) + tuple(int(v) for v in os.environ["DGRAD_ONLY"].split(",")),)
for name, Cout, Cin in LAYERS:
    x = torch.zeros(N, Cin, 7, 7, device=dev).contiguous(memory_format=CL)
    gy = (torch.randn(N, Cout, 7, 7, device=dev) * 1e-3).contiguous(memory_format=CL)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
    t = 1.0 if os.environ.get("DGRAD_ONLY") else timed(lambda: dgrad(gy, x, w))
    fl = 2.0 * Cout * Cin * 9 * N * 49 / 1e9
    tn = timed(lambda: ops.conv3x3_dgrad(gy, w, Cin, form="bf16x3"))
    tf = timed(lambda: ops.conv3x3_dgrad(gy, w, Cin, form="f16x2"))
    line = f"{name} {Cin:3d}->{Cout:3d}  {fl:6.1f} GFLOP  bf16x3 {tn:7.1f} us ({fl / tn * 1e3:6.1f} TFLOP/s) | f16x2 {tf:7.1f} us ({fl / tf * 1e3:6.1f} TFLOP/s) | library {t:7.1f} us ({fl / t * 1e3:6.1f} TFLOP/s)"
    if Cin == 320 and len(sys.argv) > 2:
        for parts in ((256, 64), (128, 128, 64), (64,) * 5):
            ws, xs, o = [], [], 0
            for c in parts:
                ws.append(w[:, o:o + c].contiguous(memory_format=CL)); xs.append(x[:, o:o + c].contiguous(memory_format=CL)); o += c
            tp = timed(lambda: [dgrad(gy, xx, ww) for xx, ww in zip(xs, ws)])
            ref = dgrad(gy, x, w)
            got = torch.cat([dgrad(gy, xx, ww) for xx, ww in zip(xs, ws)], 1)
            line += f" | slices {parts}: {tp:7.1f} us (max diff {float((ref - got).abs().max()):.1e})"
    print(line, flush=True)
