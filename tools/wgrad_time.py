"""Time spk_conv3x3_wgrad_bf16 next to the framework's weight-gradient operator at the denoiser's shapes (B = 32 token maps x
T = 16 = 512 images).  usage: python tools/wgrad_time.py [N=512]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda"); N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
LAYERS = (("conv2", 128, 64), ("conv3", 256, 128), ("conv4", 512, 256), ("conv5", 256, 512), ("conv6", 128, 320))
if os.environ.get("WGRAD_ONLY"):                     # "Cout,Cin": one shape, native kernel only (the counter passes)
    LAYERS = (("only",) + tuple(int(v) for v in os.environ["WGRAD_ONLY"].split(",")),)
for name, Cout, Cin in LAYERS:
    s = (torch.rand(N, Cin, 7, 7, device=dev) < 0.06).float().contiguous(memory_format=torch.channels_last)
    gy = (torch.randn(N, Cout, 7, 7, device=dev) * 1e-3).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    fns = (("native", lambda: ops.conv3x3_wgrad(gy, s, Cout, Cin)),
           ("library", lambda: torch.ops.aten.convolution_backward(gy, s, w, [Cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                                    [False, True, False])))
    out = []
    for label, fn in (fns[:1] if os.environ.get("WGRAD_ONLY") else fns):
        for _ in range(3):
            fn()
        evs = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        out.append(f"{label} {sorted(a.elapsed_time(b) for a, b in evs)[7] * 1e3:7.1f} us")
    fl = 2.0 * Cout * Cin * 9 * N * 49 / 1e9
    print(f"{name} {Cin:3d}->{Cout:3d}  {fl:6.1f} GFLOP  " + " | ".join(out), flush=True)
