"""Where a workgroup of vae_fp6_kernel spends its cycles (decoder convT2 shape: 64 -> 32 channels, 14x14 -> 28x28, B = 1024): shader-clock cycles
per wave and phase of workgroup 0, accumulated by a -DSPK_VT_STAMP=1 build of csrc/vae_fp6.hip (tools/build_variant.sh) and read through its
spk_vt_stamps.  usage: SPKDIFF_LIB=<variant.so> python tools/vae_phase.py [B=1024]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import ops, _lib
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = ctypes.CDLL(os.environ["SPKDIFF_LIB"])
g = torch.Generator().manual_seed(1)
w = ((torch.rand(64, 32, 3, 3, generator=g) - 0.5) * 0.3).to(dev)
bias = ((torch.rand(32, generator=g) - 0.5) * 0.1).to(dev)
a = (torch.rand(32, generator=g) * 2 + 0.5).to(dev); b = (torch.rand(32, generator=g) - 0.8).to(dev)
coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
s32 = ops.spikes_to_s32((torch.rand(16, B, 64, 14, 14, generator=g) < 0.05).float().to(dev))
pk = ops.vae_fp6_pack(w, bias, True)
buf = (ctypes.c_ulonglong * (16 * 8))(); nw = ctypes.c_int(0)
for _ in range(3):
    ops.convT_fp6_collapsed(s32, pk, 32, bn_a=a, bn_b=b, coef=coef)
torch.cuda.synchronize()
assert lib.spk_vt_stamps(buf, ctypes.byref(nw)) == 0
reps = 10
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.convT_fp6_collapsed(s32, pk, 32, bn_a=a, bn_b=b, coef=coef)
e1.record(); torch.cuda.synchronize()
assert lib.spk_vt_stamps(buf, ctypes.byref(nw)) == 0
names = ["copy+wait", "popcounts", "multiply", "scan+store", "end barrier", "held reads", "(passes)", "total"]
print(f"{e0.elapsed_time(e1) / reps * 1e3:.1f} us per call; cycles per launch of workgroup 0, by wave:")
print("wave " + " ".join(f"{n:>11s}" for n in names))
tot = [0.0] * 8
for wv in range(nw.value):
    row = [buf[wv * 8 + k] / reps for k in range(8)]
    tot = [t + r for t, r in zip(tot, row)]
    print(f"{wv:4d} " + " ".join(f"{r:11.0f}" for r in row))
print("mean " + " ".join(f"{t / nw.value:11.0f}" for t in tot))
tt = tot[7]
print("frac " + " ".join(f"{t / tt:11.3f}" for t in tot))
