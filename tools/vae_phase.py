"""Where a workgroup of vae_fp6_kernel spends its cycles (decoder convT2 shape: 64 -> 32 channels, 14x14 -> 28x28, B = 1024): shader-clock cycles
per wave and phase of workgroup 0, accumulated by a -DSPK_VT_STAMP=1 build of csrc/vae_fp6.hip (tools/build_variant.sh) and read through its
spk_vt_stamps.  usage: SPKDIFF_LIB=<variant.so> python tools/vae_phase.py [B=1024] [convT2 | convT1 | conv2]
(convT1: 16 -> 64 channels, 7x7 -> 14x14, spike output; conv2: the encoder's 32 -> 64 stride-2 layer, 14x14 -> 7x7)"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import ops, _lib
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
LAYER = sys.argv[2] if len(sys.argv) > 2 else "convT2"
lib = ctypes.CDLL(os.environ["SPKDIFF_LIB"])
g = torch.Generator().manual_seed(1)
coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
if LAYER == "convT2":
    Cin, Cout, H, tr, okind = 64, 32, 14, True, ops.VAE_OUT_COLLAPSED
elif LAYER == "convT1":
    Cin, Cout, H, tr, okind = 16, 64, 7, True, ops.VAE_OUT_S32
else:
    Cin, Cout, H, tr, okind = 32, 64, 14, False, ops.VAE_OUT_PTC
w = ((torch.rand(*((Cin, Cout, 3, 3) if tr else (Cout, Cin, 3, 3)), generator=g) - 0.5) * 0.3).to(dev)
bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.1).to(dev)
a = (torch.rand(Cout, generator=g) * 2 + 0.5).to(dev); b = (torch.rand(Cout, generator=g) - 0.8).to(dev)
Cpad = (Cin + 31) // 32 * 32                                # (S32 records carry 32 channels: the padding channels are zero)
x = (torch.rand(16, B, Cpad, H, H, generator=g) < 0.05).float()
x[:, :, Cin:] = 0
s32 = ops.spikes_to_s32(x.to(dev))
pk = ops.vae_fp6_pack(w, bias, tr)
run = lambda: ops.vae_fp6_fwd(s32, pk, Cout, bn_a=a, bn_b=b, transposed=tr, out_kind=okind, coef=coef)
buf = (ctypes.c_ulonglong * (16 * 8))(); nw = ctypes.c_int(0)
for _ in range(3):
    run()
torch.cuda.synchronize()
assert lib.spk_vt_stamps(buf, ctypes.byref(nw)) == 0
reps = 10
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
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
