"""Phase picture of the duo form of spk_den_conv3x3_mfma_fp6v2 (two workgroups per CU): needs a library built with
-DSPK_V2_DUO_DBG=1 (tools/build_variant.sh den_mfma_fp6v2.hip "-DSPK_V2_DUO_DBG=1" ../spkdiff/variants/duo_dbg.so; SPKDIFF_LIB).
Every workgroup stamps s_memrealtime (10 ns ticks) at the start of each item's K loop and of its scan, plus its CU key and its
arrival parity.  Prints, for one launch of the conv4 shape (B = 256): per-item K-loop and scan durations, and for every CU the
fraction of one workgroup's scan time that lies inside its partner's K loops (1.0 = the scans run beside MFMAs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import numpy as np
import torch
from spkdiff import ops

dev = torch.device("cuda")
B, H, W = 256, 7, 7
Cout, Cin = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (512, 256)))
torch.manual_seed(0)
w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
bias = (torch.rand(Cout, device=dev) - 0.5) * 0.1
x = (torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float()
a = torch.rand(Cout, device=dev) * 8 + 2; b = torch.rand(Cout, device=dev) * 0.8
p2 = ops.den_pack_weight_fp6v2(w, bias)
x2 = ops.spikes_to_s32(x)
for _ in range(3):
    ops.den_conv3x3_mfma_fp6v2(x2, p2, Cout, bn_a=a, bn_b=b)
torch.cuda.synchronize()
buf = next(iter(ops._FLAG_DEFAULT.values()))
cap = 1 << 20
raw = buf[2 + cap // 2: 2 + cap // 2 + 512 * 64 * 2].cpu().numpy().view(np.uint64).reshape(512, 64).astype(np.int64)
nits = ((raw[:, :62] != 0).sum(1) // 2).astype(int)
t0 = raw[:, 0][raw[:, 0] > 0].min()
ks, sc, ends = [], [], []
for w in range(512):
    n = nits[w]
    if n == 0:
        continue
    ks += list(raw[w, 1:2 * n:2] - raw[w, 0:2 * n:2])
    sc += list(raw[w, 2:2 * n:2] - raw[w, 1:2 * n - 1:2])
    ends.append(raw[w, 2 * n - 1])
ks, sc = np.array(ks), np.array(sc)
print(f"items per workgroup {nits.min()}..{nits.max()} (mean {nits.mean():.1f}); K loop {ks.mean() / 100:.2f} us (min {ks.min() / 100:.2f}, max {ks.max() / 100:.2f}); "
      f"scan {sc.mean() / 100:.2f} us (min {sc.min() / 100:.2f}, max {sc.max() / 100:.2f}); last scan starts {(np.min(ends) - t0) / 100:.1f} .. {(np.max(ends) - t0) / 100:.1f} us")
keys, late = raw[:, 62], raw[:, 63]
fr = []
for k in np.unique(keys):
    wg = np.nonzero(keys == k)[0]
    if len(wg) != 2:
        continue
    for me, other in ((wg[0], wg[1]), (wg[1], wg[0])):
        tot = ins = 0
        for i in range(nits[me] - 1):
            s0, s1 = raw[me, 2 * i + 1], raw[me, 2 * i + 2]
            tot += s1 - s0
            for j in range(nits[other]):
                k0, k1 = raw[other, 2 * j], raw[other, 2 * j + 1]
                ins += max(0, min(s1, k1) - max(s0, k0))
        fr.append(ins / max(tot, 1))
print(f"CUs with exactly two workgroups: {len(fr) // 2} of {len(np.unique(keys))}; arrival parities 0/1: {(late == 0).sum()}/{(late == 1).sum()}; "
      f"scan time inside the partner's K loops: mean {np.mean(fr):.2f}, 10th pct {np.percentile(fr, 10):.2f}, 90th {np.percentile(fr, 90):.2f}")
wg = np.nonzero(keys == keys[0])[0][:2]
for g in wg:
    print(f"wg {g} (parity {late[g]}):", " ".join(f"{(v - t0) / 100:.1f}" for v in raw[g, :2 * nits[g]]))
