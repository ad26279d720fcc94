"""Micro-benchmark of spk_den_conv3x3_mfma: time vs number of K chunks (Cin/32) at fixed Cout, B=256, 7x7, T=16.
A linear fit  t = items_per_CU * (nchunks * t_chunk + t_epilogue)  separates the K-loop cost from the epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops

dev = torch.device("cuda")
ONLY = None
if len(sys.argv) > 1:      # "--only Cout,Cin": a few launches of one shape (for rocprofv3 counter passes)
    ONLY = tuple(int(v) for v in sys.argv[-1].split(","))
B, H, W = 256, 7, 7
torch.manual_seed(0)
res = {}
for Cout in (128, 512):
    for Cin in (32, 64, 128, 256, 512):
        if ONLY and (Cout, Cin) != ONLY:
            continue
        w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
        bias = torch.zeros(Cout, device=dev)
        packed = ops.den_pack_weight_i8(w, bias)
        x = (torch.rand(B, Cin // 32, H, W, 16, 32, device=dev) < 0.06).to(torch.uint8)
        a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
        for mode in (ops.MODE_LIF, ops.MODE_MEAN):
            for _ in range(3):
                ops.den_conv3x3_mfma(x, packed, Cout, mode=mode, bn_a=a, bn_b=b)
            evs = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.den_conv3x3_mfma(x, packed, Cout, mode=mode, bn_a=a, bn_b=b); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(p.elapsed_time(q) for p, q in evs)[5]
            items_per_cu = B * (Cout // 16) / 256
            per_item_us = ms * 1e3 / items_per_cu
            res[(Cout, Cin, mode)] = per_item_us
            print(f"Cout={Cout} Cin={Cin} nchunks={Cin//32} mode={'LIF' if mode==0 else 'MEAN'}: {ms:.3f} ms, per item {per_item_us:.2f} us", flush=True)
for Cout in (() if ONLY else (128, 512)):
    for mode in (0, 3):
        xs = [c // 32 for c in (32, 64, 128, 256, 512)]
        ys = [res[(Cout, c, mode)] for c in (32, 64, 128, 256, 512)]
        n = len(xs); sx = sum(xs); sy = sum(ys); sxx = sum(v * v for v in xs); sxy = sum(p * q for p, q in zip(xs, ys))
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
        print(f"fit Cout={Cout} mode={'LIF' if mode==0 else 'MEAN'}: t_chunk = {slope:.3f} us, t_epilogue+fixed = {icpt:.3f} us")

# ---- the fp6 x fp4 block-scaled kernel (64-channel K chunks, LIF mode only) at the same shapes
if not ONLY:
    res6 = {}
    for Cout in (128, 512):
        for Cin in (64, 128, 256, 512):
            w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
            packed = ops.den_pack_weight_fp6(w, torch.zeros(Cout, device=dev))
            s = (torch.rand(16, B, Cin, H, W, device=dev) < 0.06).float()
            x = ops.spikes_to_c4(s)
            del s
            a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
            for _ in range(3):
                ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b)
            evs = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ms = sorted(p.elapsed_time(q) for p, q in evs)[5]
            per_item_us = ms * 1e3 / (B * (Cout // 16) / 256)
            res6[(Cout, Cin)] = per_item_us
            print(f"fp6 Cout={Cout} Cin={Cin} nchunks={Cin//64}: {ms:.3f} ms, per item {per_item_us:.2f} us", flush=True)
    for Cout in (128, 512):
        xs = [c // 64 for c in (64, 128, 256, 512)]
        ys = [res6[(Cout, c)] for c in (64, 128, 256, 512)]
        n = len(xs); sx = sum(xs); sy = sum(ys); sxx = sum(v * v for v in xs); sxy = sum(p * q for p, q in zip(xs, ys))
        slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
        print(f"fit fp6 Cout={Cout}: t_chunk(64 ch) = {slope:.3f} us, t_epilogue+fixed = {icpt:.3f} us")
