"""Time spk_den_conv3x3_mfma_fp6 at the four denoiser shapes (B=256, 7x7) for every libspkdiff variant given on the
command line (each in a fresh process: the library is chosen at import time through SPKDIFF_LIB)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
    import torch
    from spkdiff import ops
    dev = torch.device("cuda"); B, H, W = 256, 7, 7
    torch.manual_seed(0)
    out = []
    for name, Cout, Cin in (("conv2", 128, 64), ("conv3", 256, 128), ("conv4", 512, 256), ("conv5", 256, 512)):
        w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
        packed = ops.den_pack_weight_fp6(w, torch.zeros(Cout, device=dev))
        x = ops.spikes_to_c4((torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float())
        a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
        for _ in range(3):
            y = ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b)
        evs = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.den_conv3x3_mfma_fp6(x, packed, Cout, bn_a=a, bn_b=b); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sorted(p.elapsed_time(q) for p, q in evs)[10]
        out.append(f"{name} {ms*1e3:7.1f} us (sum {int(y.view(torch.uint8).sum())})")
    # the int8 kernel at conv4 / conv5 on the same box, for reference
    for name, Cout, Cin in (("i8.conv4", 512, 256), ("i8.conv5", 256, 512)):
        w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
        packed = ops.den_pack_weight_i8(w, torch.zeros(Cout, device=dev))
        x = (torch.rand(B, Cin // 32, H, W, 16, 32, device=dev) < 0.05).to(torch.uint8)
        a = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
        for _ in range(3):
            ops.den_conv3x3_mfma(x, packed, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b)
        evs = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.den_conv3x3_mfma(x, packed, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        out.append(f"{name} {sorted(p.elapsed_time(q) for p, q in evs)[10]*1e3:7.1f} us")
    print(" | ".join(out), flush=True)
else:
    libs = sys.argv[1:] or [os.path.join(ROOT, "spiking-diffusion_amd/spkdiff/libspkdiff.so")]
    for lib in libs:
        env = dict(os.environ, SPKDIFF_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):40s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
