"""Time the second-generation fp6 denoiser convolution (spk_den_conv3x3_mfma_fp6v2) next to the first-generation kernel at
the four denoiser shapes (B=256, 7x7), for every libspkdiff variant given on the command line (SPKDIFF_LIB, fresh process
each).  Inputs fire at a few percent with BatchNorm terms that put the membrane potentials around the threshold; prints the
spike mismatches between the two kernels and -- for a build with -DSPK_V2_DBG=64 -- the number of flagged neurons."""
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
        bias = (torch.rand(Cout, device=dev) - 0.5) * 0.1
        x = (torch.rand(16, B, Cin, H, W, device=dev) < 0.05).float()
        a = torch.rand(Cout, device=dev) * 8 + 2; b = torch.rand(Cout, device=dev) * 0.8
        p1, p2 = ops.den_pack_weight_fp6(w, bias), ops.den_pack_weight_fp6v2(w, bias)
        x1, x2 = ops.spikes_to_c4(x), ops.spikes_to_s32(x)
        for _ in range(3):
            y1 = ops.den_conv3x3_mfma_fp6(x1, p1, Cout, bn_a=a, bn_b=b)
            y2 = ops.den_conv3x3_mfma_fp6v2(x2, p2, Cout, bn_a=a, bn_b=b)
        torch.cuda.synchronize()
        flagged = sum(int(v[0]) for v in ops._FLAG_DEFAULT.values())      # non-zero only for a -DSPK_V2_DBG=64 build
        for v in ops._FLAG_DEFAULT.values():
            v[:2].zero_()
        s1, s2 = ops.c4_to_spikes(y1), ops.s32_to_spikes(y2)
        mism = int((s1 != s2).sum())
        ts = []
        for fn in (lambda: ops.den_conv3x3_mfma_fp6(x1, p1, Cout, bn_a=a, bn_b=b),
                   lambda: ops.den_conv3x3_mfma_fp6v2(x2, p2, Cout, bn_a=a, bn_b=b)):
            evs = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); evs.append((e0, e1))
            torch.cuda.synchronize()
            ts.append(sorted(p.elapsed_time(q) for p, q in evs)[10] * 1e3)
        out.append(f"{name} v1 {ts[0]:6.1f} v2 {ts[1]:6.1f} us rate {float(s1.mean()):.3f} mism {mism} flagged {flagged // 3}")
    print(" | ".join(out), flush=True)
else:
    libs = sys.argv[1:] or [os.path.join(ROOT, "spiking-diffusion_amd/spkdiff/libspkdiff.so")]
    for lib in libs:
        env = dict(os.environ, SPKDIFF_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):32s} {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
