"""Time the fused LIF scan (spk_lif_fwd) at BASELINE config-3 size for every libspkdiff variant given: GB/s of algorithmic bytes.
usage: python tools/lif_time.py <lib.so> [<lib.so> ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
    import torch
    from spkdiff import ops
    T, N = 16, 1024 * 32 * 28 * 28
    x = torch.randn(T, N, device="cuda") * 1.5
    v = torch.zeros(N, device="cuda")
    for _ in range(3):
        v.zero_(); ops.lif_fwd(x, v)
    ts = []
    for _ in range(20):
        v.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = ops.lif_fwd(x, v); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    by = 8.0 * T * N + 8.0 * N
    ts.sort()
    print(f"median {ts[10]:.4f} ms  {by / (ts[10] * 1e-3) / 1e9:.0f} GB/s  (best {by / (ts[0] * 1e-3) / 1e9:.0f})", flush=True)
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ, SPKDIFF_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
