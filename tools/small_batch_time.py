"""The reverse process at small batches (R/main.py's own call is B = 16, 49 steps): ms per sample() call as one hipGraph replay in the
dense (fused step tail), elimination and elimination + position-list forms, for every libspkdiff given on the command line
(SPKDIFF_LIB, fresh process each).  usage: small_batch_time.py [lib.so ...]   (SB_BATCHES=16,32,64 SB_STEPS=49)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
    import torch
    sys.argv = ["bench.py"]
    import bench
    dev = torch.device("cuda", 0)
    steps = int(os.environ.get("SB_STEPS", "49"))
    out = []
    for B in [int(x) for x in os.environ.get("SB_BATCHES", "16,32,64").split(",")]:
        model, den, ab = bench.build_models(dev, 16)
        ab.n_samples = B
        res = {}
        for name, (sk, li) in {"dense": (False, False), "elim": (True, False), "lists": (True, True)}.items():
            ab.skip_untouched, ab.list_positions = sk, li
            torch.manual_seed(1)
            for _ in range(3):
                ab.sample(temp=1.0, sample_steps=steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                ab.sample(temp=1.0, sample_steps=steps)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / n * 1e3
        out.append(f"B={B}: " + " ".join(f"{k} {v:6.2f}" for k, v in res.items()) + " ms")
        ab._graphs.clear()
        del model, den, ab
    print(" | ".join(out), flush=True)
else:
    libs = sys.argv[1:] or [os.path.join(ROOT, "spiking-diffusion_amd/spkdiff/libspkdiff.so")]
    for rep in range(2):
        for lib in libs:
            env = dict(os.environ, SPKDIFF_LIB=os.path.abspath(lib))
            r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
            print(f"{os.path.basename(lib):28s} {r.stdout.strip() or r.stderr.strip()[-600:]}", flush=True)
