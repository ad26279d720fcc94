"""Generates tests/golden/f15_bench_job_tokens.npz: the BENCH LINE'S OWN JOB against the CPU oracle.

TEST INFRASTRUCTURE.  Runs on a GPU box (``gpurun -- python oracle/gen_f15_bench_job.py``): the job's noise is the on-device
Philox stream (``spk_philox_noise`` dumps exactly what ``spk_psample_step`` / ``spk_den_step_tail`` consume), the oracle runs
on that box's host cores (minutes per batch).  The job is the one the driver times with the default ``python bench.py``:
seed 42, ``--steps 20 --warmup 5``, MNIST, B = 256, 100 reverse steps, T = 16, synthetic weights, dense.  bench.py seeds torch's
CPU generator with 42 and every sample() call takes ONE 62-bit draw as its key: two set-up passes, 5 warm-up batches, 20 timed
batches -- so the FIRST timed batch uses draw #8 and the LAST draw #27.  For each of the two batches the file holds

  key, tokens of the fp32 CPU oracle on the dumped noise (R/snn_model/vq_diffusion.py:103-142 as oracle/snn_ref.py restates it),
  a per-image token checksum, and the FRAGILE images: images whose fp32-oracle trajectory meets a neuron-step whose membrane
  potential is so close to the threshold that the oracle's own convolution rounding (oneDNN, fp32 accumulation in an order nobody
  controls) decides the spike differently from the exact dot product.  For those the file also holds the tokens of the oracle
  with every convolution evaluated exactly (fp64 sums, one rounding: the arithmetic contract of the HIP kernels), the reverse
  step where the two part, and the margin |h - 1| of the flipped spike in the exact evaluation.

The test (tests/test_gpu_parity.py::test_f15_bench_job_tokens_vs_fixture) and bench.py's ``oracle_fixture`` object require the
HIP tokens to equal the fp32 oracle's outside the fragile images and the exact-convolution oracle's inside them."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "spiking-diffusion_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED, WARMUP, STEPS, SETUP_DRAWS = 42, 5, 20, 2
B, SAMPLE_STEPS, K, L, T = 256, 100, 128, 7, 16


def main():
    from oracle import snn_ref as ref
    from spkdiff import ops, synth
    from spkdiff import dist as sdist
    from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional
    dev = torch.device("cuda:0")
    cfg = synth.MNIST
    sd = synth.synth_denoiser_state(cfg)
    den = DummyModel(1, K).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(sd)
    den.eval()
    HW = L * L
    out = {"config": np.array([SEED, WARMUP, STEPS, B, SAMPLE_STEPS, T, SETUP_DRAWS], dtype=np.int64),
           "weights_checksum": np.frombuffer(synth.state_checksum(sd).encode(), dtype=np.uint8)}
    log = []
    quick = os.environ.get("F15_QUICK") == "1"            # (smoke run of this script: 3 reverse steps)
    steps = 3 if quick else SAMPLE_STEPS
    for tag, skip in (("first", SETUP_DRAWS + WARMUP), ("last", SETUP_DRAWS + WARMUP + STEPS - 1)):
        ab = AbsorbingDiffusion(den, mask_id=K)
        ab.n_samples = B
        ab.set_shard(0, B)
        ab.sync_key = False
        ab.skip_untouched = False
        torch.manual_seed(SEED)
        for _ in range(skip):
            ab._philox_key()
        st = torch.get_rng_state()
        key = ab._philox_key()
        torch.set_rng_state(st)
        hip = ab.sample(temp=1.0, sample_steps=steps).cpu().reshape(B, HW)
        assert ab.last_key == key

        def noise(t, first=0, n=B):
            u, q = ops.philox_noise(key, (steps - t) * (1 << 40) + first * HW * K, n, HW, K, dev)
            return u.cpu().view(n, 1, L, L), q.cpu()
        t0 = time.time()
        rec = []
        want = ref.absorbing_sample(sd, B, K, 1.0, steps, L, T, noise=noise, record=rec).reshape(B, HW)
        dt = time.time() - t0
        bad = sorted(set(torch.nonzero((hip != want).any(1)).flatten().tolist()))
        frag_img, frag_tok, frag_step, frag_margin = [], [], [], []
        for b in bad:
            rx = []
            tx = ref.absorbing_sample(sd, 1, K, 1.0, steps, L, T, noise=lambda t: noise(t, b, 1), record=rx,
                                      exact_conv=True).reshape(HW)
            # the reverse step where the fp32 oracle and the exact evaluation part, and the margin of the flipped spike there
            k = next(i for i in range(steps) if not torch.equal(rx[i][1][0], rec[i][1][b]))
            t = rx[k][0]
            x_in = rx[k - 1][1] if k > 0 else torch.full((1, 1, L, L), K, dtype=torch.int64)
            tt = torch.full((1,), t, dtype=torch.long)
            with torch.inference_mode():
                xb = rec[k - 1][1] if k > 0 else torch.full((B, 1, L, L), K, dtype=torch.int64)
                _, lay_all = ref.denoiser_forward(xb.float(), torch.full((B,), t, dtype=torch.long), sd, T, return_layers=True)
                lay32 = [(a[:, b:b + 1].clone(), y[:, b:b + 1].clone()) for a, y in lay_all]
                del lay_all
                _, layx = ref.denoiser_forward(x_in.float(), tt, sd, T, return_layers=True, exact_conv=True)
            margin = float("nan")
            for (s32, _), (sx, yx) in zip(lay32, layx):
                diff = s32 != sx
                if bool(diff.any()):
                    v = torch.zeros_like(yx[0]); hs = []
                    for ts in range(T):
                        h = v + (yx[ts] - v) * 0.5
                        hs.append(h); v = torch.where(h >= 1.0, torch.zeros_like(h), h)
                    first = int(torch.nonzero(diff.flatten(1).any(1)).min())
                    margin = float((torch.stack(hs)[first] - 1.0).abs()[diff[first]].max())
                    break
            frag_img.append(b); frag_tok.append(tx.numpy().astype(np.uint8)); frag_step.append(t); frag_margin.append(margin)
            log.append(f"{tag}: image {b}: fp32 oracle parts from the exact evaluation at reverse step {t}, flipped spike's exact "
                       f"margin {margin:.3g}; HIP == exact-convolution oracle: {bool(torch.equal(hip[b], tx))}; "
                       f"tokens differing HIP vs fp32 oracle: {int((hip[b] != want[b]).sum())}")
        out[f"{tag}_key"] = np.array([key], dtype=np.int64)
        out[f"{tag}_tokens_oracle"] = want.numpy().astype(np.uint8)
        out[f"{tag}_image_checksum"] = np.array([sdist.token_checksum(want[i:i + 1], i) for i in range(B)], dtype=np.int64)
        out[f"{tag}_fragile_images"] = np.array(frag_img, dtype=np.int64)
        out[f"{tag}_fragile_tokens_exact"] = (np.stack(frag_tok) if frag_tok else np.zeros((0, HW), dtype=np.uint8))
        out[f"{tag}_fragile_step"] = np.array(frag_step, dtype=np.int64)
        out[f"{tag}_fragile_margin"] = np.array(frag_margin, dtype=np.float64)
        n_eq = int((hip == want).all(1).sum())
        log.append(f"{tag}: key {key}: HIP == fp32 oracle on {n_eq}/{B} images ({int((hip != want).sum())} of {want.numel()} tokens "
                   f"differ), fragile images {frag_img}; oracle {dt:.0f} s on {torch.get_num_threads()} threads")
        print(log[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = "f15_bench_job_tokens" + ("_quick" if quick else "") + ".npz"
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", name), **out)
    with open(os.path.join(ROOT, "gpurun_out", "f15_gen_log.txt"), "w") as f:
        f.write("\n".join(log) + "\n")
    print("\n".join(log))
    print("wrote gpurun_out/" + name + "  (copy to tests/golden/ and commit)")


if __name__ == "__main__":
    main()
