"""Train the two models of the path the way R/main.py does -- on procedurally generated stroke images, there being no data set
and no network on the build machines -- and save plain state_dicts with the reference's keys.

  stage 1  R/main.py:100-146   SNN_VQVAE, AdamW(lr 1e-3, betas (0.9, 0.999), weight_decay 1e-3), batch 32,
                               loss = loss_eq + loss_rec, reset_net after every step
  stage 2  R/main.py:202-252   get_data_for_diff (code indices of the training set), DummyModel + AbsorbingDiffusion,
                               same optimizer, batch 32, loss = train_iter(indices)['loss']

Both start from the synthetic, BN-calibrated initialisation (spkdiff/synth.py: an un-calibrated random init is degenerate,
SURVEY.md App. B.7).  Runs on one MI355X through this build's training path (native BN+LIF block tails, exact MFMA forward,
native weight / data gradients); the resulting fp32 tensors are what matters -- they are committed under
spiking-diffusion_amd/checkpoints/ and loaded bit for bit into the REAL reference by oracle/gen_golden.py for the
``*_trained`` fixtures.

usage (GPU box): python tools/train_on_strokes.py [--vae-iters 3000] [--den-iters 12000] [--out gpurun_out/trained]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]

import numpy as np
import torch


def strokes(n, seed, chunk=512):
    from spkdiff import synth
    return torch.cat([synth.stroke_images(min(chunk, n - i), seed + i) for i in range(0, n, chunk)], 0)


def save_npz(path, sd):
    np.savez(path, **{k: v.detach().float().cpu().numpy() if v.is_floating_point() else v.detach().cpu().numpy()
                      for k, v in sd.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vae-iters", type=int, default=3000)
    ap.add_argument("--den-iters", type=int, default=12000)
    ap.add_argument("--n-images", type=int, default=8192)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trained"))
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    from spkdiff import synth, ops
    from snn_model.vae_model import SNN_VQVAE
    from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional, get_data_for_diff
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    cfg = synth.MNIST
    log = {"args": vars(args), "vae": [], "den": []}
    data = strokes(args.n_images, 2024)                                   # [N,1,28,28] in [0,1]
    var = float((data - 0.5).var())                                       # train_data_variance of R/main.py
    log["data"] = {"n": int(data.shape[0]), "mean": float(data.mean()), "variance": var}
    B = 32

    # ---- stage 1: the spiking VQ-VAE (R/main.py:100-146)
    model = SNN_VQVAE(1, cfg.latent_dim, cfg.num_embeddings, var).to(dev)
    functional.set_step_mode(net=model, step_mode='m')
    model.load_state_dict(synth.cached_state('vqvae', cfg))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    g = torch.Generator().manual_seed(args.seed)
    t0 = time.time()
    model.train()
    for it in range(args.vae_iters):
        idx = torch.randint(0, data.shape[0], (B,), generator=g)
        images = (data[idx] - 0.5).to(dev)
        images_spike = images.unsqueeze(0).repeat(16, 1, 1, 1, 1)
        loss_eq, loss_rec, real = model(images_spike, images)
        opt.zero_grad()
        (loss_eq + loss_rec).backward()
        opt.step()
        functional.reset_net(model)
        if it % 100 == 0 or it == args.vae_iters - 1:
            log["vae"].append([it, float(loss_eq), float(loss_rec), float(real)])
            print(f"vae {it:5d} loss_eq {float(loss_eq):.4f} loss_rec {float(loss_rec):.4f} mse {float(real):.5f}", flush=True)
    log["vae_seconds"] = time.time() - t0
    model.eval()
    sd_v = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    save_npz(os.path.join(args.out, "mnist_strokes_vqvae.npz"), sd_v)

    # ---- stage 2: code indices of the training set, then the denoiser (R/main.py:202-252)
    loader = [(data[i:i + B], None) for i in range(0, data.shape[0] - B + 1, B)]
    functional.reset_net(model)
    train_indices = get_data_for_diff(loader, model)                      # list of [32,7,7] int64 (state carried, as there)
    functional.reset_net(model)
    allidx = torch.stack(train_indices)
    used = int(torch.unique(allidx).numel())
    log["codes_used"] = used
    print("codes used:", used, "of", cfg.num_embeddings, flush=True)
    den = DummyModel(1, cfg.num_embeddings).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.cached_state('denoiser', cfg))
    ab = AbsorbingDiffusion(den, mask_id=cfg.num_embeddings)
    opt = torch.optim.AdamW(den.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    den.train()
    t0 = time.time()
    it = 0
    while it < args.den_iters:
        for indices in train_indices:
            x = indices.float().to(dev).unsqueeze(1)
            loss = ab.train_iter(x)['loss']
            opt.zero_grad()
            loss.backward()
            opt.step()
            functional.reset_net(net=den)
            if it % 200 == 0 or it == args.den_iters - 1:
                log["den"].append([it, float(loss)])
                print(f"den {it:6d} loss {float(loss):.4f}", flush=True)
            it += 1
            if it >= args.den_iters:
                break
    log["den_seconds"] = time.time() - t0
    den.eval()
    sd_d = {k: v.detach().cpu().contiguous() for k, v in den.state_dict().items()}
    save_npz(os.path.join(args.out, "mnist_strokes_denoiser.npz"), sd_d)

    # ---- what the trained weights look like (the axes the exactness claim of the MFMA kernels depends on)
    stats = {}
    for name, sd in (("vqvae", sd_v), ("denoiser", sd_d)):
        for k, v in sd.items():
            if k.endswith(".weight") and v.dim() == 4:
                w = v.float()
                co = w.shape[0] if "decoder" not in k else w.shape[1]
                wc = w.reshape(w.shape[0], -1) if "decoder" not in k else w.transpose(0, 1).reshape(w.shape[1], -1)
                mx = wc.abs().amax(1)
                stats[f"{name}.{k}"] = {"channels": int(co), "max_abs": float(mx.max()), "min_channel_max": float(mx.min()),
                                        "kurtosis": float(((wc - wc.mean()) ** 4).mean() / (wc.var() ** 2 + 1e-30)),
                                        "frac_below_2^-6_of_channel_max": float((wc.abs() < mx[:, None] / 64).float().mean())}
            if k.endswith("1.weight") and v.dim() == 1 or ".running_var" in k:
                stats[f"{name}.{k}"] = {"min": float(v.min()), "max": float(v.max())}
    log["weight_stats"] = stats
    # a sample with the trained pair: 100 reverse steps + decode, and the firing rates / flagged fractions of the denoiser
    ab.n_samples = 64
    torch.manual_seed(1)
    tok = ab.sample(temp=1.0, sample_steps=100)
    pred, u8 = model.decode_tokens(tok.reshape(64, 7, 7))
    log["sample"] = {"pixel_mean": float(u8.float().mean() / 255), "tokens_used": int(torch.unique(tok).numel())}
    try:
        import bench
        st = bench.layer_statistics(den, ab, 64, 7, 100)
        log["layer_statistics"] = st
    except Exception as e:
        log["layer_statistics"] = {"error": repr(e)}
    np.save(os.path.join(args.out, "sample_u8.npy"), u8.cpu().numpy())
    with open(os.path.join(args.out, "train_log.json"), "w") as f:
        json.dump(log, f, indent=1)
    print(json.dumps({k: log[k] for k in ("data", "codes_used", "vae_seconds", "den_seconds", "sample")}))


if __name__ == "__main__":
    main()
