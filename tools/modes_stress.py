"""Dense, image-elimination and position-list sampling must give the same tokens for the same seed: modes_stress.py [seeds=10] [B=256]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth
from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = synth.MNIST
den = DummyModel(1, cfg.num_embeddings).cuda(0)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(cfg))
den.eval()
abs_ = {}
for name, skip, lists in (("dense", False, False), ("elim", True, False), ("lists", True, True)):
    ab = AbsorbingDiffusion(den, mask_id=cfg.num_embeddings)
    ab.n_samples = B
    ab.skip_untouched, ab.list_positions = skip, lists
    abs_[name] = ab
bad = 0
for seed in range(seeds):
    out = {}
    for name, ab in abs_.items():
        torch.manual_seed(1234 + seed)
        out[name] = ab.sample(temp=1.0, sample_steps=100).cpu()
    ok = torch.equal(out["dense"], out["elim"]) and torch.equal(out["dense"], out["lists"])
    bad += 0 if ok else 1
    if seed and torch.equal(out["dense"], prev):
        raise SystemExit("two seeds gave the same tokens")
    prev = out["dense"]
print(f"{seeds} seeds x B={B}: {bad} seeds with differing tokens between the modes")
assert bad == 0
