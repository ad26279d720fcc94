"""Time the reverse process (MNIST config, 100 steps): dense, image elimination, image elimination + position lists.
usage: python tools/listed_time.py [B=256] [reps=5] [only=<mode name>]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
import torch
from spkdiff import synth
from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3] if len(sys.argv) > 3 else None
cfg = synth.MNIST
den = DummyModel(1, cfg.num_embeddings).cuda(0)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(cfg))
den.eval()
if os.environ.get('SPKDIFF_NO_TAIL') == '1':          # A/B: the three separate launches instead of the fused step tail
    den.use_step_tail = False
res = {}
for name, skip, lists, radii in (("dense", False, False, 4), ("elim", True, False, 4), ("elim+lists4", True, True, 4),
                                 ("elim+lists3", True, True, 3), ("elim+lists2", True, True, 2), ("elim+lists1", True, True, 1)):
    if only and name != only:
        continue
    ab = AbsorbingDiffusion(den, mask_id=cfg.num_embeddings)
    ab.n_samples = B
    ab.skip_untouched, ab.list_positions, ab.list_radii = skip, lists, radii
    torch.manual_seed(7)
    x = ab.sample(temp=1.0, sample_steps=100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x = ab.sample(temp=1.0, sample_steps=100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    torch.manual_seed(7)
    res[name] = ab.sample(temp=1.0, sample_steps=100).cpu()
    print(f"{name:12s} {dt * 1e3:8.2f} ms / sample  {B / dt:8.1f} images/s", flush=True)
    ab._graphs.clear()
if not only:
    print("tokens equal:", all(torch.equal(res["dense"], v) for v in res.values()))
