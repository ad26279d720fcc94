"""One warm + N timed sampler calls for rocprofv3 (`--kernel-trace --stats`): the headline job with the untouched-image
elimination on (default) or off (`--dense`).  usage: python tools/sample_profile.py [--dense] [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import synth
from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional

n = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 3
den = DummyModel(1, 128).cuda(0)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
den.eval()
ab = AbsorbingDiffusion(den, mask_id=128)
ab.n_samples = 256
ab.skip_untouched = "--dense" not in sys.argv
ab.sample(temp=1.0, sample_steps=100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ab.sample(temp=1.0, sample_steps=100)
torch.cuda.synchronize()
print(f"skip_untouched={ab.skip_untouched}: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per 256-image reverse process")
