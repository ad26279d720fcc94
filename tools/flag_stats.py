"""Flagged-neuron fractions and tail-launch times of the certified denoiser layers on a real trajectory (bench.layer_statistics),
for the library given by SPKDIFF_LIB.  usage: python tools/flag_stats.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
import bench
from spkdiff import synth
from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional
dev = torch.device("cuda", 0)
den = DummyModel(1, 128).to(dev)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.cached_state('denoiser', synth.MNIST))
den.eval()
ab = AbsorbingDiffusion(den, mask_id=128)
torch.manual_seed(0)
st = bench.layer_statistics(den, ab, 256, 7, 100)
for n, v in st["certified_layers"].items():
    print(n, "flagged", v["flagged_per_launch"], f"frac {v['flagged_frac']:.2e} repair {v['repair_ms'] * 1e3:.1f} us")
