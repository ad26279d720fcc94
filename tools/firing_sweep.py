"""How does the headline depend on the firing rates of the synthetic checkpoint?  

The BatchNorm biases of every denoiser block are shifted by a common amount (the calibrated synthetic checkpoint fires at
3 - 11 %; a positive shift raises every layer's rate), the resulting per-layer rates are MEASURED (spk_count_spikes on a real
trajectory, bench.layer_statistics), and the MNIST 100-step sample is timed dense and with the untouched-image elimination +
position lists.  Prints a markdown table; commit it under profiles/.
usage: python tools/firing_sweep.py [B=256] [reps=3] [shift ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch                                                       # noqa: E402
import bench                                                       # noqa: E402
from spkdiff import synth                                          # noqa: E402
from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional   # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
shifts = [float(x) for x in sys.argv[3:]] or [0.0, 0.15, 0.3, 0.5, 0.8]
base = synth.cached_state('denoiser', synth.MNIST)
rows = []
for sh in shifts:
    sd = {k: v.clone() for k, v in base.items()}
    for i in range(1, 6):
        sd[f"conv{i}.1.bias"] += sh
    den = DummyModel(1, 128).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(sd)
    den.eval()
    times = {}
    for name, skip in (("dense", False), ("elim+lists", True)):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples, ab.skip_untouched = B, skip
        torch.manual_seed(7)
        ab.sample(temp=1.0, sample_steps=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ab.sample(temp=1.0, sample_steps=100)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / reps
        ab._graphs.clear()
    ab = AbsorbingDiffusion(den, mask_id=128)
    st = bench.layer_statistics(den, ab, B, 7, 100)
    fr = st["firing_rates"]
    cl = st.get("certified_layers", {})
    rows.append((sh, fr, cl, times))
    del den, ab
    bench._release()

print(f"| BN bias shift | firing rate conv1..conv5 (%) | flagged fraction conv2..conv5 | repair ms conv4 / conv5 | dense ms / sample "
      f"(images/s) | elimination + lists ms (images/s) |")
print("|---|---|---|---|---|---|")
for sh, fr, cl, times in rows:
    r = " / ".join(f"{100 * fr[n]['mean']:.1f}" for n in sorted(fr))
    fl = " / ".join(f"{cl[n]['flagged_frac']:.1e}" for n in sorted(cl))
    rp = " / ".join(f"{cl[n]['repair_ms']:.3f}" for n in ("den.conv4", "den.conv5") if n in cl)
    print(f"| {sh:+.2f} | {r} | {fl} | {rp} | {times['dense'] * 1e3:.1f} ({B / times['dense']:.0f}) | "
          f"{times['elim+lists'] * 1e3:.1f} ({B / times['elim+lists']:.0f}) |")
