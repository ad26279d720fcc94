"""How many neurons does the fp6v2 kernel flag per layer in the real sampler pipeline (synthetic BN-calibrated weights, B=256,
a mid-trajectory token state)?  Needs a -DSPK_V2_DBG=64 build (SPKDIFF_LIB): the fixup launch then leaves the counter alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops, synth
from snn_model.vq_diffusion import DummyModel, functional
dev = torch.device("cuda")
den = DummyModel(1, 128).to(dev)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
den.eval()
B = 256
g = torch.Generator().manual_seed(1)
orig = ops.den_conv3x3_mfma_fp6v2
log = []
def hooked(in0, packed, Cout, **kw):
    r = orig(in0, packed, Cout, **kw)
    torch.cuda.synchronize()
    tot = sum(int(v[0]) for v in ops._FLAG_DEFAULT.values())
    for v in ops._FLAG_DEFAULT.values():
        v[:2].zero_()
    log.append((Cout, in0.shape[1] * 32, tot, B * Cout * 49))
    return r
ops.den_conv3x3_mfma_fp6v2 = hooked
for t, frac in ((90, 0.9), (50, 0.5), (5, 0.05)):
    x_t = torch.randint(0, 128, (B, 1, 7, 7), generator=g)
    x_t[torch.rand(B, 1, 7, 7, generator=g) < frac] = 128
    log.clear()
    with torch.inference_mode():
        den.logits_from_tokens(x_t.to(dev), t)
    print(f"t={t}: " + " | ".join(f"Cout={c} Cin={ci}: flagged {n} of {tot} ({n / tot:.2e})" for c, ci, n, tot in log), flush=True)
a = [float(getattr(den, f"conv{i}")[1].affine_terms()[0].abs().mean()) for i in range(1, 6)]
print("mean |bn_a| per layer:", [round(x, 2) for x in a])
