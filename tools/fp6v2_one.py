"""A few launches of spk_den_conv3x3_mfma_fp6v2 at one shape (rocprofv3 counter passes): fp6v2_one.py Cout Cin [n] [mnist|cifar].
Inputs are a real denoiser layer's: the synthetic BN-calibrated checkpoint and the spikes a mid-trajectory call produces
(mnist: 7x7 latents, B = 256; cifar: 8x8 latents, B = 512 -- BASELINE configs[3])."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops, synth
from snn_model.vq_diffusion import DummyModel, functional
Cout, Cin = int(sys.argv[1]), int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = synth.CIFAR if (len(sys.argv) > 4 and sys.argv[4] == "cifar") else synth.MNIST
dev = torch.device("cuda"); B = 512 if cfg is synth.CIFAR else 256; L = cfg.latent
den = DummyModel(1, 128).to(dev)
functional.set_step_mode(net=den, step_mode='m')
den.load_state_dict(synth.synth_denoiser_state(cfg))
den.eval()
g = torch.Generator().manual_seed(1)
x_t = torch.randint(0, 128, (B, 1, L, L), generator=g)
x_t[torch.rand(B, 1, L, L, generator=g) < 0.5] = 128
rec = []
with torch.inference_mode():
    den.logits_from_tokens(x_t.to(dev), 50, record=rec)
blk = {(128, 64): (den.conv2, 0), (256, 128): (den.conv3, 1), (512, 256): (den.conv4, 2), (256, 512): (den.conv5, 3)}[(Cout, Cin)]
conv, bn = blk[0][0], blk[0][1]
x = rec[blk[1]]
a, b = bn.affine_terms()
packed = conv._spk_params.get_fp6v2(conv)
for _ in range(n):
    y = ops.den_conv3x3_mfma_fp6v2(x, packed, Cout, bn_a=a, bn_b=b)
torch.cuda.synchronize()
print("done", ops.count_spikes(y))
