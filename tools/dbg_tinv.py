import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
from spkdiff.ops import IN_TINV, MODE_LIF
dev = torch.device("cuda")
lo, hi = 0x3F700000, 0x40080000
bits = torch.arange(lo, hi, dtype=torch.int64, device=dev)
special = torch.tensor([0.0, -0.0, 1.0, 2.0, float("inf"), -float("inf"), float("nan"), 3.4e38, 1e-45, -5.0, 0.5, 4.0, 1e9], dtype=torch.float32, device=dev)
x = torch.cat([bits.to(torch.int32).view(torch.float32), special])
n = x.numel(); B = (n + 4095) // 4096
xp = torch.cat([x, torch.zeros(B * 4096 - n, device=dev)]).reshape(B, 1, 64, 64).contiguous()
packed = ops.pack_conv_weight(torch.ones(16, 1, 1, 1, device=dev), False)
one, zero = torch.ones(16, device=dev), torch.zeros(16, device=dev)
r = ops.conv_fused(xp, packed, None, in_kind=IN_TINV, T=16, mode=MODE_LIF, k=1, stride=1, pad=0, bn_a=one, bn_b=zero, want_ptc=True, chunk_out=16)
got = r["ptc"][:, 0, :, :, :, 0].reshape(-1, 16)[:n]
ref = ops.lif_fwd(x.unsqueeze(0).repeat(16, 1), torch.zeros(n, device=dev), spike_dtype=ops.SPIKE_U8).t()
idx = (got != ref).any(dim=1).nonzero().flatten()
for i in idx.tolist():
    print(i, n, x[i].item(), hex(x[i:i+1].view(torch.int32).item() & 0xffffffff), got[i].tolist(), ref[i].tolist())
