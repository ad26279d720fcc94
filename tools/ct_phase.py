"""Phase stamps of conv_train_gather_kernel from a -DCT_DBG=512 build (s_memrealtime, 10 ns ticks): per workgroup start / tables built /
weights staged / items done.  usage: SPKDIFF_LIB=variants/ct_dbg512.so python tools/ct_phase.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
dev = torch.device("cuda")
for name, cin, cout, k, st, pd, tr, op, H in (("dec.convT2 fwd", 64, 32, 3, 2, 1, True, 1, 14), ("enc.conv2 fwd", 32, 64, 3, 2, 1, False, 0, 14)):
    x = (torch.rand(512, cin, H, H, device=dev) < 0.1).float().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        y = ops.conv_train_forward(x, w, None, st, pd, tr, op)
    torch.cuda.synchronize()
    raw = y.permute(0, 2, 3, 1).contiguous().view(-1).view(torch.int64)[:4 * 512].cpu().view(-1, 4)
    raw = raw[raw[:, 0] > 0]
    t0 = int(raw[:, 0].min())
    d = (raw - t0).double() * 0.01          # us
    print(f"{name}: {len(raw)} workgroups; start {d[:,0].mean():.1f} (max {d[:,0].max():.1f}) | tables +{(d[:,1]-d[:,0]).mean():.1f} | "
          f"staged +{(d[:,2]-d[:,1]).mean():.1f} | items +{(d[:,3]-d[:,2]).mean():.1f} (max {(d[:,3]-d[:,2]).max():.1f}) | end max {d[:,3].max():.1f} us")
