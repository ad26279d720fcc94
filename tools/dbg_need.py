import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "spiking-diffusion_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
from spkdiff import ops
from test_gpu_parity import _cpu_need_lists
dev = torch.device("cuda", 0)
B, t = 37, 40
g = torch.Generator().manual_seed(B * 131 + t)
unmasked = torch.rand(B, 1, 7, 7, generator=g) < 0.4
u = torch.rand(B, 1, 7, 7, generator=g) * (3.0 / t)
um, ud = unmasked.to(dev), u.to(dev)
act = ops.select_active(um, t, ud)
need = ops.NeedLists(B, 4, dev)
ops.select_needed(um, t, act, need, ud)
torch.cuda.synchronize()
n_act = int(act[1][0].item()); active = act[0][:n_act].cpu().tolist()
want = _cpu_need_lists(unmasked.numpy(), u.numpy(), t, active, 4)
for r in range(1, 5):
    rec = need.records(r).cpu().numpy()
    for s in range(n_act):
        lst, last = want[s][r - 1]
        got = rec[s, :rec[s, 48]].tolist()
        if got != lst or rec[s, 50] != int(last):
            b = active[s]
            ch = ((u[b].reshape(7, 7) < 1.0 / t) & ~unmasked[b].reshape(7, 7)).int()
            print("slot", s, "img", b, "r", r, "\n changes\n", ch.numpy(), "\n got", got, rec[s, 50], "\n want", lst, last)
            sys.exit(0)
print("all equal")
