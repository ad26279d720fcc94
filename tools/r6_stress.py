"""Round-6 stress (one box): the forms round 6 added, on many random shapes, against kernels that are exact by construction.
  (1) fp6v2 at SMALL batches (B random in 1..48: the half-image item split, the small-batch tail form) with a random id-list capacity
      (-1, 0, 3, 64: the overflow bitmap) -- full batches and active-set calls -- against the six-plane kernel: spikes and counts;
  (2) vae_fp6 with random id-list capacities against the int8 gather kernel;
  (3) the fused step tail for random codebook sizes K in 1..512 against counts-conv6 + spk_psample_step + the first layer's launch:
      logits, tokens, unmasked, the next step's conv1 spikes and counts; Philox and injected noise.
usage: r6_stress.py [seeds=60]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
import torch
from spkdiff import ops
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)

# ---- (1)
total = mism = cmism = flagged = 0
forms = {"half": 0, "whole": 0}
try:
    for seed in range(seeds):
        g = torch.Generator().manual_seed(610000 + seed)
        B = int(torch.randint(1, 49, (1,), generator=g))
        Cout, Cin = ((128, 64), (256, 128), (512, 256), (256, 512))[int(torch.randint(0, 4, (1,), generator=g))]
        cap = (-1, 0, 3, 64)[int(torch.randint(0, 4, (1,), generator=g))]
        wamp = float(10 ** (torch.rand(1, generator=g) * 2.0 - 2.0))
        aamp = float(10 ** (torch.rand(1, generator=g) * 2.3 - 0.7))
        rate = float(10 ** (torch.rand(1, generator=g) * 1.5 - 2.0))
        w = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * wamp
        w[:, :, 1, 1] *= 3.0
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.4
        a = ((torch.rand(Cout, generator=g) - 0.3) * aamp).to(dev)
        b = ((torch.rand(Cout, generator=g) - 0.4) * 2.0).to(dev)
        sd = (torch.rand(16, B, Cin, 7, 7, generator=g) < rate).float().to(dev)
        pk, xs = ops.den_pack_weight_fp6v2(w.to(dev), bias.to(dev)), ops.spikes_to_s32(sd)
        ops.FLAG_CAP = cap
        forms["half" if B * (Cout // 32) * 2 <= 256 else "whole"] += 1
        o2, c2 = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
        n = int(torch.randint(1, B + 1, (1,), generator=g))
        with ops.active_set(torch.arange(B, dtype=torch.int32, device=dev), torch.tensor([n, 0], dtype=torch.int32, device=dev)):
            oa, ca = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
        ops.FLAG_CAP = -1
        o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(w.to(dev), bias.to(dev)), Cout, bn_a=a, bn_b=b,
                                          want_counts=True)
        s1, s2 = ops.c4_to_spikes(o1), ops.s32_to_spikes(o2)
        total += s1.numel() + n * Cout * 49 * 16
        mism += int((s1 != s2).sum()) + int((oa[:n] != o2[:n]).sum())
        cmism += int((c1 != c2).sum()) + int((ca[:n] != c2[:n]).sum())
finally:
    ops.FLAG_CAP = -1
torch.cuda.synchronize()
clean = all(int(v[0]) == 0 and int(v[2 + (1 << 20):].abs().sum()) == 0 for k, v in ops._FLAG_DEFAULT.items() if k[0] == "den")
print(f"(1) fp6v2 small batches + random capacity: neuron-steps {total:.3e}  spike mismatches {mism}  count mismatches {cmism}  "
      f"launch forms {forms}  workspaces clean {clean}")
assert mism == 0 and cmism == 0 and clean

# ---- (2)
total = mism = 0
try:
    for seed in range(seeds):
        g = torch.Generator().manual_seed(620000 + seed)
        layer, hw, Cout = (("dec2", 14, 32), ("dec1", 7, 64), ("enc2", 14, 64), ("dec2", 16, 32), ("dec1", 8, 64), ("enc2", 16, 64))[seed % 6]
        B = int(torch.randint(1, 13, (1,), generator=g))
        transposed = layer != "enc2"
        Cin = {"dec2": 64, "dec1": 16, "enc2": 32}[layer]
        kind = {"dec2": ops.VAE_OUT_COLLAPSED, "dec1": ops.VAE_OUT_S32, "enc2": ops.VAE_OUT_PTC}[layer]
        geo = dict(k=3, stride=2, pad=1, transposed=transposed, out_pad=1 if transposed else 0)
        wamp = float(10 ** (torch.rand(1, generator=g) * 2.0 - 2.0))
        aamp = float(10 ** (torch.rand(1, generator=g) * 2.3 - 0.7))
        rate = float(10 ** (torch.rand(1, generator=g) * 1.5 - 2.0))
        w = (torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * wamp
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.4
        a = ((torch.rand(Cout, generator=g) - 0.3) * aamp).to(dev)
        b = ((torch.rand(Cout, generator=g) - 0.4) * 2.0).to(dev)
        spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < rate).float().to(dev)
        wd, bd = w.to(dev), bias.to(dev)
        ptc = ops.spikes_to_ptc(spikes)
        ops.FLAG_CAP = (-1, 0, 2, 64)[int(torch.randint(0, 4, (1,), generator=g))]
        got = ops.vae_fp6_fwd(ops.ptc_to_s32(ptc), ops.vae_fp6_pack(wd, bd, transposed), Cout, bn_a=a, bn_b=b, transposed=transposed,
                              out_kind=kind, coef=coef if layer == "dec2" else None)
        ops.FLAG_CAP = -1
        pk8 = ops.pack_conv_weight_i8(wd, bd, transposed)
        if layer == "dec2":
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, collapse_coef=coef, **geo)
        else:
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)
            if layer == "dec1":
                got, want = ops.s32_to_spikes(got), ops.ptc_to_spikes(want)
        total += want.numel() * (16 if layer == "dec2" else 1)
        mism += int((want != got).sum())
finally:
    ops.FLAG_CAP = -1
print(f"(2) vae_fp6 + random capacity: neuron-steps {total:.3e}  mismatches {mism}")
assert mism == 0

# ---- (3)
import dataclasses
from spkdiff import synth
from snn_model.vq_diffusion import DummyModel, functional
cases = bad = 0
for seed in range(max(6, seeds // 4)):
    g = torch.Generator().manual_seed(630000 + seed)
    K = int(torch.randint(1, 513, (1,), generator=g))
    L = 7 if seed % 3 else 8
    B = int(torch.randint(1, 9, (1,), generator=g))
    cfg = dataclasses.replace(synth.MNIST if L == 7 else synth.CIFAR, num_embeddings=K)
    den = DummyModel(1, K).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    sd = synth.synth_denoiser_state(cfg) if seed < 3 else None
    if sd is not None:
        den.load_state_dict(sd)
    else:                                                   # (calibrated weights cost seconds per K: mostly random-init + widened logits layer)
        with torch.no_grad():
            den.conv6[0].weight.mul_(20.0)
    den.eval()
    HW = L * L
    for t in (33, 1):
        x0 = torch.randint(0, K, (B, 1, L, L), generator=g)
        un0 = torch.rand(B, 1, L, L, generator=g) < 0.5
        x0[~un0] = K
        x0, un0 = x0.to(dev), un0.to(dev)
        inp = ops.den_build_input(x0, t)
        x5, cnt5, x1, cnt1, which, impl, collapse = den._trunk(inp, False)
        assert which == 'mfma-fp6v2' and collapse and den.tail_fusable(L, L)
        conv6, packed6 = den._conv6_params()
        logits = ops.den_conv3x3_counts(cnt5, packed6, K, 16, cnt1=cnt1)
        inject = seed % 2 == 0
        u = q = None
        if inject:
            u = torch.rand(B * HW, generator=g).to(dev)
            q = torch.empty(B * HW, K).exponential_(1, generator=g).to(dev)
        xa, una = x0.clone(), un0.clone()
        nxt = torch.empty((B, 2, L, L), dtype=torch.float32, device=dev)
        ops.psample_step(logits, xa, una, t, 0.9, u, q, seed=77, offset=1000 * t, next_input=nxt if t > 1 else None)
        xb, unb = x0.clone(), un0.clone()
        conv1, bn1 = den.conv1[0], den.conv1[1]
        a1, b1 = bn1.affine_terms()
        c1 = (conv1._spk_params.get(conv1), conv1.bias.detach(), a1, b1) if t > 1 else None
        pre, lg = ops.den_step_tail(cnt5, cnt1, packed6, xb, unb, t, 0.9, T=16, K=K, u=u, q=q, seed=77, offset=1000 * t, conv1=c1,
                                    want_logits=True)
        ok = torch.equal(lg, logits) and torch.equal(xa, xb) and torch.equal(una, unb) and int(xb.max()) <= K
        if t > 1:
            r1 = den.conv1.run(nxt, ops.IN_TINV, final='ptc', T=16, stateful=False, chunk_out=ops.CHUNK_S32, want_counts=True)
            ok = ok and torch.equal(pre[0], r1['ptc']) and torch.equal(pre[1], r1['cnt'])
        cases += 1
        bad += 0 if ok else 1
        if not ok:
            print("   step tail differs: K", K, "L", L, "B", B, "t", t)
    functional.reset_net(den)
print(f"(3) fused step tail, K in 1..512: cases {cases}  differing {bad}")
assert bad == 0
