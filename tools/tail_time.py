"""Time spk_den_step_tail at B=256 (7x7) next to the three launches it replaces, for every libspkdiff variant given
(SPKDIFF_LIB, fresh process each).  usage: python tools/tail_time.py [lib.so ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [os.path.join(ROOT, "spiking-diffusion_amd"), ROOT]
    import torch
    from spkdiff import ops, synth
    from snn_model.vq_diffusion import DummyModel, functional
    dev = torch.device("cuda"); B, L, K = 256, 7, 128
    den = DummyModel(1, K).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.cached_state('denoiser', synth.MNIST))
    den.eval()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randint(0, K, (B, 1, L, L), generator=g)
    un0 = torch.rand(B, 1, L, L, generator=g) < 0.5
    x0[~un0] = K
    x0, un0 = x0.to(dev), un0.to(dev)
    t = 50
    x5, cnt5, x1, cnt1, which, impl, collapse = den._trunk(ops.den_build_input(x0, t), False)
    conv6, packed6 = den._conv6_params()
    conv1, bn1 = den.conv1[0], den.conv1[1]
    a1, b1 = bn1.affine_terms()
    c1 = (conv1._spk_params.get(conv1), conv1.bias.detach(), a1, b1)
    nxt = torch.empty((B, 2, L, L), dtype=torch.float32, device=dev)

    def fused():
        ops.den_step_tail(cnt5, cnt1, packed6, x0.clone(), un0.clone(), t, 1.0, T=16, K=K, seed=1, offset=0, conv1=c1)

    def three():
        lg = ops.den_conv3x3_counts(cnt5, packed6, K, 16, cnt1=cnt1)
        ops.psample_step(lg, x0.clone(), un0.clone(), t, 1.0, None, None, 1, 0, next_input=nxt)
        den.conv1.run(nxt, ops.IN_TINV, final='ptc', T=16, stateful=False, chunk_out=ops.CHUNK_S32, want_counts=True)

    def clones():
        x0.clone(); un0.clone()
    out = []
    for name, fn in (("clones", clones), ("fused", fused), ("three", three)):
        for _ in range(5):
            fn()
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        out.append(f"{name} {sorted(p.elapsed_time(q) for p, q in evs)[15] * 1e3:6.1f} us")
    print(" | ".join(out), flush=True)
else:
    libs = sys.argv[1:] or [os.path.join(ROOT, "spiking-diffusion_amd/spkdiff/libspkdiff.so")]
    for lib in libs:
        env = dict(os.environ, SPKDIFF_LIB=os.path.abspath(lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):24s} {r.stdout.strip() or r.stderr.strip()[-600:]}", flush=True)
