"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the C-ABI
(``libspkdiff.so`` via ctypes), against (a) the golden fixtures captured from the real reference and (b) the CPU
oracle run live on the same seeded inputs.

Bars (BASELINE.json north_star): integer / index work bit-exact; decoded pixels within 1e-4 max-abs.
LIF, BN, p_sample, VQ indices are required EXACT.  Convolution pre-activations are compared with the oracle's
fp64 convolution (this library returns the correctly rounded exact dot product) and with the reference's fp32
values within 2e-6 * (1 + |y|); spikes must agree with the reference everywhere outside the recorded *fragile
set* (neuron-steps whose membrane potential is within ``frag_eps`` of the threshold in the reference itself --
there the reference's own fp32 accumulation order decides the spike, SURVEY.md §7 'Hard parts').
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import snn_ref as ref           # noqa: E402  (checker only)
from parity_report import record as parity  # noqa: E402  (one-line JSON parity summary at session end, tests/conftest.py)
from spkdiff import synth                  # noqa: E402


def unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(tuple(shape)).astype(np.float32))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from spkdiff import ops as o
    return o


FLAG_LIST = 1 << 20          # id-list entries of a certified kernel's workspace (FLAG_CAP): [count, published count, ids..., bitmap, ticket]


def flag_ws_clean(v):
    """Workspace of spk_den_conv3x3_mfma_fp6v2 / spk_vae_fp6_fwd after a call: live counter zero, overflow bitmap and ticket zero
    (word 1 keeps the number of neurons the last call flagged; the id list keeps stale ids, which nothing reads)."""
    return int(v[0]) == 0 and int(v[2 + FLAG_LIST:].abs().sum()) == 0


# Fixtures on TRAINED weights (round 4; oracle/gen_golden_trained.py, checkpoints/): tag "mnist_trained" selects them
def trained(tag):
    return tag.endswith("_trained")


def fixture_file(kind, tag):
    """kind 'f3_encode' / 'f4_decode' / 'f5_denoiser' -> file name for a tag ('mnist', 'cifar', 'mnist_trained')."""
    if trained(tag):
        return f"{kind.replace('_', 't_', 1)}_{tag}.npz"
    return f"{kind}_{tag}.npz"


def build_vae(cfg, dev, T=16, weights='synth'):
    from snn_model.vae_model import SNN_VQVAE, functional
    sd = synth.trained_state('vqvae') if weights == 'trained' else synth.synth_vqvae_state(cfg)
    if T != 16:
        sd = dict(sd)
        sd["vq_layer.memout.coef"] = synth.memout_coef(T)
        sd["memout.coef"] = synth.memout_coef(T)
    m = SNN_VQVAE(cfg.in_dim, cfg.latent_dim, cfg.num_embeddings, torch.tensor(1.0), n_steps=T)
    functional.set_step_mode(net=m, step_mode='m')
    m.load_state_dict(sd)
    return m.cuda(0).eval(), sd


def build_den(cfg, dev, weights='synth'):
    from snn_model.vq_diffusion import DummyModel, functional
    sd = synth.trained_state('denoiser') if weights == 'trained' else synth.synth_denoiser_state(cfg)
    d = DummyModel(1, cfg.num_embeddings).cuda(0)
    functional.set_step_mode(net=d, step_mode='m')
    d.load_state_dict(sd)
    return d.eval(), sd


# ------------------------------------------------------------------------------------------------- a1 LIF
def test_lif_f1_exact(golden_dir, dev, ops):
    d = load(golden_dir, "f1_lif.npz")
    x = torch.from_numpy(d["x_seq"]).to(dev)
    v = torch.zeros(x.shape[1], device=dev)
    s = ops.lif_fwd(x, v)
    assert torch.equal(s.cpu(), unpack(d["spikes"], d["spikes_shape"]))
    assert torch.equal(v.cpu(), torch.from_numpy(d["v"]))
    s2 = ops.lif_fwd(x.flip(0).contiguous(), v)                       # state carried
    assert torch.equal(s2.cpu(), unpack(d["spikes_carry"], d["spikes_shape"]))
    assert torch.equal(v.cpu(), torch.from_numpy(d["v_carry"]))


@pytest.mark.parametrize("N", [1, 63, 1000, 1001, 4096 + 3])
@pytest.mark.parametrize("T", [1, 4, 16, 19])
def test_lif_shapes_and_dtypes_vs_oracle(dev, ops, N, T):
    g = torch.Generator().manual_seed(N * 31 + T)
    x = torch.randn(T, N, generator=g) * 1.5
    v0 = torch.randn(N, generator=g) * 0.3
    want_s, want_v = ref.lif_multi_step(x, v0.clone())
    for dt in (ops.SPIKE_F32, ops.SPIKE_U8, ops.SPIKE_BITS):
        v = v0.clone().to(dev)
        s = ops.lif_fwd(x.to(dev), v, spike_dtype=dt).cpu()
        if dt == ops.SPIKE_BITS:
            words = s.numpy().view(np.uint64)
            bits = ((words[:, :, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).reshape(T, -1)[:, :N]
            s = torch.from_numpy(bits.astype(np.float32))
        assert torch.equal(s.float(), want_s), (dt, N, T)
        assert torch.equal(v.cpu(), want_v)


def test_lif_other_parameters_vs_oracle(dev, ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 777, generator=g) * 2
    for tau, vth, vr in ((3.0, 1.0, 0.0), (2.0, 0.5, -0.25), (1.5, 1.25, 0.1)):
        want_s, want_v = ref.lif_multi_step(x, 0.0 if vr == 0 else vr, vth, vr, tau)
        v = torch.full((777,), vr, device=dev)
        s = ops.lif_fwd(x.to(dev), v, tau, vth, vr)
        assert torch.equal(s.cpu(), want_s), (tau, vth, vr)
        assert torch.equal(v.cpu(), want_v)


def test_lifnode_state_lifetime(dev):
    # v is the float 0.0 after reset, a tensor after a forward, and carries across forwards
    from spikingjelly.activation_based import neuron, functional, surrogate
    node = neuron.LIFNode(surrogate_function=surrogate.ATan(), step_mode='m').eval()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(16, 2, 3, 5, 5, generator=g) * 1.5)
    assert node.v == 0.0 and isinstance(node.v, float)
    s1 = node(x.to(dev))
    assert torch.is_tensor(node.v) and node.v.shape == (2, 3, 5, 5)
    s2 = node(x.to(dev))
    o1, ov = ref.lif_multi_step(x)
    o2, ov2 = ref.lif_multi_step(x, ov)
    assert torch.equal(s1.cpu(), o1) and torch.equal(s2.cpu(), o2) and torch.equal(node.v.cpu(), ov2)
    functional.reset_net(node)
    assert node.v == 0.0 and isinstance(node.v, float)
    node.step_mode = 's'
    s = node(x[0].to(dev))
    assert torch.equal(s.cpu(), ref.lif_multi_step(x[:1])[0][0])
    with pytest.raises(ValueError):
        node.step_mode = 'x'
    with pytest.raises(NotImplementedError):
        node.backend = 'cupy'
    with pytest.raises(RuntimeError):
        node.reset(); node.step_mode = 'm'; node(x)        # CPU tensor: no fallback path


# ------------------------------------------------------------------------------------------------- a2 stateless layers
def test_bn_f7_exact(golden_dir, dev):
    from spikingjelly.activation_based import layer
    d = load(golden_dir, "f7_bn.npz")
    C = d["weight"].shape[0]
    bn = layer.BatchNorm2d(C, step_mode='m')
    bn.load_state_dict({k: torch.from_numpy(d[k]) for k in
                        ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")})
    bn = bn.cuda(0).eval()
    y = bn(torch.from_numpy(d["x"]).to(dev))
    assert torch.equal(y.cpu(), torch.from_numpy(d["y"]))
    with pytest.raises(ValueError):
        bn(torch.zeros(2, C, 4, 4, device=dev))               # 'm' mode wants 5-D, layer.py:464


@pytest.mark.parametrize("cin,cout,k,s,p,hw", [(1, 32, 3, 2, 1, 28), (32, 64, 3, 2, 1, 14), (64, 16, 1, 1, 0, 7),
                                                (2, 64, 3, 1, 1, 7), (3, 8, 3, 1, 1, 9)])
def test_conv2d_layer_vs_fp64_oracle(dev, cin, cout, k, s, p, hw):
    from spikingjelly.activation_based import layer
    torch.manual_seed(cin * 100 + cout)
    conv = layer.Conv2d(cin, cout, k, s, p, step_mode='m').eval()      # (train() + autograd = the library operator)
    x = torch.randn(3, 2, cin, hw, hw)
    want64 = ref.seq_conv2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), s, p)
    want32 = ref.seq_conv2d(x, conv.weight.detach(), conv.bias.detach(), s, p)
    y = conv.cuda(0)(x.to(dev)).cpu()
    assert y.shape == want32.shape
    assert float((y - want32).abs().max()) <= 2e-6 * (1 + float(want32.abs().max()))
    exact = want64.float()
    assert float((y != exact).float().mean()) <= 1e-6, "not the correctly rounded exact dot product"
    with pytest.raises(ValueError):
        conv(torch.zeros(2, cin, hw, hw, device=dev))


@pytest.mark.parametrize("cin,cout,k,s,p,op,hw", [(16, 64, 3, 2, 1, 1, 7), (64, 32, 3, 2, 1, 1, 14),
                                                   (32, 1, 3, 1, 1, 0, 28), (32, 3, 3, 1, 1, 0, 8)])
def test_conv_transpose2d_layer_vs_fp64_oracle(dev, cin, cout, k, s, p, op, hw):
    from spikingjelly.activation_based import layer
    torch.manual_seed(cin * 100 + cout)
    conv = layer.ConvTranspose2d(cin, cout, k, s, p, op, step_mode='m').eval()
    x = torch.randn(2, 2, cin, hw, hw)
    want64 = ref.seq_conv_transpose2d(x.double(), conv.weight.detach().double(), conv.bias.detach().double(), s, p, op)
    want32 = ref.seq_conv_transpose2d(x, conv.weight.detach(), conv.bias.detach(), s, p, op)
    y = conv.cuda(0)(x.to(dev)).cpu()
    assert y.shape == want32.shape
    assert float((y - want32).abs().max()) <= 2e-6 * (1 + float(want32.abs().max()))
    assert float((y != want64.float()).float().mean()) <= 1e-6


def test_memout_layer(dev):
    from snn_model.snn_layers import MembraneOutputLayer
    m = MembraneOutputLayer().cuda(0)
    assert tuple(m.coef.shape) == (16, 1, 1, 1, 1) and "coef" in m.state_dict()
    x = torch.randn(16, 3, 2, 5, 7)
    assert float((m(x.to(dev)).cpu() - ref.membrane_output(x)).abs().max()) <= 2e-6
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 1, 1, 2, 2, device=dev))            # T=4 against the 16-step buffer: reference raises too


def test_ptc_roundtrip(dev, ops):
    s = (torch.rand(16, 3, 8, 5, 5) < 0.1).float()
    p = ops.spikes_to_ptc(s.to(dev))
    assert p.shape == (3, 5, 5, 16, 8)
    assert torch.equal(p.cpu().permute(3, 0, 4, 1, 2).float(), s)
    assert torch.equal(ops.ptc_to_spikes(p).cpu(), s)


# ------------------------------------------------------------------------------------------------- F2 fused layers
def test_f2_fused_layers_teacher_forced(golden_dir, dev, ops):
    d = load(golden_dir, "f2_layers_mnist.npz")
    model, sd = build_vae(synth.MNIST, dev)
    assert synth.state_checksum(sd) == str(d["weights_crc"])
    from spkdiff.fused import FusedSequential
    from spkdiff.ops import IN_PTC, IN_SEQ, IN_TINV
    enc, dec, poi = model.encoder.snn_convs, model.decoder.snn_convs, model.vq_layer.poisson
    blocks = {
        "enc1": FusedSequential(*list(enc)[0:3]), "enc2": FusedSequential(*list(enc)[3:6]),
        "enc3": FusedSequential(*list(enc)[6:9]), "poisson": poi,
        "dec1": FusedSequential(*list(dec)[0:3]), "dec2": FusedSequential(*list(dec)[3:6]),
    }
    report = {}
    for name, blk in blocks.items():
        want = unpack(d[name + "_out_bits"], d[name + "_out_shape"])
        frag = unpack(d[name + "_frag_bits"], d[name + "_out_shape"]).bool()
        if name + "_in" in d:                                  # time-invariant fp32 input
            inp = torch.from_numpy(d[name + "_in"]).to(dev)
            r = blk.run(inp, IN_TINV, final='both', T=16, stateful=False, want_pre=True)
            pre = r['pre'][0].cpu().unsqueeze(0)
            # the same input through the general fp32-sequence path must give identical spikes
            r2 = blk.run(inp.unsqueeze(0).repeat(16, 1, 1, 1, 1).contiguous(), IN_SEQ, final='f32', stateful=False)
            assert torch.equal(r2['f32'], r['f32'])
        else:
            inp = unpack(d[name + "_in_bits"], d[name + "_in_shape"])
            r = blk.run(ops.spikes_to_ptc(inp.to(dev)), IN_PTC, final='both', stateful=False, want_pre=True)
            pre = r['pre'][0].cpu()
        got = r['f32'].cpu()
        assert torch.equal(ops.ptc_to_spikes(r['ptc']).cpu(), got), name      # both output formats agree
        bad = (got != want)
        report[name] = (int(bad.sum()), int(frag.sum()))
        assert not bool((bad & ~frag).any()), f"{name}: spike differs from the reference outside the fragile set"
        ywant = torch.from_numpy(d[name + "_y_b0"])
        ny = ywant.shape[0]
        ygot = pre[:ny, 0] if pre.shape[0] >= ny else pre[0, 0].unsqueeze(0).expand(ny, -1, -1, -1)
        assert float((ygot - ywant).abs().max()) <= 2e-6 * (1 + float(ywant.abs().max())), name
    print("F2 spike mismatches (all inside fragile set) / fragile-set size:", report)
    parity("f2_fused_layers_teacher_forced", spike_mismatches={k: v[0] for k, v in report.items()},
           fragile_set={k: v[1] for k, v in report.items()})
    assert all(v[0] == 0 for v in report.values()), "measured: 0 spike mismatches, even inside the fragile sets"
    # dec3 + membrane read-out (conv only, fused with sum_t coef[t] x[t])
    inp = unpack(d["dec3_in_bits"], d["dec3_in_shape"])
    dec3 = FusedSequential(list(dec)[6])
    r = dec3.run(ops.spikes_to_ptc(inp.to(dev)), IN_PTC, final='memout', coef=model.memout.coef.flatten())
    assert float((r['f32'].cpu() - torch.from_numpy(d["memout"])).abs().max()) <= 1e-5
    raw = dec3.run(ops.spikes_to_ptc(inp.to(dev)), IN_PTC, final='f32')['f32'].cpu()
    ywant = torch.from_numpy(d["dec3_y_b0"])
    assert float((raw[:, 0] - ywant).abs().max()) <= 2e-6 * (1 + float(ywant.abs().max()))


# ------------------------------------------------------------------------------------------------- F3 encode
@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR), ("mnist_trained", synth.MNIST)])
def test_f3_encode_decode_end_to_end(golden_dir, dev, tag, cfg):
    from snn_model.vae_model import functional
    d = load(golden_dir, fixture_file("f3_encode", tag))
    model, sd = build_vae(cfg, dev, weights='trained' if trained(tag) else 'synth')
    assert synth.state_checksum(sd) == str(d["weights_crc"])
    images = torch.from_numpy(d["images"])
    B = images.shape[0]
    x = images.unsqueeze(0).repeat(16, 1, 1, 1, 1)
    with torch.inference_mode():
        e, xr, idx = model(x.to(dev), images.to(dev))
        functional.reset_net(model)
    L = cfg.latent
    want_idx = torch.from_numpy(d["indices"]).reshape(B, L * L)
    got_idx = idx.cpu().reshape(B, L * L)
    want_xr = torch.from_numpy(d["x_recon"])
    # an image is "clean" when no neuron-step of the reference sits within 2e-5 of the threshold; the live oracle
    # tells which images are clean (per-image margins over all six LIF layers)
    with torch.inference_mode():
        z, enc_layers = ref.encoder_forward(x, sd, return_layers=True)
        q = torch.nn.functional.embedding(want_idx.reshape(-1), sd["vq_layer.embeddings.weight"]).view(
            B, L, L, -1).permute(0, 3, 1, 2).contiguous()
        pe, py = ref.poisson_forward(q, sd, 16)
        _, dec_layers = ref.decoder_forward(pe, sd, return_layers=True)
    margin = torch.full((B,), 1e9)
    for (_, y) in list(enc_layers) + [(pe, py)] + list(dec_layers):
        v = torch.zeros_like(y[0])
        for t in range(16):
            h = v + (y[t] - v) / 2
            margin = torch.minimum(margin, (h - 1).abs().flatten(1).min(1).values)
            s = (h >= 1).float()
            v = (1 - s) * h
    clean = margin > 3e-6
    same_idx = (got_idx == want_idx).all(1)
    err = (xr.cpu() - want_xr).abs().flatten(1).max(1).values
    print(f"F3 {tag}: {int(clean.sum())}/{B} clean images (margin > 3e-6); index-exact {int(same_idx.sum())}/{B}; "
          f"pixels within 1e-4 on {int((err <= 1e-4).sum())}/{B}; max recon err {float(err.max()):.2e}; "
          f"per-image margins {[f'{m:.1e}' for m in margin.tolist()]}")
    parity("f3_encode_" + tag, images=B, index_exact_images=int(same_idx.sum()), pixels_within_1e_4=int((err <= 1e-4).sum()),
           max_recon_err=float(err.max()), clean_images=int(clean.sum()))
    # Measured on MI355X: every image index-exact and within 1e-4, the non-"clean" ones included (this library returns the
    # correctly rounded exact dot product; the margins above say where the REFERENCE's own fp32 order could decide a spike).
    # The kernels and the fixtures are deterministic, so the measured value is the bar.
    assert bool(same_idx.all()), "code indices must be bit-exact on every image"
    assert bool((err <= 1e-4).all()), "decoded pixels must be within 1e-4 on every image"
    # the time-invariant fast path (repeat folded into the kernel) gives the same indices as the module call
    with torch.inference_mode():
        idx2 = model.encode_images(images.to(dev), 16)
    assert torch.equal(idx2.reshape(B, L * L).cpu(), got_idx)


def test_vq_argmin_and_quantize(dev):
    model, sd = build_vae(synth.MNIST, dev)
    g = torch.Generator().manual_seed(3)
    flat = torch.rand(500, 16, generator=g) * 2
    cb = sd["vq_layer.embeddings.weight"]
    d64 = ref.vq_distances(flat.double(), cb.double())
    got = model.vq_layer.get_code_indices(flat.to(dev)).cpu()
    assert got.dtype == torch.int64 and torch.equal(got, d64.argmin(1))
    gap = torch.topk(ref.vq_distances(flat, cb), 2, dim=1, largest=False).values
    safe = (gap[:, 1] - gap[:, 0]) > 1e-4
    assert torch.equal(got[safe], ref.vq_code_indices(flat, cb)[safe])
    tok = torch.randint(0, 128, (3, 7, 7), generator=g)
    qz = model.vq_layer.quantize(tok.to(dev)).cpu()
    assert qz.shape == (3, 7, 7, 16) and torch.equal(qz, torch.nn.functional.embedding(tok, cb))
    # ties -> first index, like torch.argmin
    model.vq_layer.embeddings.weight.data[5] = model.vq_layer.embeddings.weight.data[77]
    x = model.vq_layer.embeddings.weight.data[77:78].clone()
    assert int(model.vq_layer.get_code_indices(x)) == 5


# ------------------------------------------------------------------------------------------------- F4 decode glue
@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR), ("mnist_trained", synth.MNIST)])
def test_f4_decode_glue(golden_dir, dev, tag, cfg):
    from snn_model.vae_model import functional
    d = load(golden_dir, fixture_file("f4_decode", tag))
    model, sd = build_vae(cfg, dev, weights='trained' if trained(tag) else 'synth')
    tokens = torch.from_numpy(d["tokens"])
    want = torch.from_numpy(d["pred"])
    # (1) the reference's own call sequence, R/main.py:388-401, on the drop-in modules
    with torch.inference_mode():
        z = model.vq_layer.quantize(tokens.cuda(0))
        z = z.permute(0, 3, 1, 2).contiguous()
        quantized = torch.unsqueeze(z, dim=0)
        quantized = quantized.repeat(16, 1, 1, 1, 1)
        quantized = model.vq_layer.poisson(quantized)
        pred = model.decoder(quantized)
        pred = torch.tanh(model.memout(pred))
    generated = np.array(np.clip((pred + 0.5).cpu().numpy(), 0., 1.) * 255, dtype=np.uint8)
    functional.reset_net(model)
    err = (pred.cpu() - want).abs().flatten(1).max(1).values
    # (2) the fused three-launch path
    pred2, u8 = model.decode_tokens(tokens.cuda(0))
    err2 = (pred2.cpu() - want).abs().flatten(1).max(1).values
    print(f"F4 {tag}: per-image max err (module sequence) {err.tolist()}, fused {err2.tolist()}")
    ok = (err <= 1e-4)
    parity("f4_decode_" + tag, images=len(err), within_1e_4_module_sequence=int(ok.sum()), within_1e_4_fused=int((err2 <= 1e-4).sum()),
           max_err=float(max(err.max(), err2.max())))
    assert bool(ok.all()) and bool((err2 <= 1e-4).all())
    assert float((pred2.cpu() - pred.cpu()).abs().max()) <= 1e-5
    safe = torch.from_numpy(d["u8_edge_dist"] > 1e-3) & ok.view(-1, 1, 1, 1)
    assert np.array_equal(generated[safe.numpy()], d["u8"][safe.numpy()])
    assert np.array_equal(u8.cpu().numpy()[safe.numpy()], d["u8"][safe.numpy()])


# ------------------------------------------------------------------------------------------------- F5 denoiser
@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR), ("mnist_trained", synth.MNIST)])
def test_f5_denoiser(golden_dir, dev, ops, tag, cfg):
    from spkdiff.ops import IN_PTC, IN_TINV
    from snn_model.vq_diffusion import functional
    d = load(golden_dir, fixture_file("f5_denoiser", tag))
    den, sd = build_den(cfg, dev, weights='trained' if trained(tag) else 'synth')
    assert synth.state_checksum(sd) == str(d["weights_crc"])
    x_t = torch.from_numpy(d["x_t"])
    t = torch.from_numpy(d["t"])
    want_logits = torch.from_numpy(d["logits"])
    # (1) teacher-forced per layer: golden input spikes -> this layer's spikes, equal outside the fragile set
    inp0 = ops.den_build_input(x_t.float().to(dev), t.to(dev))
    assert torch.equal(inp0.cpu(), torch.cat((x_t.float(), torch.ones_like(x_t).float() * t.view(-1, 1, 1, 1)), 1))
    prev = None
    report = {}
    for i, blk in enumerate((den.conv1, den.conv2, den.conv3, den.conv4, den.conv5), 1):
        want = unpack(d[f"s{i}_bits"], d[f"s{i}_shape"])
        frag = unpack(d[f"frag{i}_bits"], d[f"s{i}_shape"]).bool()
        if i == 1:
            r = blk.run(inp0, IN_TINV, final='f32', T=16, stateful=False)
        else:
            r = blk.run(ops.spikes_to_ptc(prev.to(dev)), IN_PTC, final='f32', stateful=False)
        got = r['f32'].cpu()
        bad = got != want
        report[f"conv{i}"] = (int(bad.sum()), int(frag.sum()), int(want.sum()))
        assert not bool((bad & ~frag).any()), f"conv{i}: spike differs outside the fragile set"
        prev = want
    print(f"F5 {tag} teacher-forced (mismatches, fragile, spikes):", report)
    report_tf = report
    s5 = unpack(d["s5_bits"], d["s5_shape"]); s1 = unpack(d["s1_bits"], d["s1_shape"])
    lg = den.conv6.run(ops.spikes_to_ptc(s5.to(dev)), IN_PTC, final='mean', in1=ops.spikes_to_ptc(s1.to(dev)))['f32']
    # (trained weights: conv6's weights reach 4.4 and the logits tens -- the same few ulp are a larger absolute number)
    tol5 = 2e-6 * (1.0 + want_logits.abs()) if trained(tag) else torch.full_like(want_logits, 1e-5)
    assert bool(((lg.cpu() - want_logits).abs() <= tol5).all()), "conv6 + time mean on the reference's spikes"
    # (2) end to end through the module API (spike flips may cascade: SURVEY.md §7) -- report, and bound loosely
    with torch.inference_mode():
        logits = den(x_t.float().to(dev), t=t.to(dev))
        functional.reset_net(den)
        logits_b = den.logits_from_tokens(x_t.to(dev), 5)
        functional.reset_net(den)
        logits_c = den(x_t.float().to(dev), t=torch.full((x_t.shape[0],), 5, dtype=torch.long, device=dev))
        functional.reset_net(den)
    assert torch.equal(logits_b, logits_c), "sampler fast path == module call + reset_net"
    diff = (logits.cpu() - want_logits).abs()
    frac_close = float((diff <= 1e-4).float().mean())
    print(f"F5 {tag} end-to-end logits: max abs diff {float(diff.max()):.3e}, within 1e-4: {frac_close:.5f}")
    parity("f5_denoiser_" + tag, logits_max_abs_diff=float(diff.max()), frac_within_1e_4=frac_close,
           teacher_forced_mismatches={k: v[0] for k, v in report_tf.items()})
    if trained(tag):
        # the reference's fp32 convolutions and the exact arithmetic agree on every spike of this fixture (oracle, exact_conv):
        # what remains is the rounding of conv6's 2 880-term sums, relative to logits of magnitude ~10
        lg_x, lay_x = ref.denoiser_forward(x_t.float(), t, sd, 16, return_layers=True, exact_conv=True)
        dx = (logits.cpu() - lg_x).abs()
        parity("f5_denoiser_" + tag + "_vs_exact_conv_oracle", logits_max_abs_diff=float(dx.max()),
               logits_max_abs=float(want_logits.abs().max()))
        assert bool((dx <= 1e-6 * (1.0 + lg_x.abs())).all()), "logits == the exact-convolution oracle's to one rounding"
        assert bool((diff <= 2e-6 * (1.0 + want_logits.abs())).all())
    else:
        assert float(diff.max()) <= 1e-6, "end-to-end logits (measured 8.9e-8 / 6.0e-8: no spike flips against the reference)"


@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR), ("mnist_trained", synth.MNIST)])
def test_f5_denoiser_mfma_int8_kernel(golden_dir, dev, ops, tag, cfg):
    """conv2..conv6 on the matrix cores (four exact int8 digit planes) against the golden spikes (teacher forced),
    and bit-for-bit against the fp64-accumulating direct kernel: both are the correctly rounded exact dot product."""
    from spkdiff.ops import IN_PTC
    from snn_model.vq_diffusion import functional
    d = load(golden_dir, fixture_file("f5_denoiser", tag))
    den, sd = build_den(cfg, dev, weights='trained' if trained(tag) else 'synth')
    den.conv_impl_request = 'i8'
    assert den.conv_impl == 'mfma-i8x4'
    spikes = {i: unpack(d[f"s{i}_bits"], d[f"s{i}_shape"]) for i in range(1, 6)}
    report = {}
    for i, blk in enumerate((den.conv2, den.conv3, den.conv4, den.conv5), 2):
        frag = unpack(d[f"frag{i}_bits"], d[f"s{i}_shape"]).bool()
        x_c = ops.spikes_to_ptc(spikes[i - 1].to(dev), chunk=32)
        assert torch.equal(ops.ptc_to_spikes(x_c).cpu(), spikes[i - 1])                       # CPTC round trip
        got_c = blk.run(x_c, IN_PTC, final='ptc', stateful=False, chunk_out=32)['ptc']
        assert got_c.dim() == 6 and got_c.shape[-1] == 32
        got = ops.ptc_to_spikes(got_c).cpu()
        direct = blk.run(x_c, IN_PTC, final='f32', stateful=False, impl='direct')['f32'].cpu()
        bad = got != spikes[i]
        report[f"conv{i}"] = (int(bad.sum()), int(frag.sum()), int((got != direct).sum()))
        assert not bool((bad & ~frag).any()), f"conv{i} (MFMA): spike differs outside the fragile set"
        assert torch.equal(got, direct), f"conv{i}: MFMA int8 path != fp64 direct path"
    print(f"F5 {tag} MFMA teacher-forced (mismatch vs golden, fragile, mismatch vs direct):", report)
    parity("f5_int8_mfma_" + tag, mismatch_vs_golden={k: v[0] for k, v in report.items()},
           mismatch_vs_direct={k: v[2] for k, v in report.items()})
    x5 = ops.spikes_to_ptc(spikes[5].to(dev), chunk=32); x1 = ops.spikes_to_ptc(spikes[1].to(dev), chunk=32)
    lg = den.conv6.run(x5, IN_PTC, final='mean', in1=x1)['f32'].cpu()
    lg_d = den.conv6.run(x5, IN_PTC, final='mean', in1=x1, impl='direct')['f32'].cpu()
    wl = torch.from_numpy(d["logits"])
    assert bool(((lg - wl).abs() <= (2e-6 * (1.0 + wl.abs()) if trained(tag) else 1e-5)).all())
    # weights below 2^-7 of their channel's maximum are fixed-point rounded at 2^-30 of that maximum (den_mfma.hip):
    # a pre-activation may then round to the neighbouring fp32 value; bound: a few percent of logits, by <= 1 ulp
    # (trained weights: 16 % of conv6's weights lie below 2^-6 of their channel's maximum and the logits reach tens: an ulp is
    #  up to 2e-6 there and more sums land on a rounding boundary)
    # (one ulp of the 2 880-term sum before the division by T: take the ulp at the magnitude of the largest logit's scale, >= 6e-8)
    ulp = torch.clamp(2.0 ** (torch.floor(torch.log2(lg_d.abs().clamp_min(1e-30))) - 23), min=6e-8)
    parity("f5_int8_conv6_vs_direct_" + tag, frac_differing=float((lg != lg_d).float().mean()),
           max_abs_diff=float((lg - lg_d).abs().max()), max_abs_logit=float(lg_d.abs().max()))
    # (trained conv6: weights up to 4.4 and logits that are small differences of large terms -- the 2^-30 quantisation of a
    #  channel's small weights, summed over ~300 active counts, is an ABSOLUTE 6e-7 there, more than an ulp of a logit near 1)
    tol_q = 2e-6 * (1.0 + lg_d.abs()) if trained(tag) else ulp
    assert float((lg != lg_d).float().mean()) <= (0.10 if trained(tag) else 0.03) and bool(((lg - lg_d).abs() <= tol_q).all())
    # module semantics on the MFMA path: state carried across forwards without reset == direct path, then reset
    x_t = torch.from_numpy(d["x_t"]).float().to(dev); t = torch.from_numpy(d["t"]).to(dev)
    with torch.inference_mode():
        a1 = den(x_t, t=t); a2 = den(x_t, t=t)
        functional.reset_net(den)
        den.conv_impl_request = 'direct'
        b1 = den(x_t, t=t); b2 = den(x_t, t=t)
        functional.reset_net(den)
        den.conv_impl_request = 'auto'
    tol_ab = 2e-6 * (1.0 + b1.abs()) if trained(tag) else torch.full_like(b1, 1e-6)
    assert bool(((a1 - b1).abs() <= tol_ab).all()) and bool(((a2 - b2).abs() <= 2e-6 * (1.0 + b2.abs()) + 1e-6).all())
    assert not torch.equal(a1, a2), "second call starts from the carried membrane potentials"


@pytest.mark.parametrize("tag", ["mnist", "mnist_trained"])
def test_f5_denoiser_mfma_fp6_kernel(golden_dir, dev, ops, tag):
    """conv2..conv5 on the block-scaled fp6 x fp4 MFMA (six exact radix-32 digit planes, C4 nibble-packed spikes)
    against the golden spikes (teacher forced) and bit-for-bit against the fp64-accumulating direct kernel; spike
    counts, carried membrane state and the whole-call logits against the direct path."""
    from spkdiff.ops import IN_PTC
    from snn_model.vq_diffusion import functional
    d = load(golden_dir, fixture_file("f5_denoiser", tag))
    den, sd = build_den(synth.MNIST, dev, weights='trained' if trained(tag) else 'synth')
    assert den.impl_for(7, 7, stateful=True) == 'mfma-fp6x6' and den.impl_for(7, 7) == 'mfma-fp6v2'
    spikes = {i: unpack(d[f"s{i}_bits"], d[f"s{i}_shape"]) for i in range(1, 6)}
    report, got_all = {}, {}
    for i, blk in enumerate((den.conv2, den.conv3, den.conv4, den.conv5), 2):
        frag = unpack(d[f"frag{i}_bits"], d[f"s{i}_shape"]).bool()
        x_4 = ops.spikes_to_c4(spikes[i - 1].to(dev))
        assert x_4.dtype == ops.C4_DTYPE and x_4.shape[-1] == 32
        assert torch.equal(ops.c4_to_spikes(x_4).cpu(), spikes[i - 1])                        # C4 round trip
        r = blk.run(x_4, IN_PTC, final='ptc', stateful=False, chunk_out=ops.CHUNK_C4, want_counts=True)
        got = ops.c4_to_spikes(r['ptc']).cpu()
        x_c = ops.spikes_to_ptc(spikes[i - 1].to(dev), chunk=32)
        direct = blk.run(x_c, IN_PTC, final='f32', stateful=False, impl='direct')['f32'].cpu()
        bad = got != spikes[i]
        report[f"conv{i}"] = (int(bad.sum()), int(frag.sum()), int((got != direct).sum()))
        assert not bool((bad & ~frag).any()), f"conv{i} (fp6 MFMA): spike differs outside the fragile set"
        assert torch.equal(got, direct), f"conv{i}: fp6 MFMA path != fp64 direct path"
        got_all[i] = got
        cnt = r['cnt'].cpu()                                   # [B, C/32, H, W, 32] == sum over T of the spikes
        B, C, H, W = got.shape[1:]
        want_cnt = got.sum(0).reshape(B, C // 32, 32, H, W).permute(0, 1, 3, 4, 2)
        assert torch.equal(cnt.float(), want_cnt)
    print(f"F5 {tag} fp6-MFMA teacher-forced (mismatch vs golden, fragile, mismatch vs direct):", report)
    # the sampler's kernel (four digits + certification + exact repair) on the same inputs: the six-plane kernel's spikes again
    for i, blk in enumerate((den.conv2, den.conv3, den.conv4, den.conv5), 2):
        r2 = blk.run(ops.spikes_to_s32(spikes[i - 1].to(dev)), IN_PTC, final='ptc', stateful=False, chunk_out=ops.CHUNK_S32)
        assert torch.equal(ops.s32_to_spikes(r2['ptc']).cpu(), got_all[i]), f"conv{i}: fp6v2 != six-plane kernel"
    parity("f5_fp6_mfma_" + tag, mismatch_vs_golden={k: v[0] for k, v in report.items()},
           mismatch_vs_direct={k: v[2] for k, v in report.items()})
    x_t = torch.from_numpy(d["x_t"]).float().to(dev); t = torch.from_numpy(d["t"]).to(dev)
    with torch.inference_mode():
        a1 = den(x_t, t=t); a2 = den(x_t, t=t)
        functional.reset_net(den)
        den.conv_impl_request = 'direct'
        b1 = den(x_t, t=t); b2 = den(x_t, t=t)
        functional.reset_net(den)
        den.conv_impl_request = 'auto'
    wl = torch.from_numpy(d["logits"]).to(dev)
    assert bool(((a1 - wl).abs() <= (2e-6 * (1.0 + wl.abs()) if trained(tag) else 1e-5)).all())
    assert bool(((a1 - b1).abs() <= 1e-6 * (1.0 + b1.abs())).all()) and bool(((a2 - b2).abs() <= 1e-6 * (1.0 + b2.abs())).all())
    assert not torch.equal(a1, a2), "second call starts from the carried membrane potentials"


@pytest.mark.parametrize("B,hw", [(1, 7), (5, 7), (64, 7), (3, 8), (33, 8)])
def test_fp6v2_kernel_bit_equal_to_the_exact_kernels(dev, ops, B, hw):
    """spk_den_conv3x3_mfma_fp6v2 (five digit planes with shared accumulators + certified spike decisions + exact
    recomputation of flagged neurons) against the first-generation fp6 kernel (six planes, fp64 recombination) on the four
    denoiser shapes with random spikes, weights and BatchNorm terms -- including large and negative BN scales, which widen
    the certification margins: spikes and spike counts must be bit-equal, and the flag bitmap must come back clean."""
    g = torch.Generator().manual_seed(1000 + B)
    H = W = hw                                         # 8x8: the row-band form (two items per image, no last-position launch)
    total = mism = 0
    for li, (Cout, Cin) in enumerate(((128, 64), (256, 128), (512, 256), (256, 512))):
        for trial, (wamp, aamp, rate) in enumerate(((0.05, 12.0, 0.06), (0.3, 3.0, 0.3))):
            if B == 64 and li >= 2 and trial == 1:
                continue
            w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * wamp)
            w[:, :, 1, 1] *= 3.0                                      # uneven digit structure across taps
            bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
            a = (torch.rand(Cout, generator=g) - 0.3) * aamp          # some negative BN scales
            b = (torch.rand(Cout, generator=g) - 0.4) * 1.5
            spikes = (torch.rand(16, B, Cin, H, W, generator=g) < rate).float()
            wd, biasd, ad, bd, sd = w.to(dev), bias.to(dev), a.to(dev), b.to(dev), spikes.to(dev)
            o2, c2 = ops.den_conv3x3_mfma_fp6v2(ops.spikes_to_s32(sd), ops.den_pack_weight_fp6v2(wd, biasd), Cout, bn_a=ad,
                                                bn_b=bd, want_counts=True)
            o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(wd, biasd), Cout, bn_a=ad, bn_b=bd,
                                              want_counts=True)
            s2, s1 = ops.s32_to_spikes(o2), ops.c4_to_spikes(o1)
            assert torch.equal(ops.s32_to_spikes(ops.spikes_to_s32(sd)), sd)                  # S32 round trip
            total += s1.numel(); mism += int((s1 != s2).sum())
            assert torch.equal(s1, s2), (Cout, Cin, trial, int((s1 != s2).sum()))
            assert torch.equal(c1, c2)
            assert 0.001 < float(s1.mean()) < 0.9
    torch.cuda.synchronize()
    assert all(flag_ws_clean(v) for k, v in ops._FLAG_DEFAULT.items() if k[0] == "den"), \
        "live counter, overflow bitmap and hand-over ticket come back clean"
    parity(f"fp6v2_vs_fp6_B{B}_{hw}x{hw}", neuron_steps=total, spike_mismatches=mism)


@pytest.mark.parametrize("Cout,Cin", [(128, 64), (256, 512)])
def test_mfma_kernels_wide_dynamic_range_weights(dev, ops, Cout, Cin):
    """The one regime the exactness claim of the MFMA kernels has a caveat for: weights far below their channel's
    maximum are rounded at 2^-29 (fp6 digits) / 2^-30 (int8 digits) of that maximum.  Weights here span 2^-20 .. 1 of the channel
    maximum log-uniformly (synthetic N(0, sigma) weights span ~2^-8), one channel is all but dead (largest weight 1e-12), one has
    a single outlier 2^18 above the rest, BatchNorm scales reach +-28.  Claims checked:
      * fp6v2 (four digits + certification + exact repair) == the six-plane fp6 kernel, bit for bit (same quantised weights);
      * each MFMA family differs from the fp64 direct kernel (TRUE fp32 weights, fp64 sums) only at neuron-steps whose exact
        membrane potential lies within the quantisation bound of the threshold:  |h - 1| <= |a| * n_active * 2^-29 * max|w_c|
        (+ fp32 round-off), n_active = active inputs of the row -- and the spikes are the exact arithmetic's ON THE QUANTISED
        weights (CPU oracle in fp64 on rint(w * 2^s) / 2^s), which is the contract."""
    from spkdiff.ops import IN_PTC, MODE_LIF
    g = torch.Generator().manual_seed(77 + Cout)
    B, H, W = 6, 7, 7
    mag = torch.exp2(-20.0 * torch.rand(Cout, Cin, 3, 3, generator=g))
    w = mag * torch.sign(torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * (0.02 + 0.3 * torch.rand(Cout, 1, 1, 1, generator=g))
    w[3] *= 1e-12 / w[3].abs().max()                                   # a near-dead channel
    w[5] *= 2.0 ** -18
    w[5, 7, 1, 1] = 0.4                                                # one outlier 2^18 above the rest of its channel
    bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
    a = (torch.rand(Cout, generator=g) - 0.3) * 40.0
    b = (torch.rand(Cout, generator=g) - 0.2) * 1.5
    spikes = (torch.rand(16, B, Cin, H, W, generator=g) < 0.05).float()
    wd, biasd, ad, bd, sd = w.to(dev), bias.to(dev), a.to(dev), b.to(dev), spikes.to(dev)
    o2 = ops.den_conv3x3_mfma_fp6v2(ops.spikes_to_s32(sd), ops.den_pack_weight_fp6v2(wd, biasd), Cout, bn_a=ad, bn_b=bd)
    o1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(wd, biasd), Cout, bn_a=ad, bn_b=bd)
    x_c = ops.spikes_to_ptc(sd, chunk=32)
    o8 = ops.den_conv3x3_mfma(x_c, ops.den_pack_weight_i8(wd, biasd), Cout, mode=MODE_LIF, bn_a=ad, bn_b=bd)
    od = ops.conv_fused(x_c, ops.pack_conv_weight(wd, False), biasd, in_kind=IN_PTC, T=16, mode=MODE_LIF, k=3, stride=1, pad=1,
                        bn_a=ad, bn_b=bd, want_f32=True)['f32']
    s_v2, s_fp6, s_i8, s_dir = (ops.s32_to_spikes(o2).cpu(), ops.c4_to_spikes(o1).cpu(), ops.ptc_to_spikes(o8).cpu(), od.cpu())
    assert torch.equal(s_v2, s_fp6), "fp6v2 == six-plane kernel (same quantised weights): bit for bit"

    # CPU: exact pre-activations for the true weights and for the two quantisations, the reference's fp32 BN (fma form) and LIF
    def spikes_of(wq):
        y = torch.nn.functional.conv2d(spikes.flatten(0, 1).double(), wq.double(), bias.double(), 1, 1).float()
        z = (y.double() * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)).float().view(16, B, Cout, H, W)
        v = torch.zeros_like(z[0]); out = []; hs = []
        for t in range(16):
            h = v + (z[t] - v) * 0.5
            sp = h >= 1.0
            hs.append(h); out.append(sp.float()); v = torch.where(sp, torch.zeros_like(h), h)
        return torch.stack(out), torch.stack(hs)

    def quantise(bits):
        m = w.flatten(1).abs().amax(1)
        e = torch.ceil(torch.log2(m.double())).clamp_min(-200)           # m <= 2^e  (frexp: m = f 2^e, f in [0.5, 1))
        e = torch.where(torch.exp2(e) == m.double(), e + 1, e)
        sh = (bits - e).view(-1, 1, 1, 1)
        return (torch.round(w.double() * torch.exp2(sh)) / torch.exp2(sh)).float()
    s_true, h_true = spikes_of(w)
    n_act = torch.nn.functional.conv2d(spikes.flatten(0, 1), torch.ones(1, Cin, 3, 3), None, 1, 1).view(16, B, 1, H, W)
    wmax = w.flatten(1).abs().amax(1).view(1, 1, Cout, 1, 1)
    rep = {}
    for name, got, bits in (("fp6", s_fp6, 29), ("i8", s_i8, 30), ("direct", s_dir, None)):
        if bits is None:
            want, bound = s_true, torch.zeros_like(h_true)
        else:
            want, _ = spikes_of(quantise(bits))
            bound = a.abs().view(1, 1, Cout, 1, 1) * n_act * wmax * 2.0 ** -bits
        bad_q = got != want                                                 # vs the exact arithmetic on the kernel's own weights
        bad_t = got != s_true                                               # vs the exact arithmetic on the true weights
        # a decision may differ from the true-weight result only after some step came within the bound of the threshold
        near = ((h_true - 1.0).abs() <= 2.0 * bound + 4e-6 * (1.0 + h_true.abs())).float().cummax(0).values.bool()
        rep[name] = dict(vs_own_weights=int(bad_q.sum()), vs_true_weights=int(bad_t.sum()),
                         unexplained=int((bad_t & ~near).sum()), near=int(near.sum()))
        assert int((bad_t & ~near).sum()) == 0, (name, rep[name])
        # against the exact arithmetic on its own weights a kernel may differ only inside fp32 round-off of the threshold
        tiny = ((h_true - 1.0).abs() <= 4e-6 * (1.0 + h_true.abs()) + 2.0 * bound).float().cummax(0).values.bool()
        assert int((bad_q & ~tiny).sum()) == 0, (name, rep[name])
    print(f"wide dynamic range {Cout}x{Cin}: firing {float(s_true.mean()):.3f}; mismatching neuron-steps of {s_true.numel()}:", rep)
    parity(f"mfma_wide_dynamic_range_{Cout}x{Cin}", neuron_steps=int(s_true.numel()), firing=float(s_true.mean()), **rep)
    assert 0.002 < float(s_true.mean()) < 0.6
    assert bool((s_fp6[:, :, 3] == s_dir[:, :, 3]).all()), "the near-dead channel: its bias and BatchNorm terms decide alone"


def _cpu_need_lists(unmasked, u, t, active, R):
    """Host restatement of spk_select_needed: per active slot and radius, the sorted positions (< 48) within Chebyshev
    distance r of a change, and whether position 48 is among them."""
    import numpy as np
    out = []
    for b in active:
        ch = ((u[b].reshape(7, 7) < np.float32(1.0) / np.float32(t)) & ~unmasked[b].reshape(7, 7))
        per_r = []
        m = ch.copy()
        for r in range(1, R + 1):
            d = np.zeros_like(m)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    sh = np.zeros_like(m)
                    ys = slice(max(0, dy), 7 + min(0, dy)); yd = slice(max(0, -dy), 7 + min(0, -dy))
                    xs = slice(max(0, dx), 7 + min(0, dx)); xd = slice(max(0, -dx), 7 + min(0, -dx))
                    sh[yd, xd] = m[ys, xs]
                    d |= sh
            m = d
            flat = np.flatnonzero(m.reshape(-1))
            per_r.append(([int(p) for p in flat if p < 48], bool(m[6, 6])))
        out.append(per_r)
    return out


@pytest.mark.parametrize("B,t", [(1, 3), (37, 40), (256, 100), (64, 1)])
def test_select_needed_position_lists(dev, ops, B, t):
    """spk_select_needed against a host restatement: records (sorted positions, padding, count, class, last-position
    flag) for radii 1..4 and the per-class slot lists; two calls on the same buffer (the ticket re-arms itself)."""
    import numpy as np
    g = torch.Generator().manual_seed(B * 131 + t)
    for rep in range(2):
        unmasked = torch.rand(B, 1, 7, 7, generator=g) < 0.4
        u = torch.rand(B, 1, 7, 7, generator=g) * (3.0 / t if t > 1 else 1.0)
        um, ud = unmasked.to(dev), u.to(dev)
        act = ops.select_active(um, t, ud)
        if rep == 0:
            need = ops.NeedLists(B, 4, dev)
        ops.select_needed(um, t, act, need, ud)
        torch.cuda.synchronize()
        n_act = int(act[1][0].item())
        active = act[0][:n_act].cpu().tolist()
        want = _cpu_need_lists(unmasked.numpy(), u.numpy(), t, active, 4)
        assert n_act == int(((u < 1.0 / t) & ~unmasked).flatten(1).any(1).sum())
        buf = need.buf.cpu().numpy()
        R = 4
        for r in range(1, R + 1):
            rec = need.records(r).cpu().numpy()
            classes = [[] for _ in range(6)]
            for s in range(n_act):
                lst, last = want[s][r - 1]
                n = len(lst)
                assert rec[s, 48] == n and rec[s, 50] == int(last)
                assert rec[s, :n].tolist() == lst
                assert all(v == (lst[-1] if n else 0) for v in rec[s, n:48].tolist())
                k = max(1, (n + 7) // 8)
                assert rec[s, 49] == k
                classes[k - 1].append(s)
            cnt = buf[64 + (r - 1) * 64: 64 + (r - 1) * 64 + 24].view(np.int32)
            off = 64 + R * 64 + (r - 1) * 6 * B * 4
            lists = buf[off: off + 6 * B * 4].view(np.int32).reshape(6, B)
            for k in range(6):
                assert cnt[k] == len(classes[k])
                assert sorted(lists[k, :cnt[k]].tolist()) == classes[k]
        assert int(buf[:4].view(np.uint32)[0]) == 0, "ticket re-armed"


@pytest.mark.parametrize("B,t", [(5, 2), (96, 30), (256, 100)])
def test_fp6v2_listed_positions_equal_the_full_layer(dev, ops, B, t):
    """spk_den_conv3x3_mfma_fp6v2_listed: on every listed position (and the 49th) spikes and spike counts are bit-equal
    to the full launch, for each radius / denoiser shape; unlisted positions of the output are left untouched."""
    g = torch.Generator().manual_seed(77 + B)
    unmasked = torch.rand(B, 1, 7, 7, generator=g) < 0.5
    u = torch.rand(B, 1, 7, 7, generator=g) * (2.0 / t)
    um, ud = unmasked.to(dev), u.to(dev)
    act = ops.select_active(um, t, ud)
    need = ops.select_needed(um, t, act, ops.NeedLists(B, 4, dev), ud)
    n_act = int(act[1][0].item())
    assert n_act > 0
    total = 0
    for (Cout, Cin), radius in (((128, 64), 4), ((256, 128), 3), ((512, 256), 2), ((256, 512), 1)):
        w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.1)
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
        a = (torch.rand(Cout, generator=g) - 0.3) * 8.0
        b = (torch.rand(Cout, generator=g) - 0.4) * 1.5
        spikes = (torch.rand(16, B, Cin, 7, 7, generator=g) < 0.15).float()
        pk = ops.den_pack_weight_fp6v2(w.to(dev), bias.to(dev))
        s32 = ops.spikes_to_s32(spikes.to(dev))
        with ops.active_set(*act):
            full, cfull = ops.den_conv3x3_mfma_fp6v2(s32, pk, Cout, bn_a=a.to(dev), bn_b=b.to(dev), want_counts=True)
        with ops.active_set(*act, need=need):
            part, cpart = ops.den_conv3x3_mfma_fp6v2(s32, pk, Cout, bn_a=a.to(dev), bn_b=b.to(dev), want_counts=True,
                                                     need_radius=radius)
        rec = need.records(radius).cpu().numpy()
        listed = torch.zeros(B, 49, dtype=torch.bool)
        for s in range(n_act):
            listed[s, rec[s, :rec[s, 48]].tolist()] = True
            listed[s, 48] = True
        assert 0 < int(listed[:n_act, :48].sum()) < n_act * 48 or radius >= 3
        f = full.cpu().view(B, Cout // 32, 49, 16, 16)
        q = part.cpu().view(B, Cout // 32, 49, 16, 16)
        m = listed[:, None, :, None, None].expand_as(f)
        assert torch.equal(f[m], q[m]), (Cout, Cin, int((f[m] != q[m]).sum()))
        mc = listed[:, None, :, None].expand(B, Cout // 32, 49, 32)
        assert torch.equal(cfull.cpu().view(B, Cout // 32, 49, 32)[mc], cpart.cpu().view(B, Cout // 32, 49, 32)[mc])
        total += int(m.sum()) * 2
    parity(f"fp6v2_listed_vs_full_B{B}_t{t}", neuron_steps=total, spike_mismatches=0)



@pytest.mark.parametrize("B,H,W,Cin,Cout,transposed", [(3, 28, 28, 32, 1, True), (2, 32, 32, 32, 3, True), (5, 9, 13, 16, 2, False)])
def test_collapsed_readout_layer(dev, ops, B, H, W, Cin, Cout, transposed):
    """spk_readout_collapsed_fwd (the decoder's linear last layer applied ONCE to sum_t coef[t] * spikes[t]) against
    sum_t coef[t] * conv(spikes[t]) evaluated frame by frame in fp64; and the producing gather-MFMA layer's collapsed
    output against its own spike frames."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 7 + H)
    T = 16
    spikes = (torch.rand(T, B, Cin, H, W, generator=g) < 0.2).float()
    coef = torch.pow(torch.tensor(0.8), torch.arange(T - 1, -1, -1).float())
    w = (torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * 0.4
    bias = (torch.rand(Cout, generator=g) - 0.5)
    conv = (lambda x: F.conv_transpose2d(x, w.double(), bias.double(), stride=1, padding=1)) if transposed else \
           (lambda x: F.conv2d(x, w.double(), bias.double(), stride=1, padding=1))
    want = sum(coef[t].double() * conv(spikes[t].double()) for t in range(T))
    x = (spikes * coef.view(T, 1, 1, 1, 1)).sum(0).permute(0, 2, 3, 1).contiguous()          # [B,H,W,Cin]
    r = ops.readout_collapsed(x.to(dev), w.to(dev), bias.to(dev), coef, apply_tanh=True, want_u8=True, transposed=transposed)
    err = float((r["f32"].cpu().double() - torch.tanh(want)).abs().max())
    assert err <= 5e-6, err                 # fp32 round-off of pre-activations of magnitude ~10 (ulp 1e-6)
    u8 = (torch.clamp(torch.tanh(want).float() + 0.5, 0, 1) * 255).to(torch.uint8)
    assert int((r["u8"].cpu().int() - u8.int()).abs().max()) <= 1
    parity(f"collapsed_readout_{H}x{W}_c{Cout}", max_abs_err=err)


def test_gather_layer_collapsed_output_matches_its_spikes(dev, ops):
    """spk_conv_mfma_fused_fwd with coef + out_f32 in LIF mode: the collapsed tensor equals sum_t coef[t] * (its own spike
    frames), bit for bit (same fp32 additions in the same order)."""
    g = torch.Generator().manual_seed(5)
    B, Cin, Cout, H = 3, 64, 32, 14
    spikes = (torch.rand(16, B, Cin, H, H, generator=g) < 0.15).float().to(dev)
    w = ((torch.rand(Cin, Cout, 3, 3, generator=g) - 0.5) * 0.3).to(dev)
    bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.1).to(dev)
    a = (torch.rand(Cout, generator=g) * 2 + 0.5).to(dev); b = ((torch.rand(Cout, generator=g) - 0.5)).to(dev)
    coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
    pk = ops.pack_conv_weight_i8(w, bias, True)
    geo = dict(k=3, stride=2, pad=1, transposed=True, out_pad=1)
    ptc = ops.spikes_to_ptc(spikes)
    s = ops.conv_mfma_fused(ptc, pk, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)           # [B,Ho,Wo,16,Cout] u8
    c = ops.conv_mfma_fused(ptc, pk, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, collapse_coef=coef, **geo)
    want = torch.zeros_like(c)
    for t in range(16):
        want = want + s[:, :, :, t, :].float() * coef[t]
    assert torch.equal(c, want) and float(c.max()) > 0



@pytest.mark.parametrize("layer,B,hw,Cout", [("dec2", 1, 14, 32), ("dec2", 5, 14, 32), ("dec2", 37, 14, 32), ("dec2", 3, 16, 32),
                                             ("dec2", 4, 14, 64), ("dec1", 1, 7, 64), ("dec1", 21, 7, 64), ("dec1", 3, 8, 64),
                                             ("enc2", 1, 14, 64), ("enc2", 19, 14, 64), ("enc2", 3, 16, 64)])
def test_vae_fp6_kernel_equals_the_int8_gather_kernel(dev, ops, layer, B, hw, Cout):
    """spk_vae_fp6_fwd (five fp6 digit planes, certified decisions, exact repair) against the int8 gather-MFMA kernel (exact
    by construction) on the VQ-VAE's stride-2 spike-input layers -- decoder convT2 (time-collapsed output), decoder convT1
    (S32 output, Cin = 16 padded to a 32-channel chunk), encoder conv2 (u8 PTC output) -- with random spikes, weights and
    BatchNorm terms including large and negative scales: outputs must be bit-equal and the flag workspace must come back clean."""
    g = torch.Generator().manual_seed(300 + B + hw + len(layer))
    coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
    transposed = layer != "enc2"
    Cin = {"dec2": 64, "dec1": 16, "enc2": 32}[layer]
    kind = {"dec2": ops.VAE_OUT_COLLAPSED, "dec1": ops.VAE_OUT_S32, "enc2": ops.VAE_OUT_PTC}[layer]
    geo = dict(k=3, stride=2, pad=1, transposed=transposed, out_pad=1 if transposed else 0)
    assert ops.vae_fp6_kind(Cin, Cout, 3, 2, 1, geo["out_pad"], transposed, 16, hw, hw) == kind
    total = mism = 0
    for trial, (wamp, aamp, rate) in enumerate(((0.08, 10.0, 0.08), (0.4, 2.5, 0.3))):
        w = ((torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * wamp)
        w[:, :, 1, 1] *= 3.0
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
        a = ((torch.rand(Cout, generator=g) - 0.3) * aamp).to(dev)
        b = ((torch.rand(Cout, generator=g) - 0.4) * 1.5).to(dev)
        spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < rate).float().to(dev)
        wd, bd = w.to(dev), bias.to(dev)
        ptc = ops.spikes_to_ptc(spikes)
        pk8 = ops.pack_conv_weight_i8(wd, bd, transposed)
        s32 = ops.ptc_to_s32(ptc)
        if Cin % 32 == 0:
            assert torch.equal(s32, ops.spikes_to_s32(spikes))
        got = ops.vae_fp6_fwd(s32, ops.vae_fp6_pack(wd, bd, transposed), Cout, bn_a=a, bn_b=b, transposed=transposed,
                              out_kind=kind, coef=coef if layer == "dec2" else None)
        if layer == "dec2":
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, collapse_coef=coef, **geo)
        else:
            want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)      # u8 [B,Ho,Wo,16,Cout]
            if layer == "dec1":
                got, want = ops.s32_to_spikes(got), ops.ptc_to_spikes(want)
        total += want.numel(); mism += int((want != got).sum())
        assert torch.equal(want, got), (layer, trial, int((want != got).sum()))
        assert 0.0 < float((want > 0).float().mean()) < 1.0
    torch.cuda.synchronize()
    cap = 1 << 20
    assert all(int(v[0]) == 0 and int(v[2 + cap:].abs().sum()) == 0 for k, v in ops._FLAG_DEFAULT.items() if k[0] == "vae"), \
        "live counter, overflow bitmap and hand-over ticket come back clean"
    if layer == "dec1":          # the int8 kernel's own S32 output form (used when the fp6 kernel has no instance for a shape)
        sp1 = (torch.rand(16, B, 16, hw, hw, generator=g) < 0.3).float().to(dev)
        w1 = ((torch.rand(16, 64, 3, 3, generator=g) - 0.5) * 0.5).to(dev)
        a1 = (torch.rand(64, generator=g) * 2 + 0.5).to(dev); b1 = (torch.rand(64, generator=g) - 0.5).to(dev)
        pk1 = ops.pack_conv_weight_i8(w1, None, True)
        u8 = ops.conv_mfma_fused(ops.spikes_to_ptc(sp1), pk1, 64, mode=ops.MODE_LIF, bn_a=a1, bn_b=b1, **geo)
        s32o = ops.conv_mfma_fused(ops.spikes_to_ptc(sp1), pk1, 64, mode=ops.MODE_LIF, bn_a=a1, bn_b=b1, out_s32=True, **geo)
        assert torch.equal(ops.s32_to_spikes(s32o), ops.ptc_to_spikes(u8))
    parity(f"vae_fp6_{layer}_vs_int8_B{B}_{hw}x{hw}_c{Cout}", values=total, mismatches=mism)


# ------------------------------------------------------------------------------------------------- round 6: the module-API read-out layer
@pytest.mark.parametrize("B,H,Cin,Cout,transposed,T", [(16, 28, 32, 1, True, 16), (3, 32, 32, 3, True, 16), (5, 9, 16, 4, False, 16),
                                                       (2, 11, 48, 1, False, 4)])
def test_raw_output_of_few_channel_spike_layers_step_parallel_kernel(dev, ops, B, H, Cin, Cout, transposed, T):
    """`pred = model.decoder(quantized)` (R/main.py:397) returns the read-out layer's per-step convolution [T,B,C,28,28]; memout and tanh
    are main.py's own calls.  For few output channels that layer runs conv_raw_steps_kernel (one thread per position AND step) instead of the
    generic one-thread-per-output kernel: its values must equal the generic kernel's on the same spikes given as fp32 [T,B,C,H,W] (same fp64
    sums in the same order) bit for bit, and the fp64 convolution to fp32 round-off."""
    import torch.nn.functional as F
    from spkdiff.ops import IN_PTC, IN_SEQ, MODE_RAW
    g = torch.Generator().manual_seed(90 + B + Cout)
    spikes = (torch.rand(T, B, Cin, H, H, generator=g) < 0.15).float()
    w = (torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * 0.4
    bias = torch.rand(Cout, generator=g) - 0.5
    wp = ops.pack_conv_weight(w.to(dev), transposed)
    geo = dict(k=3, stride=1, pad=1, transposed=transposed)
    sd = spikes.to(dev)
    fast = ops.conv_fused(ops.spikes_to_ptc(sd), wp, bias.to(dev), in_kind=IN_PTC, T=T, mode=MODE_RAW, want_f32=True, **geo)["f32"]
    gen = ops.conv_fused(sd, wp, bias.to(dev), in_kind=IN_SEQ, T=T, mode=MODE_RAW, want_f32=True, **geo)["f32"]
    conv = (lambda x: F.conv_transpose2d(x, w.double(), bias.double(), stride=1, padding=1)) if transposed else \
           (lambda x: F.conv2d(x, w.double(), bias.double(), stride=1, padding=1))
    want = torch.stack([conv(spikes[t].double()) for t in range(T)])
    err = float((fast.cpu().double() - want).abs().max())
    parity(f"raw_steps_kernel_B{B}_{H}x{H}_c{Cout}", values=int(fast.numel()), mismatches_vs_generic=int((fast != gen).sum()), max_abs_err_vs_fp64=err)
    assert fast.shape == (T, B, Cout, H, H) and torch.equal(fast, gen)
    assert err <= 2e-6 * (1 + float(want.abs().max()))


# ------------------------------------------------------------------------------------------------- round 6: small batches
@pytest.mark.parametrize("B", [1, 5, 16, 31])
def test_fp6v2_small_batch_split_bit_equal_to_whole_image_items(dev, ops, B):
    """R/main.py's own call shape is n_samples = 16 (R/snn_model/vq_diffusion.py:51): B x Cout / 32 whole-image items leave half the
    CUs idle in conv2 / conv3 / conv5.  The automatic form (spk_den_conv3x3_mfma_fp6v2 form 0) then runs two half-image items per
    image on four-wave workgroups; spikes AND spike counts must equal the whole-image form (form 1) and the six-plane exact kernel
    bit for bit -- full batches and the sampler's active-set calls (device-side image count), repeated launches on one workspace."""
    g = torch.Generator().manual_seed(7100 + B)
    total = mism = 0
    for Cout, Cin in ((128, 64), (256, 128), (512, 256), (256, 512)):
        w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05)
        w[:, :, 1, 1] *= 3.0
        bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
        a = ((torch.rand(Cout, generator=g) - 0.3) * 12.0).to(dev)
        b = ((torch.rand(Cout, generator=g) - 0.4) * 1.5).to(dev)
        sd = (torch.rand(16, B, Cin, 7, 7, generator=g) < 0.08).float().to(dev)
        pk, xs = ops.den_pack_weight_fp6v2(w.to(dev), bias.to(dev)), ops.spikes_to_s32(sd)
        outs = {}
        try:
            for form in (1, 0):
                ops.FP6V2_FORM = form
                for rep in range(2):
                    o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                outs[form] = (o.clone(), c.clone())
                if B >= 5:
                    n = B // 2 + 1
                    active = torch.arange(B, dtype=torch.int32, device=dev)
                    n_act = torch.tensor([n, 0], dtype=torch.int32, device=dev)
                    with ops.active_set(active, n_act):
                        o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                    outs[(form, "active")] = (o[:n].clone(), c[:n].clone())
        finally:
            ops.FP6V2_FORM = 0
        bad = int((outs[0][0] != outs[1][0]).sum()) + int((outs[0][1] != outs[1][1]).sum())
        if B >= 5:
            bad += int((outs[(0, "active")][0] != outs[(1, "active")][0]).sum()) + int((outs[(0, "active")][1] != outs[(1, "active")][1]).sum())
            assert torch.equal(outs[(0, "active")][0], outs[0][0][:B // 2 + 1]), "the first n images of the full batch"
        o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(w.to(dev), bias.to(dev)), Cout, bn_a=a, bn_b=b,
                                          want_counts=True)
        bad += int((ops.c4_to_spikes(o1) != ops.s32_to_spikes(outs[0][0])).sum()) + int((c1 != outs[0][1]).sum())
        total += outs[0][0].numel() * 2; mism += bad
        assert bad == 0, (Cout, Cin, bad)
        assert 0.001 < float(ops.s32_to_spikes(outs[0][0]).mean()) < 0.9
    torch.cuda.synchronize()
    assert all(flag_ws_clean(v) for k, v in ops._FLAG_DEFAULT.items() if k[0] == "den")
    parity(f"fp6v2_small_batch_split_B{B}", neuron_steps=total, spike_mismatches=mism)


# ------------------------------------------------------------------------------------------------- round 6: the overflow path of the flag list
def _den_flagged(ops, B, Cout, H, W):
    """flag_words[1] of the workspace the last eager fp6v2 call of this shape used = neurons that call flagged."""
    from spkdiff import _lib
    words = int(_lib.lib.spk_den_fp6v2_flag_words(B, Cout, H, W))
    v = [v for k, v in ops._FLAG_DEFAULT.items() if k[0] == "den" and k[3] == words]
    assert v
    return int(v[-1][1]), v[-1]


@pytest.mark.parametrize("B,hw,form", [(64, 7, "full"), (5, 7, "full"), (40, 7, "active"), (96, 7, "listed"), (33, 8, "full"), (12, 8, "active")])
def test_flag_overflow_path_fp6v2_small_capacity(dev, ops, B, hw, form):
    """VERDICT r5 'What's weak' 1: the certified kernels list flagged neurons in an id list and, beyond its capacity, in an overflow
    bitmap the tail launch scans (csrc/den_mfma_fp6v2.hip fp6v2_fixup_body) -- a branch no earlier test executed.  The capacity is a
    per-call argument now: with 64 entries (list AND bitmap in one launch) and with 0 (bitmap only) every layer shape -- merged tail
    launch of full 7x7 batches, the sampler's active-set tail, the listed-positions launch, the repair-only tail of 8x8 latents --
    must give the spikes and spike counts of the default capacity AND of the six-plane exact kernel bit for bit, and leave the
    workspace clean (live counter, bitmap, ticket zero).  BN terms include large and negative scales so that 1e2..1e4 neurons are
    flagged per launch.  R/snn_model/vq_diffusion.py:166-184, SJ/activation_based/neuron.py:799-811."""
    g = torch.Generator().manual_seed(4100 + B + hw)
    H = W = hw
    total = mism = 0
    flagged = {}
    act = need = None
    if form != "full":
        unmasked = torch.rand(B, 1, hw, hw, generator=g) < 0.5
        u = torch.rand(B, 1, hw, hw, generator=g) * (2.0 / 30)
        act = ops.select_active(unmasked.to(dev), 30, u.to(dev))
        if form == "listed":
            need = ops.select_needed(unmasked.to(dev), 30, act, ops.NeedLists(B, 4, dev), u.to(dev))
        n_act = int(act[1][0].item())
        assert 0 < n_act <= B
    try:
        for (Cout, Cin), radius in (((128, 64), 4), ((256, 128), 3), ((512, 256), 2), ((256, 512), 1)):
            w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05)
            w[:, :, 1, 1] *= 3.0
            bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
            a = ((torch.rand(Cout, generator=g) - 0.3) * 12.0).to(dev)
            b = ((torch.rand(Cout, generator=g) - 0.4) * 1.5).to(dev)
            sd = (torch.rand(16, B, Cin, H, W, generator=g) < 0.08).float().to(dev)
            pk, xs = ops.den_pack_weight_fp6v2(w.to(dev), bias.to(dev)), ops.spikes_to_s32(sd)

            def run():
                if form == "full":
                    return ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                with ops.active_set(*act, need=need):
                    return ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True,
                                                      need_radius=radius if form == "listed" else None)
            outs = {}
            for cap in (-1, 64, 0):
                ops.FLAG_CAP = cap
                for rep in range(2):                                      # (twice on one workspace: it must come back re-armed)
                    o, c = run()
                    if form == "listed":                                  # unlisted positions are left as they were: start from the same
                        pass
                outs[cap] = (o.clone(), c.clone())
                nfl, ws = _den_flagged(ops, B, Cout, H, W)
                torch.cuda.synchronize()
                assert flag_ws_clean(ws), (Cout, Cin, cap, "workspace not clean")
                flagged[(Cout, cap)] = nfl
            n_img = B if form == "full" else n_act
            if form == "listed":
                rec = need.records(radius).cpu().numpy()
                listed = torch.zeros(B, 49, dtype=torch.bool)
                for si in range(n_act):
                    listed[si, rec[si, :rec[si, 48]].tolist()] = True
                    listed[si, 48] = True
                m = listed[:, None, :, None, None].to(dev)
                mc = listed[:, None, :, None].to(dev)
                view = lambda o: o.view(B, Cout // 32, 49, 16, 16)
                for cap in (64, 0):
                    bad = int(((view(outs[cap][0]) != view(outs[-1][0])) & m).sum()) + \
                        int(((outs[cap][1].view(B, Cout // 32, 49, 32) != outs[-1][1].view(B, Cout // 32, 49, 32)) & mc).sum())
                    mism += bad
                    assert bad == 0, (Cout, Cin, cap, bad)
            else:
                for cap in (64, 0):
                    bad = int((outs[cap][0][:n_img] != outs[-1][0][:n_img]).sum()) + int((outs[cap][1][:n_img] != outs[-1][1][:n_img]).sum())
                    mism += bad
                    assert bad == 0, (Cout, Cin, cap, bad)
                # and the exact six-plane kernel
                ops.FLAG_CAP = -1
                o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(w.to(dev), bias.to(dev)), Cout, bn_a=a,
                                                  bn_b=b, want_counts=True)
                s1 = ops.c4_to_spikes(o1)[:, :n_img]
                for cap in (64, 0):
                    s2 = ops.s32_to_spikes(outs[cap][0])[:, :n_img]
                    bad = int((s1 != s2).sum())
                    mism += bad
                    assert bad == 0 and torch.equal(c1[:n_img], outs[cap][1][:n_img]), (Cout, Cin, cap, bad)
            assert flagged[(Cout, 64)] == flagged[(Cout, -1)] == flagged[(Cout, 0)], flagged
            total += outs[-1][0][:n_img].numel() * 2
    finally:
        ops.FLAG_CAP = -1
    parity(f"flag_overflow_fp6v2_{form}_B{B}_{hw}x{hw}", neuron_steps=total, spike_mismatches=mism,
           flagged_per_layer={str(k[0]): v for k, v in flagged.items() if k[1] == -1})
    assert min(v for k, v in flagged.items()) > 0
    if B >= 33:
        assert max(v for k, v in flagged.items()) > 64, ("the id list of 64 must overflow for this test to mean anything", flagged)


@pytest.mark.parametrize("layer,B,hw", [("dec2", 9, 14), ("dec2", 3, 16), ("dec1", 21, 7), ("dec1", 3, 8), ("enc2", 19, 14), ("enc2", 3, 16)])
def test_flag_overflow_path_vae_fp6_small_capacity(dev, ops, layer, B, hw):
    """The same branch of spk_vae_fp6_fwd (csrc/vae_fp6.hip vae_fp6_fixup_kernel): id-list capacities 8 and 0 against the default
    capacity and against the int8 gather kernel (exact by construction), all three output kinds; workspace clean afterwards.
    R/snn_model/vae_model.py:101-159."""
    from spkdiff import _lib
    g = torch.Generator().manual_seed(5200 + B + hw + len(layer))
    coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
    transposed = layer != "enc2"
    Cin, Cout = {"dec2": (64, 32), "dec1": (16, 64), "enc2": (32, 64)}[layer]
    kind = {"dec2": ops.VAE_OUT_COLLAPSED, "dec1": ops.VAE_OUT_S32, "enc2": ops.VAE_OUT_PTC}[layer]
    geo = dict(k=3, stride=2, pad=1, transposed=transposed, out_pad=1 if transposed else 0)
    Ho = 2 * hw if transposed else hw // 2
    small = 8 if B >= 9 else 1                     # id-list entries of the list-AND-bitmap case (a B = 3 launch flags 2 .. 30 neurons)
    total = mism = 0
    flagged = {}
    try:
        for trial, (wamp, aamp, rate) in enumerate(((0.08, 10.0, 0.08), (0.4, 2.5, 0.3))):
            w = ((torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * wamp)
            w[:, :, 1, 1] *= 3.0
            bias = (torch.rand(Cout, generator=g) - 0.5) * 0.2
            a = ((torch.rand(Cout, generator=g) - 0.3) * aamp).to(dev)
            b = ((torch.rand(Cout, generator=g) - 0.4) * 1.5).to(dev)
            spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < rate).float().to(dev)
            wd, bd = w.to(dev), bias.to(dev)
            ptc = ops.spikes_to_ptc(spikes)
            s32 = ops.ptc_to_s32(ptc)
            pk = ops.vae_fp6_pack(wd, bd, transposed)
            if layer == "dec2":
                want = ops.conv_mfma_fused(ptc, ops.pack_conv_weight_i8(wd, bd, transposed), Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b,
                                           collapse_coef=coef, **geo)
            else:
                want = ops.conv_mfma_fused(ptc, ops.pack_conv_weight_i8(wd, bd, transposed), Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)
                if layer == "dec1":
                    want = ops.ptc_to_spikes(want)
            words = int(_lib.lib.spk_vae_fp6_flag_words(B, Cout, Ho, Ho))
            for cap in (-1, small, 0):
                ops.FLAG_CAP = cap
                for rep in range(2):
                    got = ops.vae_fp6_fwd(s32, pk, Cout, bn_a=a, bn_b=b, transposed=transposed, out_kind=kind,
                                          coef=coef if layer == "dec2" else None)
                if layer == "dec1":
                    got = ops.s32_to_spikes(got)
                torch.cuda.synchronize()
                ws = [v for k, v in ops._FLAG_DEFAULT.items() if k[0] == "vae" and k[3] == words][-1]
                assert flag_ws_clean(ws), (layer, trial, cap, "workspace not clean")
                flagged[(trial, cap)] = int(ws[1])
                bad = int((want != got).sum())
                total += want.numel(); mism += bad
                assert bad == 0, (layer, trial, cap, bad)
            assert flagged[(trial, -1)] == flagged[(trial, small)] == flagged[(trial, 0)] > 0, flagged
    finally:
        ops.FLAG_CAP = -1
    assert max(flagged.values()) > small, ("the short id list must overflow for this test to mean anything", flagged)
    parity(f"flag_overflow_vae_fp6_{layer}_B{B}_{hw}x{hw}", values=total, mismatches=mism,
           flagged_per_trial={str(k[0]): v for k, v in flagged.items() if k[1] == -1})


@pytest.mark.parametrize("layer,B,hw", [("dec2", 64, 14), ("dec1", 128, 7), ("enc2", 512, 14)])
def test_flag_overflow_path_vae_fp6_real_capacity_adversarial_layer(dev, ops, layer, B, hw):
    """spk_vae_fp6_fwd with the REAL id-list capacity (2^20) overflowed: BatchNorm scale ~ 0 and shift 2, so that every membrane
    potential sits within a few ulp of the threshold (see the denoiser-layer test below) -- 1.6 M neurons per launch, nearly all
    flagged; against the int8 gather kernel (exact by construction), bit for bit; workspace clean.  R/snn_model/vae_model.py:101-159."""
    from spkdiff import _lib
    g = torch.Generator().manual_seed(6300 + B)
    coef = torch.pow(torch.tensor(0.8), torch.arange(15, -1, -1).float()).to(dev)
    transposed = layer != "enc2"
    Cin, Cout = {"dec2": (64, 32), "dec1": (16, 64), "enc2": (32, 64)}[layer]
    kind = {"dec2": ops.VAE_OUT_COLLAPSED, "dec1": ops.VAE_OUT_S32, "enc2": ops.VAE_OUT_PTC}[layer]
    geo = dict(k=3, stride=2, pad=1, transposed=transposed, out_pad=1 if transposed else 0)
    Ho = 2 * hw if transposed else hw // 2
    w = ((torch.rand((Cin, Cout, 3, 3) if transposed else (Cout, Cin, 3, 3), generator=g) - 0.5) * 0.2).to(dev)
    bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.2).to(dev)
    a = (2e-6 * (torch.rand(Cout, generator=g) + 0.5) * torch.sign(torch.rand(Cout, generator=g) - 0.3)).to(dev)
    b = torch.full((Cout,), 2.0).to(dev)
    spikes = (torch.rand(16, B, Cin, hw, hw, generator=g) < 0.1).float().to(dev)
    ptc = ops.spikes_to_ptc(spikes)
    s32 = ops.ptc_to_s32(ptc)
    for rep in range(2):
        got = ops.vae_fp6_fwd(s32, ops.vae_fp6_pack(w, bias, transposed), Cout, bn_a=a, bn_b=b, transposed=transposed, out_kind=kind,
                              coef=coef if layer == "dec2" else None)
    torch.cuda.synchronize()
    words = int(_lib.lib.spk_vae_fp6_flag_words(B, Cout, Ho, Ho))
    ws = [v for k, v in ops._FLAG_DEFAULT.items() if k[0] == "vae" and k[3] == words][-1]
    nfl = int(ws[1])
    pk8 = ops.pack_conv_weight_i8(w, bias, transposed)
    if layer == "dec2":
        want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, collapse_coef=coef, **geo)
    else:
        want = ops.conv_mfma_fused(ptc, pk8, Cout, mode=ops.MODE_LIF, bn_a=a, bn_b=b, **geo)
        if layer == "dec1":
            got, want = ops.s32_to_spikes(got), ops.ptc_to_spikes(want)
    bad = int((want != got).sum())
    parity(f"flag_overflow_vae_fp6_real_capacity_{layer}_B{B}", neurons=B * Cout * Ho * Ho, flagged=nfl, id_list_capacity=FLAG_LIST,
           mismatches=bad)
    assert nfl > FLAG_LIST, (nfl, "the layer must overflow the real id list")
    assert bad == 0, bad
    assert flag_ws_clean(ws), "workspace not clean after an overflowing launch"


@pytest.mark.parametrize("hw,gamma", [(7, 0.0), (7, 2e-6), (8, 2e-6)])
def test_flag_overflow_path_real_capacity_adversarial_layer(dev, ops, hw, gamma):
    """The REAL capacity (2^20 ids) overflowed by a degenerate layer: BatchNorm scale ~ 0 with the shift at twice the threshold, so
    that every neuron's membrane potential sits AT the threshold at every step (h = v + (x - v) / 2 with x = 2 + gamma * y: h = 1 +
    gamma * y / 2; R/snn_model/vq_diffusion.py:166-184 with such a checkpoint, SJ/activation_based/neuron.py:799-811).  B = 64 on the
    256 -> 512 layer = 1.6 M neurons (2.1 M on 8x8), nearly all flagged: > 2^20, so list and bitmap are both in use.  gamma = 0: every
    neuron fires at every step (also in the reference: x = 2 exactly); gamma = 2e-6: x = 2 +- a few ulp, so the SIGN and the last bits
    of the exact pre-activation decide each first spike, i.e. the result is the exact path's or it is wrong.  Against the six-plane exact kernel, bit for bit; workspace clean."""
    g = torch.Generator().manual_seed(77)
    B, Cout, Cin, H, W = 64, 512, 256, hw, hw
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05).to(dev)
    bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.2).to(dev)
    a = (torch.full((Cout,), gamma) * (torch.rand(Cout, generator=g) + 0.5) * torch.sign(torch.rand(Cout, generator=g) - 0.3)).to(dev)
    b = torch.full((Cout,), 2.0).to(dev)
    sd = (torch.rand(16, B, Cin, H, W, generator=g) < 0.05).float().to(dev)
    pk, xs = ops.den_pack_weight_fp6v2(w, bias), ops.spikes_to_s32(sd)
    for rep in range(2):
        o2, c2 = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
    nfl, ws = _den_flagged(ops, B, Cout, H, W)
    torch.cuda.synchronize()
    o1, c1 = ops.den_conv3x3_mfma_fp6(ops.spikes_to_c4(sd), ops.den_pack_weight_fp6(w, bias), Cout, bn_a=a, bn_b=b, want_counts=True)
    s1, s2 = ops.c4_to_spikes(o1), ops.s32_to_spikes(o2)
    bad = int((s1 != s2).sum())
    parity(f"flag_overflow_real_capacity_{hw}x{hw}_gamma{gamma:g}", neurons=B * Cout * H * W, flagged=nfl, id_list_capacity=FLAG_LIST,
           spike_mismatches=bad, firing_rate=float(s1.mean()))
    assert nfl > FLAG_LIST, (nfl, "the layer must overflow the real id list")
    assert bad == 0 and torch.equal(c1, c2), bad
    assert flag_ws_clean(ws), "workspace not clean after an overflowing launch"
    if gamma == 0.0:
        assert float(s1.mean()) == 1.0
    else:
        assert 0.3 < float(s1.mean()) < 0.99         # a negative gamma * y delays the first spike by a step


def test_graphed_training_step_trains(dev):
    """spkdiff.train.GraphedTrainStep: one captured iteration of the reference's diffusion training loop (train_iter, backward,
    AdamW, reset_net) replayed per batch -- finite decreasing-on-average loss, parameters move, fresh noise per replay."""
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional
    from spkdiff.train import GraphedTrainStep
    den = DummyModel(1, 128, n_steps=16).cuda(0)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=128)
    opt = torch.optim.AdamW(den.parameters(), lr=1e-3, weight_decay=0.001, capturable=True)
    x0 = torch.randint(0, 128, (8, 1, 7, 7), generator=torch.Generator().manual_seed(3)).float().to(dev)
    w_before = den.conv4[0].weight.detach().clone()
    step = GraphedTrainStep(ab, opt, x0)
    losses = [float(step(x0).detach()) for _ in range(12)]
    assert all(l == l and abs(l) < 1e4 for l in losses)
    assert len(set(losses)) > 6, "every replay draws its own t and mask"
    assert float((den.conv4[0].weight.detach() - w_before).abs().max()) > 0
    with pytest.raises(RuntimeError):
        GraphedTrainStep(ab, torch.optim.AdamW(den.parameters(), lr=1e-3), x0)



def test_full_size_properties_of_the_benchmark_configs(dev, ops):
    """BASELINE.json's full sizes through properties that do not need the oracle: (configs[1]) the 100-step sample at
    B = 256 gives the SAME tokens with every image evaluated at every step, with the untouched images eliminated, and
    with the position lists on top (graph replay, Philox noise); (configs[2]) encode -> decode at B = 1024 through the fp6
    kernel family gives the code indices of the fp64 direct kernels exactly and their images to fp32 round-off."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    from spkdiff.ops import IN_PTC, IN_TINV
    den, _ = build_den(synth.MNIST, dev)
    toks = {}
    for mode in ("dense", "images", "lists"):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples = 256
        ab.skip_untouched, ab.list_positions = mode != "dense", mode == "lists"
        torch.manual_seed(2024)
        toks[mode] = ab.sample(temp=1.0, sample_steps=100).cpu()
        assert int(toks[mode].max()) < 128
    assert torch.equal(toks["dense"], toks["images"]) and torch.equal(toks["dense"], toks["lists"])
    model, _ = build_vae(synth.MNIST, dev)
    B = 1024
    img = (torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(42)) - 0.5).to(dev)
    idx = model.encode_images(img, 16)
    z_d = model.encoder.snn_convs.run(img, IN_TINV, final='ptc', T=16, stateful=False, impl='direct')['ptc']
    idx_d, _ = model.vq_layer._quantize_ptc(z_d)
    assert torch.equal(idx.reshape(-1), idx_d.reshape(-1))
    pred, u8 = model.decode_tokens(idx, 16)
    zq = ops.embedding(idx, model.vq_layer.embeddings.weight, nchw_hw=(7, 7))
    e = model.vq_layer.poisson.run(zq, IN_TINV, final='ptc', T=16, stateful=False)['ptc']
    rd = model.decoder.snn_convs.run(e, IN_PTC, final='memout', coef=model.memout.coef.flatten(), apply_tanh=True, want_u8=True,
                                     stateful=False, impl='direct')
    err = float((pred - rd['f32']).abs().max())
    assert err <= 5e-6, err                 # the collapsed read-out sums in another order: fp32 round-off over 800 k pixels
    d8 = (u8.int() - rd['u8'].int()).abs()
    assert int(d8.max()) <= 1 and float((d8 > 0).float().mean()) < 1e-4          # truncation edges only
    parity("full_size_properties", sample_B256_100_steps_modes_equal=True, encdec_B1024_index_mismatches=0,
           encdec_B1024_pred_max_abs_diff_vs_direct=err, encdec_B1024_u8_off_by_one_frac=float((d8 > 0).float().mean()))



# ------------------------------------------------------------------------------------------------- F8 LIF training
@pytest.mark.parametrize("det", [False, True])
def test_f8_lif_training_bptt_vs_reference_fixture(golden_dir, dev, det):
    """LIFNode in train mode -- HIP forward that keeps h + HIP surrogate-gradient backward behind a
    torch.autograd.Function -- against the reference's torch-backend autograd (fixture F8): two calls without reset
    (state in the graph), spikes exact, final v and dL/dx to fp32 round-off (the reference sums the same terms in
    autograd's order)."""
    from spikingjelly.activation_based import neuron, surrogate
    d = load(golden_dir, f"f8_lif_train_{'detach' if det else 'nodetach'}.npz")
    x = torch.from_numpy(d["x_seq"]).to(dev).requires_grad_(True)
    w1, w2, w3 = (torch.from_numpy(d[k]).to(dev) for k in ("w1", "w2", "w3"))
    node = neuron.LIFNode(surrogate_function=surrogate.ATan(), detach_reset=det, step_mode='m').train()
    sa = node(x); sb = node(x.flip(0))
    ((sa * w1).sum() + (sb * w2).sum() + (node.v * w3).sum()).backward()
    assert torch.equal(sa.detach().cpu(), unpack(d["spikes_a"], d["spikes_shape"]))
    assert torch.equal(sb.detach().cpu(), unpack(d["spikes_b"], d["spikes_shape"]))
    want_v, want_g = torch.from_numpy(d["v"]), torch.from_numpy(d["grad_x"])
    assert float((node.v.detach().cpu() - want_v).abs().max()) <= 1e-6
    err = (x.grad.cpu() - want_g).abs()
    assert float((err / (1e-6 + 1e-5 * want_g.abs())).max()) <= 1.0, float(err.max())
    node.reset()
    assert node.v == 0.0 and isinstance(node.v, float)
    node.eval()                                  # and the inference kernel is untouched by the training path
    with torch.no_grad():
        assert torch.equal(node(x.detach()).cpu(), unpack(d["spikes_a"], d["spikes_shape"]))


@pytest.mark.parametrize("N,tau,vr", [(1000, 2.0, 0.0), (1001, 3.0, -0.25)])
def test_lif_train_kernels_vs_live_oracle(dev, ops, N, tau, vr):
    """spk_lif_train_fwd / spk_lif_train_bwd through the C-ABI on vector (N % 4 == 0) and scalar paths, tau not a power
    of two, v_reset != 0, non-zero initial state and an incoming gradient on the final state, against the oracle's
    autograd."""
    g = torch.Generator().manual_seed(7 + N)
    xs = torch.randn(9, N, generator=g) * 1.5; v0 = torch.rand(N, generator=g) - 0.5
    gs = torch.randn(9, N, generator=g); gv = torch.randn(N, generator=g)
    xo = xs.clone().requires_grad_(True); vo = v0.clone().requires_grad_(True)
    so, vlast = ref.lif_multi_step_train(xo, vo, v_reset=vr, tau=tau, alpha=2.0)
    ((so * gs).sum() + (vlast * gv).sum()).backward()
    s, h, vl = ops.lif_train_fwd(xs.to(dev), v0.to(dev), tau, 1.0, vr)
    gx, gv0 = ops.lif_train_bwd(gs.to(dev), gv.to(dev), h, tau, 1.0, vr, 2.0, False)
    assert torch.equal(s.cpu(), so.detach())
    assert float((vl.cpu() - vlast.detach()).abs().max()) <= 1e-6
    for got, want in ((gx.cpu(), xo.grad), (gv0.cpu(), vo.grad)):
        err = (got - want).abs()
        assert float((err / (1e-6 + 1e-5 * want.abs())).max()) <= 1.0, float(err.max())


# ------------------------------------------------------------------------------------------------- untouched-image elimination
def test_select_active_matches_the_change_test(dev, ops):
    """spk_select_active against the `changes` expression of R/snn_model/vq_diffusion.py:113-124 evaluated with torch on
    injected uniforms: ascending list of the images with at least one change, and its length; B > 256 walks in chunks."""
    g = torch.Generator().manual_seed(3)
    for B, t in ((300, 7), (16, 100), (5, 1)):
        u = torch.rand(B, 1, 7, 7, generator=g)
        unmasked = torch.rand(B, 1, 7, 7, generator=g) < 0.6
        changes = (u < 1.0 / t) & ~unmasked
        want = torch.nonzero(changes.flatten(1).any(1)).flatten().int()
        act, n = ops.select_active(unmasked.to(dev), t, u.to(dev))
        assert int(n[0]) == want.numel() and int(n[1]) == 0
        assert torch.equal(act[:int(n[0])].cpu(), want)


@pytest.mark.parametrize("cfgname,impl", [("cifar", "auto"), ("cifar", "i8"), ("mnist", "i8"), ("mnist", "direct")])
def test_skipping_untouched_images_other_kernel_families(dev, cfgname, impl):
    """The same elimination on the fp6 row-band form (8x8 latents), the int8 MFMA family (8x8 and 7x7 on request) and the
    fp64 direct kernels."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    cfg = synth.CIFAR if cfgname == "cifar" else synth.MNIST
    den, _ = build_den(cfg, dev)
    den.conv_impl_request = impl
    out = []
    for skip in (False, True):
        ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(cfg.latent, cfg.latent))
        ab.n_samples = 8 if impl == "direct" else 20
        ab.skip_untouched = skip
        torch.manual_seed(77)
        out.append(ab.sample(temp=1.0, sample_steps=20 if impl == "direct" else 64).cpu())
    assert torch.equal(out[0], out[1]) and int(out[0].max()) < 128


@pytest.mark.parametrize("steps", [100, 49])
def test_sampler_skipping_untouched_images_gives_the_same_tokens(dev, steps):
    """AbsorbingDiffusion.skip_untouched: the denoiser is evaluated only for the images a step touches; tokens are
    identical to the dense loop -- graph-replayed Philox mode (two consecutive calls), eager Philox mode and host-noise
    mode (the reference's CPU RNG order) -- because only `changes` positions ever read the logits (:140)."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, _ = build_den(synth.MNIST, dev)
    out = {}
    for skip in (False, True, 'lists'):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples = 24
        ab.skip_untouched = bool(skip)
        ab.list_positions = skip == 'lists'          # ... and, of the touched images, only the positions the step reads
        torch.manual_seed(1234)
        a = ab.sample(temp=0.9, sample_steps=steps).cpu()
        b = ab.sample(temp=0.9, sample_steps=steps).cpu()          # second replay: next Philox base
        ab.use_graph = False
        torch.manual_seed(1234)
        c = ab.sample(temp=0.9, sample_steps=steps).cpu()          # eager, same key as `a` (re-seeded)
        ab.noise_source = 'host'
        torch.manual_seed(99)
        d = ab.sample(temp=0.9, sample_steps=min(steps, 12)).cpu()
        out[skip] = (a, b, c, d)
        assert torch.equal(a, c) and not torch.equal(a, b)
        assert int(a.max()) < 128 and int(b.max()) < 128
    for x, y, z in zip(out[False], out[True], out['lists']):
        assert torch.equal(x, y) and torch.equal(x, z)


# ------------------------------------------------------------------------------------------------- F9 training step
def _rel_l2(got, want):
    return float((got - want).norm() / (want.norm() + 1e-30))


@pytest.mark.parametrize("shape,det,with_v", [((16, 3, 8, 5, 5), False, False), ((9, 2, 5, 7, 7), True, True),
                                               ((16, 40, 3, 7, 7), False, True)])
def test_bn_lif_train_operator_vs_oracle(dev, ops, shape, det, with_v):
    """spk_bn_lif_train_fwd / _bwd (training BatchNorm + surrogate-gradient LIF as one operator, the backward recomputing
    the membrane potentials from y) against the oracle's F.batch_norm(training=True) + lif_multi_step_train autograd:
    spikes equal outside the fragile set |h - 1| < 1e-5, state / running statistics / gradients to fp32 round-off."""
    import torch.nn.functional as F
    T, B, C, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = torch.randn(shape, generator=g) * 2 + 0.3
    gamma = 1 + 0.3 * torch.randn(C, generator=g); beta = 0.5 * torch.randn(C, generator=g)
    rm = torch.randn(C, generator=g); rv = torch.rand(C, generator=g) + 0.5
    v0 = (torch.rand(B, C, H, W, generator=g) - 0.5) if with_v else None
    gs = torch.randn(shape, generator=g); gv = torch.randn(B, C, H, W, generator=g)
    # oracle
    yo, go, bo = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    vo = v0.clone().requires_grad_(True) if with_v else None
    rmo, rvo = rm.clone(), rv.clone()
    z = F.batch_norm(yo.flatten(0, 1), rmo, rvo, go, bo, True, 0.1, 1e-5).view(shape)
    so, vlast = ref.lif_multi_step_train(z, 0.0 if vo is None else vo, detach_reset=det)
    ((so * gs).sum() + (vlast * gv).sum()).backward()
    hs = []
    with torch.no_grad():
        v = torch.zeros(B, C, H, W) if v0 is None else v0.clone()
        for t in range(T):
            h = v + (z[t] - v) / 2.0
            hs.append(h); v = torch.where(h >= 1.0, torch.zeros_like(h), h)
    fragile = (torch.stack(hs) - 1.0).abs() < 1e-5
    # HIP
    yd, gd, bd = (a.clone().to(dev).requires_grad_(True) for a in (y, gamma, beta))
    vd = v0.clone().to(dev).requires_grad_(True) if with_v else None
    rmd, rvd = rm.to(dev), rv.to(dev)
    s, vl = ops.BNLIFTrainFunction.apply(yd, gd, bd, vd, rmd, rvd, 0.1, 1e-5, 2.0, 1.0, 0.0, 2.0, det)
    ((s * gs.to(dev)).sum() + (vl * gv.to(dev)).sum()).backward()
    assert not bool(fragile.any()) or int(fragile.sum()) < 10
    assert torch.equal(s.detach().cpu()[~fragile], so.detach()[~fragile])
    assert float((rmd.cpu() - rmo).abs().max()) <= 1e-6 and float((rvd.cpu() - rvo).abs().max()) <= 1e-5
    if not bool(fragile.any()):
        assert float((vl.detach().cpu() - vlast.detach()).abs().max()) <= 1e-5
        pairs = [(yd.grad, yo.grad), (gd.grad, go.grad), (bd.grad, bo.grad)] + ([(vd.grad, vo.grad)] if with_v else [])
        for got, want in pairs:
            assert _rel_l2(got.cpu(), want) <= 2e-5, _rel_l2(got.cpu(), want)


@pytest.mark.parametrize("B,cin,cout,hw", [(3, 64, 128, (7, 7)), (5, 128, 64, (6, 6)), (2, 256, 64, (7, 5))])
def test_spike_conv_train_forward_exact_and_library_backward(dev, ops, B, cin, cout, hw):
    """ops.SpikeConvTrainFunction (spk_spikes_nhwc_to_fp4 + spk_den_pack_weight_fp6 + spk_den_conv3x3_fp6_raw): the
    training forward of a spike-input 3x3 convolution is the correctly rounded fixed-point-exact dot product (fp64 oracle
    to one fp32 rounding), written channels-last; weight / bias / input gradients come from the library and match the oracle's autograd."""
    g = torch.Generator().manual_seed(B * cin + cout)
    s = (torch.rand(16, B, cin, *hw, generator=g) < 0.07).float()
    w = (torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1
    bias = torch.randn(cout, generator=g) * 0.1
    gy = torch.randn(16, B, cout, *hw, generator=g)
    so, wo, bo = s.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = ref.seq_conv2d(so, wo, bo, 1, 1)
    (yo * gy).sum().backward()
    exact = ref.seq_conv2d(s.double(), w.double(), bias.double(), 1, 1).float()
    sd_, wd, bd = (a.clone().to(dev).requires_grad_(True) for a in (s, w, bias))
    y = ops.SpikeConvTrainFunction.apply(sd_, wd, bd)
    assert y.permute(0, 1, 3, 4, 2).is_contiguous()
    (y * gy.to(dev)).sum().backward()
    # six radix-32 digits = 29-bit fixed point per output channel: weights below 2^-6 of their channel's maximum (1.5 % of
    # these uniform ones) are rounded at 2^-29 of it, so a result may sit one fp32 rounding away from the fp64 value
    yc = y.detach().cpu()
    assert float((yc - exact).abs().max()) <= 2.5e-7 * (1 + float(exact.abs().max()))
    assert float((yc != exact).float().mean()) <= 0.05, "not (almost everywhere) the correctly rounded exact dot product"
    for got, want in ((sd_.grad, so.grad), (wd.grad, wo.grad), (bd.grad, bo.grad)):
        assert _rel_l2(got.cpu(), want) <= 1e-5, _rel_l2(got.cpu(), want)


@pytest.mark.parametrize("hw,cin", [((28, 28), 1), ((32, 32), 3), ((7, 7), 2), ((8, 8), 4)])
def test_exact_conv_train_backward_small_cin_any_map_size(dev, ops, hw, cin):
    """ops.ExactConvTrainFunction.backward sends 3x3 / s1 / p1 layers with <= 4 input channels and no input gradient
    to spk_conv3x3_wgrad_small, whose maps hold at most 64 positions (the denoiser's first layer).  A generic first layer on a
    28x28 or 32x32 input answers the workspace query with -1: it must take the framework's operator, not crash."""
    g = torch.Generator().manual_seed(hw[0] * 10 + cin)
    x = torch.randn(6, cin, *hw, generator=g)
    w = torch.randn(16, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(16, generator=g) * 0.1
    gy = torch.randn(6, 16, *hw, generator=g)
    wo, bo = w.double().requires_grad_(True), b.double().requires_grad_(True)
    (torch.nn.functional.conv2d(x.double(), wo, bo, 1, 1) * gy.double()).sum().backward()
    small = ops.lib.spk_conv3x3_wgrad_small_ws_bytes(6, hw[0], hw[1], 16, cin) > 0
    assert small == (hw[0] * hw[1] <= 64)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.ExactConvTrainFunction.apply(x.to(dev), wd, bd, 1, 1, False, 0)
    (y * gy.to(dev)).sum().backward()
    assert _rel_l2(wd.grad.cpu().double(), wo.grad) <= 1e-5 and _rel_l2(bd.grad.cpu().double(), bo.grad) <= 1e-5


@pytest.mark.parametrize("B,K,hw", [(4, 128, (7, 7)), (3, 128, (8, 8)), (2, 10, (3, 5))])
def test_masked_ce_vs_torch(dev, ops, B, K, hw):
    """spk_masked_ce: cross-entropy with ignore_index=-1 and its gradient, against torch's F.cross_entropy on CPU
    (the call of R/snn_model/vq_diffusion.py:85-88)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * K)
    logits = (torch.randn(B, K, *hw, generator=g) * 3).requires_grad_(True)
    tgt = torch.randint(0, K, (B, 1, *hw), generator=g).float()
    tgt[torch.rand(B, 1, *hw, generator=g) < 0.4] = -1
    coef = torch.rand(B, generator=g) + 0.1
    HW = hw[0] * hw[1]
    ce = F.cross_entropy(logits.reshape(B, K, HW), tgt.reshape(B, HW).long(), ignore_index=-1, reduction='none')
    want = (ce.sum(1) * coef).sum()
    want.backward()
    ld = logits.detach().clone().to(dev).requires_grad_(True)
    got = ops.MaskedCEFunction.apply(ld, tgt.to(dev), coef.to(dev))
    (got * 1.5).backward()
    assert float((ops.masked_ce(ld.detach(), tgt.to(dev)).cpu() - ce.detach()).abs().max()) <= 2e-5
    assert abs(float(got.detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach()))
    assert float((ld.grad.cpu() / 1.5 - logits.grad).abs().max()) <= 1e-6


def test_f9_diffusion_train_step_vs_reference_fixture(golden_dir, dev):
    """one training step of the absorbing diffusion -- DummyModel in train() mode (library
    convolutions, native fused BatchNorm+LIF with surrogate gradient), the reweighted-ELBO masked cross-entropy
    (spk_masked_ce) and loss.backward() -- teacher-forced with the fixture's (x_t, t, x_0_ignore) from the reference run
    under torch.manual_seed(909).  Floating point: the library convolutions round differently from the reference's CPU
    ones, so a few of the 3.8 M neuron-steps can land on the other side of the threshold (each such flip moves a 3x3 patch
    of logits of one sample by ~1.2e-3 = 0.35 % of the logit range, 8e-5 of it on average); tolerances sized for ~10 flips:
    loss 1e-3 relative, logits max 2e-2 / mean 1e-3 of their range, recorded gradients 2 % relative L2, every gradient norm
    2 %, running statistics 1e-5 relative.  Measured on MI355X: with no flip the loss is identical, logits agree to 6e-8
    and gradients to 1e-6; with one flip (the usual case with the NHWC kernels) loss 6e-6, logits 1.2e-3, gradients 7e-4."""
    from snn_model.vq_diffusion import AbsorbingDiffusion, functional
    d = load(golden_dir, "f9_train_step.npz")
    den, sd = build_den(synth.MNIST, dev)
    assert str(d["weights_crc"]) == synth.state_checksum(sd)
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=128)
    x_t, t = torch.from_numpy(d["x_t"]).to(dev), torch.from_numpy(d["t"]).to(dev)
    logits = den(x_t, t)
    loss = ab._loss_from_logits(logits, torch.from_numpy(d["x0_ignore"]).to(dev), t)
    loss.backward()
    want_logits = torch.from_numpy(d["logits"])
    lerr = (logits.detach().cpu() - want_logits).abs()
    print("F9 logits: max err", float(lerr.max()), "mean", float(lerr.mean()), "range", float(want_logits.abs().max()))
    assert float(lerr.max()) <= 2e-2 * float(want_logits.abs().max()), float(lerr.max())
    assert float(lerr.mean()) <= 1e-3 * float(want_logits.abs().max()), float(lerr.mean())
    assert abs(float(loss.detach()) - float(d["loss"])) <= 1e-3 * float(d["loss"])
    grads = {k: p.grad.cpu() for k, p in den.named_parameters()}
    print("F9 measured: loss rel err", abs(float(loss.detach()) - float(d["loss"])) / float(d["loss"]), "logits max err",
          float(lerr.max()), "mean", float(lerr.mean()), "range", float(want_logits.abs().max()), "grad rel L2",
          {k[5:]: round(_rel_l2(grads[k[5:]], torch.from_numpy(d[k])), 6) for k in d.files if k.startswith("grad.")})
    for k in d.files:
        if k.startswith("grad.") and float(np.linalg.norm(d[k])) > 1e-6:
            assert _rel_l2(grads[k[5:]], torch.from_numpy(d[k])) <= 2e-2, (k, _rel_l2(grads[k[5:]], torch.from_numpy(d[k])))
    for k, n in zip(d["grad_names"].tolist(), d["grad_norms"].tolist()):
        # a convolution bias in front of a batch-statistics BN has an exactly zero gradient: only round-off is left
        assert abs(float(grads[k].norm()) - n) <= 2e-2 * n + 1e-7, (k, float(grads[k].norm()), n)
    st = den.state_dict()
    for k in d.files:
        if k.startswith("stat."):
            want = torch.from_numpy(d[k])
            assert float(((st[k[5:]].cpu() - want).abs() / (1 + want.abs())).max()) <= 1e-5, k
    assert int(st["conv3.1.num_batches_tracked"]) == int(sd["conv3.1.num_batches_tracked"]) + 1
    functional.reset_net(den)
    assert den.conv1[2].v == 0.0


def test_diffusion_train_step_8x8_latent_vs_live_oracle(dev):
    """BASELINE config 4 shapes in training (8x8 latent, 64 diffusion steps; parity unpinned by the reference, whose
    AbsorbingDiffusion hard-codes 7x7): the HIP training graph -- library forward here, the fp6 forward does not hold an
    8x8 image -- against the oracle's restatement on the same (t, u): loss 1e-3 relative, gradient norms 2 %."""
    from snn_model.vq_diffusion import AbsorbingDiffusion, functional
    den, sd = build_den(synth.CIFAR, dev)
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(8, 8))
    g = torch.Generator().manual_seed(88)
    x0 = torch.randint(0, 128, (2, 1, 8, 8), generator=g).float()
    t = torch.tensor([9, 40])
    u = torch.rand(2, 1, 8, 8, generator=g)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    want, (_, x_t, x0i, mask, want_logits) = ref.train_loss(x0, sdo, 128, num_timesteps=64, t=t, u=u)
    want.backward()
    logits = den(x_t.to(dev), t.to(dev))
    loss = ab._loss_from_logits(logits, x0i.to(dev), t.to(dev))
    loss.backward()
    functional.reset_net(den)
    assert logits.shape == (2, 128, 8, 8) and ab.num_timesteps == 64
    assert abs(float(loss.detach()) - float(want.detach())) <= 1e-3 * float(want.detach())
    for k, p in den.named_parameters():
        n = float(sdo[k].grad.norm())
        assert abs(float(p.grad.norm()) - n) <= 2e-2 * n + 1e-7, (k, float(p.grad.norm()), n)


def test_train_iter_fused_vs_module_by_module_and_optimizer_step(dev):
    """The fused training graph (FusedSequential.train_forward) against the same model run module by module (library
    BatchNorm + the LIF-only HIP pair): same loss and gradients; then the reference's training loop body
    (R/main.py:243-252: train_iter, zero_grad, backward, AdamW step, reset_net) runs and changes the weights."""
    from snn_model.vq_diffusion import AbsorbingDiffusion, functional
    den, sd = build_den(synth.MNIST, dev)
    den.train()
    ab = AbsorbingDiffusion(den, mask_id=128)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randint(0, 128, (8, 1, 7, 7), generator=g).float().to(dev)
    t = torch.randint(1, 50, (8,), generator=g).to(dev)
    x_t, x0i, mask = ab.q_sample(x0, t)
    assert bool(((x_t == 128) == mask).all()) and bool(((x0i == -1) == ~mask).all())
    loss = ab._loss_from_logits(den(x_t, t), x0i, t); loss.backward()
    g_fused = {k: p.grad.clone() for k, p in den.named_parameters()}
    functional.reset_net(den); den.zero_grad(); den.load_state_dict(sd)
    inp = torch.cat((x_t, torch.ones_like(x_t) * t.view(-1, 1, 1, 1)), dim=1).unsqueeze(0).repeat(16, 1, 1, 1, 1)
    h = inp
    outs = []
    # (the stand-alone modules hand convolutions above ops.EXACT_TRAIN_FORWARD_MACS to the library operator, whose fp32 rounding --
    #  algorithm by algorithm -- can flip a spike the exact fused forward does not: the comparison runs them exactly)
    from spkdiff import ops as _ops
    keep, _ops.EXACT_TRAIN_FORWARD_MACS = _ops.EXACT_TRAIN_FORWARD_MACS, 1 << 62
    keep_n, _ops.NATIVE_TRAIN_FORWARD = _ops.NATIVE_TRAIN_FORWARD, False
    try:
        for blk in (den.conv1, den.conv2, den.conv3, den.conv4, den.conv5):
            for m in blk:
                h = m(h)
            outs.append(h)
        x6 = den.conv6[0](torch.cat((outs[4], outs[0]), dim=2))
    finally:
        _ops.EXACT_TRAIN_FORWARD_MACS, _ops.NATIVE_TRAIN_FORWARD = keep, keep_n
    loss2 = ab._loss_from_logits(x6.sum(0) / 16, x0i, t); loss2.backward()
    assert abs(float(loss.detach()) - float(loss2.detach())) <= 1e-3 * float(loss2.detach())
    for k, p in den.named_parameters():
        if float(p.grad.norm()) > 1e-6:          # (conv biases in front of a batch-statistics BN: zero gradient)
            assert _rel_l2(g_fused[k], p.grad) <= 2e-2, (k, _rel_l2(g_fused[k], p.grad))
        else:
            assert float(g_fused[k].norm()) <= 1e-6, k
    functional.reset_net(den); den.zero_grad(); den.load_state_dict(sd)
    opt = torch.optim.AdamW(den.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    losses = []
    for _ in range(3):
        l = ab.train_iter(x0)['loss']
        opt.zero_grad(); l.backward(); opt.step(); functional.reset_net(net=den)
        losses.append(float(l.detach()))
    assert all(np.isfinite(losses)) and not torch.equal(den.conv4[0].weight.detach().cpu(), sd["conv4.0.weight"])
    den.eval()                                   # and inference still runs on the fused kernels afterwards
    ab.n_samples = 4
    tok = ab.sample(temp=1.0, sample_steps=3)
    assert tok.shape == (4, 1, 7, 7) and int(tok.max()) < 128


@pytest.mark.parametrize("shape", [(16, 3, 5, 7, 7), (7, 1001)])
def test_psp_filter_and_adjoint_vs_oracle(dev, ops, shape):
    """spk_psp forward and adjoint (PSP of the VQ-VAE training losses, R/snn_model/snn_layers.py:12-26) against the
    oracle's loop under autograd; vector and scalar paths."""
    g = torch.Generator().manual_seed(len(shape))
    x = torch.randn(shape, generator=g); w = torch.randn(shape, generator=g)
    xo = x.clone().requires_grad_(True)
    so = ref.psp_filter(xo)
    (so * w).sum().backward()
    xd = x.clone().to(dev).requires_grad_(True)
    sdv = ops.PSPFunction.apply(xd, 2.0)
    (sdv * w.to(dev)).sum().backward()
    assert float((sdv.detach().cpu() - so.detach()).abs().max()) <= 1e-6
    assert float((xd.grad.cpu() - xo.grad).abs().max()) <= 1e-5 * float(xo.grad.abs().max())


@pytest.mark.parametrize("B,hw,D,K", [(4, 7, 16, 128), (32, 7, 16, 128), (3, 8, 16, 128), (2, 5, 8, 37)])
def test_vq_train_function_vs_op_by_op_autograd(dev, ops, B, hw, D, K):
    """ops.VQTrainFunction (spk_vq_train_readout / spk_vq_argmin / spk_vq_train_quant forward, spk_vq_train_bwd backward) against
    the same algebra written op by op through autograd -- the reference's lines, R/snn_model/vae_model.py:61-78: read-out, nearest
    code, q / e latent losses, straight-through estimator.  Same indices, same values; gradients of the spikes, of alpha and of
    the codebook (fixed-order sums instead of the framework's atomics) to fp32 round-off."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 100 + hw)
    T = 16
    x = (torch.rand(T, B, D, hw, hw, generator=g) < 0.3).float().to(dev)
    coef = torch.pow(torch.tensor(0.8), torch.arange(T - 1, -1, -1).float()).to(dev)
    E0 = (torch.randn(K, D, generator=g) * 0.7).to(dev)
    gq = torch.randn(B, D, hw, hw, generator=g).to(dev)
    res = {}
    for mode in ("fused", "autograd"):
        xs = x.clone().requires_grad_(True)
        alpha = torch.tensor(0.37, device=dev, requires_grad=True)
        E = E0.clone().requires_grad_(True)
        if mode == "fused":
            q, loss = ops.VQTrainFunction.apply(xs, coef, alpha, E, 0.25)
        else:
            xm = (1 - alpha) * torch.sum(xs * coef.view(T, 1, 1, 1, 1), dim=0) + alpha * torch.sum(xs, dim=0) / T
            xm = xm.permute(0, 2, 3, 1).contiguous()
            idx = ops.vq_argmin(xm.reshape(-1, D).detach(), E)
            qq = F.embedding(idx, E).view_as(xm)
            loss = F.mse_loss(qq, xm.detach()) + 0.25 * F.mse_loss(xm, qq.detach())
            q = (xm + (qq - xm).detach()).permute(0, 3, 1, 2).contiguous()
        ((q * gq).sum() + 3.0 * loss).backward()
        res[mode] = (q.detach(), loss.detach(), xs.grad, alpha.grad, E.grad)
    qa, la, gxa, gaa, gEa = res["autograd"]
    qf, lf, gxf, gaf, gEf = res["fused"]
    assert float((qf - qa).abs().max()) <= 1e-6 * (1 + float(qa.abs().max()))
    assert abs(float(lf) - float(la)) <= 2e-6 * abs(float(la))
    assert _rel_l2(gxf, gxa) <= 2e-6 and _rel_l2(gEf, gEa) <= 2e-6
    assert abs(float(gaf) - float(gaa)) <= 2e-5 * (abs(float(gaa)) + 1e-3)
    parity(f"vq_train_function_B{B}_{hw}x{hw}_D{D}_K{K}", loss_rel=abs(float(lf) - float(la)) / abs(float(la)),
           grad_x_rel_l2=_rel_l2(gxf, gxa), grad_codebook_rel_l2=_rel_l2(gEf, gEa),
           grad_alpha_rel=abs(float(gaf) - float(gaa)) / (abs(float(gaa)) + 1e-12))


@pytest.mark.parametrize("shape", [(16, 4, 16, 7, 7), (16, 32, 16, 7, 7), (9, 3, 5, 3, 3)])
def test_psp_loss_function_vs_op_by_op_autograd(dev, ops, shape):
    """ops.PSPLossFunction (spk_psp_loss_fwd / _bwd) against the reference's lines through autograd and the stand-alone filter
    (R/snn_model/vae_model.py:79-84): the same loss and the same gradients of both spike tensors to fp32 round-off."""
    g = torch.Generator().manual_seed(shape[1] * 7 + shape[0])
    q0 = (torch.rand(*shape, generator=g) < 0.2).float().to(dev)
    x0 = (torch.rand(*shape, generator=g) < 0.3).float().to(dev)
    res = {}
    for mode in ("fused", "autograd"):
        q, x = q0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        if mode == "fused":
            loss = ops.PSPLossFunction.apply(q, x, 0.25, 2.0)
        else:
            pq, px = ops.PSPFunction.apply(q, 2.0), ops.PSPFunction.apply(x, 2.0)
            loss = torch.mean((pq - px.detach()) ** 2) + 0.25 * torch.mean((pq.detach() - px) ** 2)
        (loss * 1.7).backward()
        res[mode] = (loss.detach(), q.grad, x.grad)
    la, gqa, gxa = res["autograd"]
    lf, gqf, gxf = res["fused"]
    assert abs(float(lf) - float(la)) <= 2e-6 * abs(float(la))
    assert _rel_l2(gqf, gqa) <= 2e-6 and _rel_l2(gxf, gxa) <= 2e-6
    parity(f"psp_loss_function_{'x'.join(map(str, shape))}", loss_rel=abs(float(lf) - float(la)) / abs(float(la)),
           grad_q_rel_l2=_rel_l2(gqf, gqa), grad_x_rel_l2=_rel_l2(gxf, gxa))


@pytest.mark.parametrize("shape", [(16, 4, 1, 28, 28), (16, 3, 3, 32, 32), (5, 2, 1, 6, 7)])
def test_recon_loss_function_vs_op_by_op_autograd(dev, ops, shape):
    """ops.ReconLossFunction (spk_recon_loss_fwd / _bwd) against mse_loss(tanh(memout(y)), image) through autograd
    (R/snn_model/vae_model.py:189-196)."""
    import torch.nn.functional as F
    T = shape[0]
    g = torch.Generator().manual_seed(T * 11 + shape[1])
    y0 = (torch.randn(*shape, generator=g) * 0.4).to(dev)
    img = (torch.rand(*shape[1:], generator=g) - 0.5).to(dev)
    coef = torch.pow(torch.tensor(0.8), torch.arange(T - 1, -1, -1).float()).to(dev)
    res = {}
    for mode in ("fused", "autograd"):
        y = y0.clone().requires_grad_(True)
        if mode == "fused":
            loss = ops.ReconLossFunction.apply(y, coef, img)
        else:
            loss = F.mse_loss(torch.tanh(torch.sum(y * coef.view(T, 1, 1, 1, 1), dim=0)), img)
        (loss * 2.5).backward()
        res[mode] = (loss.detach(), y.grad)
    assert abs(float(res["fused"][0]) - float(res["autograd"][0])) <= 3e-6 * abs(float(res["autograd"][0]))
    assert _rel_l2(res["fused"][1], res["autograd"][1]) <= 3e-6
    parity(f"recon_loss_function_{'x'.join(map(str, shape))}", grad_rel_l2=_rel_l2(res["fused"][1], res["autograd"][1]))


def test_f10_vqvae_train_step_vs_reference_fixture(golden_dir, dev):
    """SURVEY §8f item 2 (second half): SNN_VQVAE.forward in train() mode and (loss_eq + loss_rec).backward() as
    R/main.py:136-142 runs it -- library (transposed) convolutions, native BatchNorm+LIF block tails, membrane read-out,
    code search, PSP filter -- against the reference's run (fixture F10).  Same floating-point caveat as F9 (a neuron-step
    may flip, which here can also move a code index): losses 2 % relative, gradients 5 % relative L2 / norms 5 %,
    running statistics 1e-4 relative.  The measured values are printed."""
    from snn_model.vae_model import functional
    d = load(golden_dir, "f10_vqvae_train_step.npz")
    model, sd = build_vae(synth.MNIST, dev)
    assert str(d["weights_crc"]) == synth.state_checksum(sd)
    model.data_variance = torch.from_numpy(d["data_variance"]).to(dev)
    model.train()
    img = torch.from_numpy(d["images"]).to(dev)
    # (the parity run keeps every convolution on the exact forward kernel -- the correctly rounded dot product, box-independent;
    #  its backward is the native data / weight gradient of csrc/conv_train.hip.  The default path -- native fp32 matrix-core
    #  forward as well -- is compared with the same fixture further down)
    from spkdiff import ops as _ops
    keep, _ops.EXACT_TRAIN_FORWARD_MACS = _ops.EXACT_TRAIN_FORWARD_MACS, 1 << 62
    keep_n, _ops.NATIVE_TRAIN_FORWARD = _ops.NATIVE_TRAIN_FORWARD, False
    try:
        leq, lrec, lreal = model(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img)
    finally:
        _ops.EXACT_TRAIN_FORWARD_MACS, _ops.NATIVE_TRAIN_FORWARD = keep, keep_n
    (leq + lrec).backward()
    rel = {k: abs(float(v.detach()) - float(d[k])) / float(d[k]) for k, v in
           (("loss_eq", leq), ("loss_rec", lrec), ("real_loss_rec", lreal))}
    grads = {k: (p.grad.cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in model.named_parameters()}
    gerr = {k[5:]: round(_rel_l2(grads[k[5:]], torch.from_numpy(d[k])), 6) for k in d.files if k.startswith("grad.")}
    print("F10 measured: loss rel err", rel, "grad rel L2", gerr)
    assert max(rel.values()) <= 2e-2, rel
    for k in d.files:
        if k.startswith("grad.") and float(np.linalg.norm(d[k])) > 1e-6:
            assert gerr[k[5:]] <= 5e-2, (k, gerr[k[5:]])
    for k, n in zip(d["grad_names"].tolist(), d["grad_norms"].tolist()):
        # (a convolution bias in front of a batch-statistics BN has a zero gradient: both sides hold only round-off,
        # here ~1e-5 because the reconstruction loss is divided by the data variance)
        assert abs(float(grads[k].norm()) - n) <= 5e-2 * n + 5e-5, (k, float(grads[k].norm()), n)
    st = model.state_dict()
    for k in d.files:
        if k.startswith("stat."):
            want = torch.from_numpy(d[k])
            assert float(((st[k[5:]].cpu() - want).abs() / (1 + want.abs())).max()) <= 1e-4, k
    functional.reset_net(model)
    # the DEFAULT training path (forward on the native fp32 matrix-core kernels too) against the same fixture, same bars
    model.load_state_dict(sd)
    model.zero_grad()
    leq2, lrec2, lreal2 = model(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img)
    (leq2 + lrec2).backward()
    rel2 = {k: abs(float(v.detach()) - float(d[k])) / float(d[k]) for k, v in
            (("loss_eq", leq2), ("loss_rec", lrec2), ("real_loss_rec", lreal2))}
    gerr2 = {k[5:]: round(_rel_l2(dict(model.named_parameters())[k[5:]].grad.cpu(), torch.from_numpy(d[k])), 6)
             for k in d.files if k.startswith("grad.") and float(np.linalg.norm(d[k])) > 1e-6}
    print("F10 (native forward) measured: loss rel err", rel2, "grad rel L2", gerr2)
    parity("f10_native_forward", loss_rel=max(rel2.values()), grad_rel_l2_max=max(gerr2.values()))
    assert max(rel2.values()) <= 2e-2 and max(gerr2.values()) <= 5e-2, (rel2, gerr2)
    functional.reset_net(model)
    model.load_state_dict(sd)
    # the reference's loop body (R/main.py:136-146) runs and moves the weights; inference still works afterwards
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), weight_decay=0.001)
    for _ in range(2):
        a, b, c = model(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img)
        opt.zero_grad(); (a + b).backward(); opt.step(); functional.reset_net(model)
    assert np.isfinite(float(a.detach())) and np.isfinite(float(b.detach()))
    assert not torch.equal(model.decoder.snn_convs[3].weight.detach().cpu(), sd["decoder.snn_convs.3.weight"])
    model.eval()
    with torch.inference_mode():
        e, xr, idx = model(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img)
    functional.reset_net(model)
    assert xr.shape == (4, 1, 28, 28) and idx.shape == (4 * 49,)


# ------------------------------------------------------------------------------------------------- F6 p_sample
def test_f6_psample_steps_exact(golden_dir, dev, ops):
    d = load(golden_dir, "f6_psample.npz")
    B = int(d["B"])
    x = torch.full((B, 1, 7, 7), 128, dtype=torch.int64, device=dev)
    un = torch.zeros((B, 1, 7, 7), dtype=torch.bool, device=dev)
    for i, t in enumerate(d["ts"]):
        logits = torch.from_numpy(d["logits"][i]).permute(0, 3, 1, 2).contiguous().to(dev)   # [B,K,h,w]
        ops.psample_step(logits, x, un, int(t), 1.0, torch.from_numpy(d["u"][i]).to(dev),
                         torch.from_numpy(d["q"][i]).to(dev))
        assert torch.equal(x.cpu(), torch.from_numpy(d["x_after"][i])), f"x_t after t={t}"
        assert torch.equal(un.cpu(), torch.from_numpy(d["unmasked_after"][i])), f"unmasked after t={t}"
    assert bool(un.all()) and int(x.max()) < 128


def test_f6_trajectory_host_noise_matches_reference_cpu_path(golden_dir, dev):
    from snn_model.vq_diffusion import AbsorbingDiffusion
    d = load(golden_dir, "f6_psample.npz")
    den, sd = build_den(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = int(d["B"])
    ab.noise_source = 'host'
    rec = []
    torch.manual_seed(int(d["seed"]))
    tok = ab.sample(temp=1.0, sample_steps=int(d["steps"]), record=rec)
    want = torch.from_numpy(d["final_tokens"])
    lerr = [float((r[3].cpu().permute(0, 2, 3, 1) - torch.from_numpy(d["logits"][i])).abs().max())
            for i, r in enumerate(rec)]
    n_bad = int((tok.cpu() != want).sum())
    print(f"F6 trajectory: token mismatches {n_bad}/{want.numel()}, per-step logits max err {lerr}")
    assert tok.shape == (4, 1, 7, 7) and tok.dtype == torch.int64
    parity("f6_trajectory_4_steps", token_mismatches=n_bad, tokens=int(want.numel()), logits_max_err=max(lerr))
    assert n_bad == 0, "same seed, same RNG order as the reference CPU path -> same tokens"


def test_psample_philox_statistics(dev, ops):
    # throughput mode: on-device Philox noise. Check the unmask rate (1/t) and the categorical frequencies.
    B, K, HW = 4096, 128, 49
    g = torch.Generator().manual_seed(0)
    row = torch.randn(K, generator=g) * 1.5
    logits = row.view(1, K, 1, 1).expand(B, K, 7, 7).contiguous().to(dev)
    x = torch.full((B, 1, 7, 7), K, dtype=torch.int64, device=dev)
    un = torch.zeros((B, 1, 7, 7), dtype=torch.bool, device=dev)
    x0 = torch.empty(B * HW, dtype=torch.int64, device=dev)
    ops.psample_step(logits, x, un, 4, 1.0, None, None, seed=123, offset=0, x0_hat=x0)
    rate = float(un.float().mean())
    assert abs(rate - 0.25) < 0.01, rate
    p = torch.softmax(row, 0)
    freq = torch.bincount(x0.cpu(), minlength=K).float() / (B * HW)
    assert float((freq - p).abs().max()) < 6 * float(torch.sqrt(p.max() / (B * HW))) + 1e-3
    assert torch.equal(x.cpu()[un.cpu()], x0.cpu().view(B, 1, 7, 7)[un.cpu()]) and bool((x[~un] == K).all())
    # deterministic in (seed, offset); different offset -> different draw
    x2 = torch.full((B, 1, 7, 7), K, dtype=torch.int64, device=dev); un2 = torch.zeros_like(un)
    ops.psample_step(logits, x2, un2, 4, 1.0, None, None, seed=123, offset=0)
    assert torch.equal(x2, x)
    x3 = torch.full((B, 1, 7, 7), K, dtype=torch.int64, device=dev); un3 = torch.zeros_like(un)
    ops.psample_step(logits, x3, un3, 4, 1.0, None, None, seed=123, offset=B * HW * K)
    assert not torch.equal(x3, x)
    # temperature -> 0 : argmax of the logits wherever unmasked
    x4 = torch.full((B, 1, 7, 7), K, dtype=torch.int64, device=dev); un4 = torch.zeros_like(un)
    ops.psample_step(logits, x4, un4, 1, 1e-3, None, None, seed=5, offset=0)
    assert bool(un4.all()) and bool((x4 == int(row.argmax())).all())


# ------------------------------------------------------------------------------------------------- full-size properties
def test_full_sample_properties_b256(dev):
    """BASELINE config 2 shape (B=256, T=16; 6 of the 100 steps to keep the test short): size-independent
    properties -- every position unmasked after t=1, tokens in range, determinism, batch-slice invariance of the
    decode, u8 == quantised pred."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, _ = build_den(synth.MNIST, dev)
    model, _ = build_vae(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 256
    torch.manual_seed(1)
    tok = ab.sample(temp=1.0, sample_steps=6)
    assert tok.shape == (256, 1, 7, 7) and int(tok.min()) >= 0 and int(tok.max()) < 128
    torch.manual_seed(1)
    tok2 = ab.sample(temp=1.0, sample_steps=6)
    assert torch.equal(tok, tok2), "torch.manual_seed(s); sample() repeats after re-seeding (the reference's contract)"
    assert len(ab._graphs) == 1, "the reverse process was replayed from one captured hipGraph"
    ab.use_graph = False
    torch.manual_seed(1)
    tok3 = ab.sample(temp=1.0, sample_steps=6)
    ab.use_graph = True
    assert torch.equal(tok, tok3), "graph replay == eager launches (same kernels, same Philox key)"
    tok4 = ab.sample(temp=1.0, sample_steps=6)
    assert not torch.equal(tok, tok4), "a second call under the same seed draws a new key"
    ab2 = AbsorbingDiffusion(den, mask_id=128)
    ab2.n_samples, ab2.philox_stream = 256, 1
    torch.manual_seed(1)
    tok5 = ab2.sample(temp=1.0, sample_steps=6)
    assert torch.equal(tok, tok5), "'global' layout: the same images of the same job, whichever rank generates them"
    ab2.set_shard(256, 256)
    torch.manual_seed(1)
    assert not torch.equal(tok, ab2.sample(temp=1.0, sample_steps=6)), "the NEXT 256 images of the job are other images"
    ab2.set_shard(0, 256)
    ab2.noise_layout = 'rank'
    torch.manual_seed(1)
    tok6 = ab2.sample(temp=1.0, sample_steps=6)
    assert not torch.equal(tok, tok6), "'rank' layout: another rank (philox_stream) seeded alike draws different noise"
    pred, u8 = model.decode_tokens(tok.reshape(256, 7, 7))
    assert pred.shape == (256, 1, 28, 28) and u8.dtype == torch.uint8
    assert float(pred.abs().max()) <= 1.0
    pred_s, u8_s = model.decode_tokens(tok.reshape(256, 7, 7)[100:104].contiguous())
    assert torch.equal(pred_s, pred[100:104]) and torch.equal(u8_s, u8[100:104]), "samples are independent"
    q = (torch.clamp(pred + 0.5, 0, 1) * 255).to(torch.uint8)
    assert torch.equal(q, u8)


def test_config1_T4_vs_oracle(dev):
    """BASELINE config 1: MNIST encode->decode, B=16, T=4.  The unmodified reference cannot run T=4 ("parity
    unpinned" there); parity is HIP vs the T-generalised oracle."""
    from snn_model.vae_model import functional
    model, sd = build_vae(synth.MNIST, dev, T=4)
    img = torch.rand(16, 1, 28, 28, generator=torch.Generator().manual_seed(42)) - 0.5
    x = img.unsqueeze(0).repeat(4, 1, 1, 1, 1)
    with torch.inference_mode():
        e, xr, idx = model(x.to(dev), img.to(dev))
        functional.reset_net(model)
    oe, oxr, oidx = ref.snn_vqvae_forward(x, sd)
    same = (idx.cpu().view(16, -1) == oidx.view(16, -1)).all(1)
    err = (xr.cpu() - oxr).abs().flatten(1).max(1).values
    print(f"T=4: index-exact images {int(same.sum())}/16, max err on those {float(err[same].max()):.2e}")
    parity("config1_T4", index_exact_images=int(same.sum()), images=16, max_err=float(err.max()))
    assert e.shape == (4, 16, 16, 7, 7) and bool(same.all()) and float(err.max()) <= 1e-4


def test_main_py_call_sequence_conformance(dev):
    """Replays the attribute / call sequence R/main.py:97-108,288-323,384-401 uses on synthetic
    tensors: star imports, constructor signatures, step mode, cuda(0), load_state_dict, eval, forward, reset_net."""
    ns = {}
    exec("from snn_model.snn_layers import *\nfrom snn_model.vae_model import *\nfrom snn_model.vq_diffusion import *", ns)
    for name in ("SNN_VQVAE", "DummyModel", "AbsorbingDiffusion", "get_data_for_diff", "functional",
                 "MembraneOutputLayer", "PSP", "SNN_VAE", "VQVAE", "SNN_VQVAE_uni"):
        assert name in ns, name
    model = ns["SNN_VQVAE"](1, 16, 128, torch.tensor(0.09))
    ns["functional"].set_step_mode(net=model, step_mode='m')
    model = model.cuda(0)
    model.load_state_dict(synth.synth_vqvae_state(synth.MNIST))
    denoise_fn = ns["DummyModel"](1, 128).cuda(0)
    ns["functional"].set_step_mode(net=denoise_fn, step_mode='m')
    denoise_fn.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    abdiff = ns["AbsorbingDiffusion"](denoise_fn, mask_id=128)
    assert abdiff.n_samples == 16 and abdiff.num_classes == 128 and denoise_fn.num_embeddings == 128
    model.eval(); denoise_fn.eval()
    want_keys = set(synth.synth_vqvae_state(synth.MNIST)) | set()
    assert set(model.state_dict()) == want_keys
    assert set(denoise_fn.state_dict()) == set(synth.synth_denoiser_state(synth.MNIST))
    images = torch.rand(32, 1, 28, 28)
    norm_images = (images - 0.5).cuda(0)
    with torch.inference_mode():
        images_spike = norm_images.unsqueeze(0).repeat(16, 1, 1, 1, 1)
        e, recon_images, _ = model(images_spike, norm_images)
        ns["functional"].reset_net(model)
        assert torch.nn.functional.mse_loss(recon_images, norm_images).item() >= 0
    sample = (abdiff.sample(temp=0.5, sample_steps=3)).reshape(16, 7, 7)
    with torch.inference_mode():
        z = model.vq_layer.quantize(sample.cuda(0))
        z = z.permute(0, 3, 1, 2).contiguous()
        quantized = torch.unsqueeze(z, dim=0).repeat(16, 1, 1, 1, 1)
        quantized = model.vq_layer.poisson(quantized)
        pred = torch.tanh(model.memout(model.decoder(quantized)))
    generated_samples = np.array(np.clip((pred + 0.5).cpu().numpy(), 0., 1.) * 255, dtype=np.uint8)
    ns["functional"].reset_net(model); ns["functional"].reset_net(denoise_fn)
    assert generated_samples.shape == (16, 1, 28, 28)
    loader = [(torch.rand(8, 1, 28, 28), torch.zeros(8)) for _ in range(2)]
    idxs = ns["get_data_for_diff"](loader, model)
    assert len(idxs) == 2 and idxs[0].shape == (8, 7, 7) and idxs[0].dtype == torch.int64
    model.train()                                 # the training branch returns the reference's three losses
    with pytest.raises(RuntimeError):             # get_data_for_diff left batch-8 membrane state behind, as the reference does
        model(norm_images.unsqueeze(0).repeat(16, 1, 1, 1, 1), norm_images)
    ns["functional"].reset_net(model)
    out = model(norm_images.unsqueeze(0).repeat(16, 1, 1, 1, 1), norm_images)
    assert len(out) == 3 and all(o.dim() == 0 for o in out)
    ns["functional"].reset_net(model)


# ------------------------------------------------------------------------------------------------- BASELINE configs 3, 4, 5
def test_config3_fmnist_b1024_encode_decode_properties(dev):
    """BASELINE config 3 (FMNIST-shaped == MNIST shapes, B=1024, T=16 encode->decode): size-independent properties.
    Samples are independent, so any slice of the big batch must equal the same images run alone -- and the small run
    is oracle-checked in test_f3 -- plus determinism and range checks."""
    from snn_model.vae_model import functional
    model, sd = build_vae(synth.MNIST, dev)
    g = torch.Generator().manual_seed(42)
    images = torch.rand(1024, 1, 28, 28, generator=g) - 0.5
    with torch.inference_mode():
        x = images.to(dev).unsqueeze(0).repeat(16, 1, 1, 1, 1)
        e, xr, idx = model(x, images.to(dev))
        functional.reset_net(model)
        e2, xr2, idx2 = model(x[:, 500:508].contiguous(), images[500:508].to(dev))
        functional.reset_net(model)
        idx_fast = model.encode_images(images.to(dev), 16)
    assert e.shape == (16, 1024, 16, 7, 7) and xr.shape == (1024, 1, 28, 28) and idx.shape == (1024 * 49,)
    assert int(idx.min()) >= 0 and int(idx.max()) < 128 and float(xr.abs().max()) <= 1.0
    assert torch.equal(idx.view(1024, 49)[500:508], idx2.view(8, 49))
    assert torch.equal(xr[500:508], xr2) and torch.equal(e[:, 500:508], e2)
    assert torch.equal(idx_fast.reshape(-1), idx)
    assert bool(((e == 0) | (e == 1)).all())
    # the first 16 images are the F3 fixture images (same generator, same seed): indices must match the reference
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "f3_encode_mnist.npz"))
    assert torch.equal(images[:16], torch.from_numpy(d["images"]))
    assert torch.equal(idx.view(1024, 49)[:16].cpu(), torch.from_numpy(d["indices"]).view(16, 49))


def test_config4_cifar_shaped_b512_sample_properties(dev):
    """BASELINE config 4 (CIFAR-shaped 3x32x32, latent 8x8, B=512, T=16): the denoiser runs the 8-tile MFMA variant;
    a few reverse steps + decode; properties as above + agreement of a slice with a small direct-kernel run."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, _ = build_den(synth.CIFAR, dev)
    model, _ = build_vae(synth.CIFAR, dev)
    ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(8, 8))
    ab.n_samples = 512
    torch.manual_seed(3)
    tok = ab.sample(temp=1.0, sample_steps=4)
    assert tok.shape == (512, 1, 8, 8) and int(tok.min()) >= 0 and int(tok.max()) < 128
    pred, u8 = model.decode_tokens(tok.reshape(512, 8, 8))
    assert pred.shape == (512, 3, 32, 32) and u8.dtype == torch.uint8 and float(pred.abs().max()) <= 1.0
    # one denoiser call at B=512 on the MFMA path == the fp64 direct path on a slice of the batch
    x_t = tok.clone(); x_t[::3] = 128
    lg = den.logits_from_tokens(x_t, 3)
    den.conv_impl_request = 'direct'
    lg_d = den.logits_from_tokens(x_t[100:104].contiguous(), 3)
    den.conv_impl_request = 'auto'
    assert float((lg[100:104] - lg_d).abs().max()) <= 1e-6


def test_config5_per_rank_shape_b1024(dev):
    """BASELINE config 5 shards B=8192 over 8 GPUs: 1024 samples per rank. Two reverse steps at that shape + decode;
    the sharding itself (one uint8 all-gather) is covered on CPU by tests/test_dist_gloo.py."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    from spkdiff import dist as sdist
    den, _ = build_den(synth.MNIST, dev)
    model, _ = build_vae(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 1024
    tok = ab.sample(temp=1.0, sample_steps=2)
    _, u8 = model.decode_tokens(tok.reshape(1024, 7, 7))
    out = sdist.gather_images(u8, 1024)
    assert out.shape == (1024, 1, 28, 28) and out.dtype == torch.uint8 and int(tok.max()) < 128
    assert sdist.shard_range(8192, 3, 8) == (3072, 4096)


@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR)])
def test_conv6_time_collapsed_form(golden_dir, dev, ops, tag, cfg):
    """conv6 + mean over T as one convolution of the spike counts (spk_den_conv3x3_counts_mfma) against the per-step
    form and the reference logits; the count tensors against the spikes they summarise."""
    from spkdiff.ops import IN_PTC
    d = load(golden_dir, f"f5_denoiser_{tag}.npz")
    den, sd = build_den(cfg, dev)
    s5 = unpack(d["s5_bits"], d["s5_shape"]); s1 = unpack(d["s1_bits"], d["s1_shape"]); s4 = unpack(d["s4_bits"], d["s4_shape"])
    x4 = ops.spikes_to_ptc(s4.to(dev), chunk=32)
    r = den.conv5.run(x4, IN_PTC, final='ptc', stateful=False, chunk_out=32, want_counts=True)
    got5 = ops.ptc_to_spikes(r['ptc']).cpu()                                  # [T,B,C,H,W]
    cnt = r['cnt'].cpu()                                                       # [B,C/32,H,W,32]
    B, C = got5.shape[1], got5.shape[2]
    want_cnt = got5.sum(0).view(B, C // 32, 32, *got5.shape[3:]).permute(0, 1, 3, 4, 2)
    assert torch.equal(cnt.float(), want_cnt)
    x_t = torch.from_numpy(d["x_t"]).to(dev)
    with torch.inference_mode():
        den.collapse_conv6 = True
        a = den.logits_from_tokens(x_t, 7)
        den.collapse_conv6 = False
        b = den.logits_from_tokens(x_t, 7)
        den.collapse_conv6 = True
        full = den(x_t.float(), t=torch.from_numpy(d["t"]).to(dev))
        from snn_model.vq_diffusion import functional
        functional.reset_net(den)
    print(f"conv6 collapsed vs per-step: max abs diff {float((a - b).abs().max()):.3e}")
    assert float((a - b).abs().max()) <= 2e-7
    assert float((full.cpu() - torch.from_numpy(d["logits"])).abs().max()) <= 1e-5


def test_f2_gather_mfma_layers_teacher_forced(golden_dir, dev, ops):
    """The spiking VQ-VAE layers on the gather-MFMA kernel (spk_conv_mfma_fused_fwd): Encoder conv2 (stride 2),
    conv3 (1x1), Decoder convT1/convT2 (transposed stride 2: 4 sub-pixel classes), convT3 + membrane read-out.
    Spikes against the reference fixtures (outside the fragile set) and bit-for-bit against the fp64 direct kernel."""
    d = load(golden_dir, "f2_layers_mnist.npz")
    model, sd = build_vae(synth.MNIST, dev)
    from spkdiff.fused import FusedSequential
    from spkdiff.ops import IN_PTC
    enc, dec = model.encoder.snn_convs, model.decoder.snn_convs
    blocks = {"enc2": FusedSequential(*list(enc)[3:6]), "enc3": FusedSequential(*list(enc)[6:9]),
              "dec1": FusedSequential(*list(dec)[0:3]), "dec2": FusedSequential(*list(dec)[3:6])}
    report = {}
    for name, blk in blocks.items():
        want = unpack(d[name + "_out_bits"], d[name + "_out_shape"])
        frag = unpack(d[name + "_frag_bits"], d[name + "_out_shape"]).bool()
        x = ops.spikes_to_ptc(unpack(d[name + "_in_bits"], d[name + "_in_shape"]).to(dev))
        got = blk.run(x, IN_PTC, final='f32', stateful=False)['f32'].cpu()                    # gather-MFMA kernel
        direct = blk.run(x, IN_PTC, final='f32', stateful=False, impl='direct')['f32'].cpu()   # fp64 direct kernel
        bad = got != want
        report[name] = (int(bad.sum()), int(frag.sum()), int((got != direct).sum()))
        assert not bool((bad & ~frag).any()), f"{name}: spike differs from the reference outside the fragile set"
        assert torch.equal(got, direct), f"{name}: gather-MFMA != fp64 direct"
    print("F2 gather-MFMA (mismatch vs golden, fragile, mismatch vs direct):", report)
    parity("f2_gather_mfma_layers", mismatch_vs_golden={k: v[0] for k, v in report.items()},
           mismatch_vs_direct={k: v[2] for k, v in report.items()})
    x = ops.spikes_to_ptc(unpack(d["dec3_in_bits"], d["dec3_in_shape"]).to(dev))
    dec3 = FusedSequential(list(dec)[6])
    r = dec3.run(x, IN_PTC, final='memout', coef=model.memout.coef.flatten(), apply_tanh=True, want_u8=True)
    rd = dec3.run(x, IN_PTC, final='memout', coef=model.memout.coef.flatten(), apply_tanh=True, want_u8=True, impl='direct')
    want = torch.tanh(torch.from_numpy(d["memout"]))
    assert float((r['f32'].cpu() - want).abs().max()) <= 1e-5
    assert float((r['f32'] - rd['f32']).abs().max()) <= 1e-6
    assert float((r['u8'].float() - rd['u8'].float()).abs().max()) <= 1
    # stateful module semantics (v carried without reset) agree between the two kernels
    from snn_model.vae_model import functional
    xin = unpack(d["dec1_in_bits"], d["dec1_in_shape"]).to(dev)
    blk = blocks["dec1"]
    a1 = blk(xin); a2 = blk(xin)
    functional.reset_net(blk)
    ptc = ops.spikes_to_ptc(xin)
    b1 = blk.run(ptc, IN_PTC, final='f32', impl='direct')['f32']; b2 = blk.run(ptc, IN_PTC, final='f32', impl='direct')['f32']
    functional.reset_net(blk)
    assert torch.equal(a1, b1) and torch.equal(a2, b2) and not torch.equal(a1, a2)


@pytest.mark.parametrize("req,name,v2", [("fp6", "mfma-fp6v2", True), ("fp6", "mfma-fp6x6", False), ("i8", "mfma-i8x4", True)])
def test_denoiser_mfma_vs_direct_b64_random_tokens(dev, req, name, v2):
    """A whole denoiser call at B=64 on random, partly masked tokens: the MFMA paths (conv2..5 + time-collapsed conv6;
    second- and first-generation fp6 kernels, int8 kernel) against the fp64 direct path -- every layer's spikes bit-equal,
    logits within 2e-7."""
    den, _ = build_den(synth.MNIST, dev)
    den.conv_impl_request = req
    den.use_fp6v2 = v2
    assert den.conv_impl == name
    g = torch.Generator().manual_seed(123)
    x_t = torch.randint(0, 128, (64, 1, 7, 7), generator=g)
    x_t[torch.rand(64, 1, 7, 7, generator=g) < 0.6] = 128
    x_t = x_t.to(dev)
    rec_m, rec_d = [], []
    with torch.inference_mode():
        lm = den.logits_from_tokens(x_t, 37, record=rec_m)
        den.conv_impl_request = 'direct'
        ld = den.logits_from_tokens(x_t, 37, record=rec_d)
    from spkdiff import ops
    for i, (a, b) in enumerate(zip(rec_m, rec_d), 1):
        sa, sb = ops.ptc_to_spikes(a), ops.ptc_to_spikes(b)
        assert torch.equal(sa, sb), f"conv{i} spikes differ between the MFMA and the direct path"
        assert 0.005 < float(sa.mean()) < 0.5
    assert float((lm - ld).abs().max()) <= 2e-7


@pytest.mark.parametrize("req,name,B,v2", [("fp6", "mfma-fp6v2", 37, True), ("fp6", "mfma-fp6v2", 3, True),
                                           ("fp6", "mfma-fp6x6", 37, False), ("fp6", "mfma-fp6x6", 3, False),
                                           ("i8", "mfma-i8x4", 20, True)])
def test_denoiser_8x8_latent_mfma_vs_direct(dev, req, name, B, v2):
    """8x8 latents (BASELINE config 4): the fp6 kernels (second and first generation) in their row-band form (two items per
    image: H/2 output rows each, one halo row) and the int8 kernel, against the fp64 direct path -- every layer's spikes
    bit-equal, logits within 2e-7; B = 37 / 3: ragged and fewer-than-CUs band items."""
    den, _ = build_den(synth.CIFAR, dev)
    den.conv_impl_request = req
    den.use_fp6v2 = v2
    assert den.impl_for(8, 8) == name
    g = torch.Generator().manual_seed(321 + B)
    x_t = torch.randint(0, 128, (B, 1, 8, 8), generator=g)
    x_t[torch.rand(B, 1, 8, 8, generator=g) < 0.5] = 128
    x_t = x_t.to(dev)
    rec_m, rec_d = [], []
    with torch.inference_mode():
        lm = den.logits_from_tokens(x_t, 21, record=rec_m)
        den.conv_impl_request = 'direct'
        ld = den.logits_from_tokens(x_t, 21, record=rec_d)
    from spkdiff import ops
    for i, (a, b) in enumerate(zip(rec_m, rec_d), 1):
        sa, sb = ops.ptc_to_spikes(a), ops.ptc_to_spikes(b)
        assert torch.equal(sa, sb), f"conv{i} spikes differ between the MFMA and the direct path"
    assert float((lm - ld).abs().max()) <= 2e-7


@pytest.mark.parametrize("B", [1, 3, 9, 17])
def test_fp6_conv_ragged_item_counts_vs_direct(dev, ops, B):
    """The persistent fp6 kernel with fewer work items than CUs (B=1,3), a ragged second round (B=9: 288 items on 256
    workgroups) and an odd batch (B=17), carried membrane state included: spikes and counts bit-equal to the fp64
    direct kernel, membrane potentials within 1e-6."""
    from spkdiff.ops import IN_PTC
    from snn_model.vq_diffusion import DummyModel, functional
    torch.manual_seed(100 + B)
    den = DummyModel(1, 128).cuda(0)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    den.eval()
    for blk, cin in ((den.conv2, 64), (den.conv4, 256)):
        s_in = (torch.rand(16, B, cin, 7, 7, device=dev) < 0.05).float()
        x4 = ops.spikes_to_c4(s_in); xc = ops.spikes_to_ptc(s_in, chunk=32)
        functional.reset_net(blk)
        a1 = blk.run(x4, IN_PTC, final='ptc', chunk_out=ops.CHUNK_C4, want_counts=True)
        a2 = blk.run(x4, IN_PTC, final='ptc', chunk_out=ops.CHUNK_C4)          # second call: carried v
        va = blk[2].v.clone()
        functional.reset_net(blk)
        b1 = blk.run(xc, IN_PTC, final='f32', impl='direct')['f32']
        b2 = blk.run(xc, IN_PTC, final='f32', impl='direct')['f32']
        vb = blk[2].v.clone()
        functional.reset_net(blk)
        g1, g2 = ops.c4_to_spikes(a1['ptc']), ops.c4_to_spikes(a2['ptc'])
        assert torch.equal(g1, b1) and torch.equal(g2, b2)
        # weights below 2^-6 of their channel's maximum are fixed-point rounded at 2^-29 of that maximum
        # (den_mfma_fp6.hip): a pre-activation may round to the neighbouring fp32 value -> v within an ulp or two
        assert float((va - vb).abs().max()) <= 1e-6 and float((va != vb).float().mean()) <= 0.1
        assert 0.001 < float(g1.mean()) < 0.5 and not torch.equal(g1, g2)
        C = g1.shape[2]
        want_cnt = g1.sum(0).reshape(B, C // 32, 32, 7, 7).permute(0, 1, 3, 4, 2)
        assert torch.equal(a1['cnt'].float(), want_cnt)


@pytest.mark.parametrize("H,W", [(5, 5), (6, 6), (7, 5), (4, 7), (3, 3), (2, 2), (8, 6)])
def test_fp6_conv_other_latent_sizes_vs_direct(dev, ops, H, W):
    """The fp6 kernel away from 7x7: odd position counts (24-full-tile form + last-position kernel with fewer tiles),
    even ones and non-square latents (7-tile form), B=5: spikes and counts bit-equal to the fp64 direct kernel."""
    from spkdiff.ops import IN_PTC
    from snn_model.vq_diffusion import DummyModel, functional
    assert ops.den_fp6_supported(128, 64, 3, 1, 1, 16, H, W)
    torch.manual_seed(10 * H + W)
    den = DummyModel(1, 128).cuda(0)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    den.eval()
    B = 5
    for blk, cin in ((den.conv2, 64), (den.conv3, 128)):
        s_in = (torch.rand(16, B, cin, H, W, device=dev) < 0.06).float()
        x4 = ops.spikes_to_c4(s_in); xc = ops.spikes_to_ptc(s_in, chunk=32)
        a = blk.run(x4, IN_PTC, final='ptc', stateful=False, chunk_out=ops.CHUNK_C4, want_counts=True)
        d = blk.run(xc, IN_PTC, final='f32', stateful=False, impl='direct')['f32']
        got = ops.c4_to_spikes(a['ptc'])
        assert torch.equal(got, d), f"{H}x{W}: fp6 path != direct"
        C = got.shape[2]
        assert torch.equal(a['cnt'].float(), got.sum(0).reshape(B, C // 32, 32, H, W).permute(0, 1, 3, 4, 2))
        assert float(got.mean()) > 0.0005


def test_syops_report_matches_golden_spike_rates(golden_dir, dev):
    """the syops-style energy report (R/syops/ops.py:14-24,121-158) from the fused kernels' own spike
    maps: layer firing rates equal the reference fixture's, ACs = overall * input rate, conv1 counts as MACs."""
    from spkdiff import syops
    d = load(golden_dir, "f5_denoiser_mnist.npz")
    den, _ = build_den(synth.MNIST, dev)
    x_t = torch.from_numpy(d["x_t"]).to(dev); t = torch.from_numpy(d["t"]).to(dev)
    logits, rep = syops.denoiser_syops(den, x_t, t)
    assert float((logits.cpu() - torch.from_numpy(d["logits"])).abs().max()) <= 1e-5
    B = x_t.shape[0]
    for i in range(1, 6):
        want = float(unpack(d[f"s{i}_bits"], d[f"s{i}_shape"]).mean())
        assert abs(rep[i - 1]["out_rate"] - want) <= 1e-6, (i, rep[i - 1]["out_rate"], want)
    assert rep[0]["macs"] == (9 * 2 * 64 + 64) * 16 * B * 49 and rep[0]["acs"] == 0.0
    assert rep[3]["overall"] == (9 * 256 * 512 + 512) * 16 * B * 49
    assert abs(rep[3]["acs"] - rep[3]["overall"] * rep[2]["out_rate"]) <= 1e-6 * rep[3]["overall"]
    assert rep[-1]["layer"] == "total" and rep[-1]["acs"] < 0.2 * rep[-1]["overall"]


def test_f11_syops_report_vs_reference_fixture(golden_dir, dev):
    """SURVEY §8f item 4, whole models: ``syops.get_model_complexity_info`` (the reference's entry point, R/main.py:325-338)
    on SNN_VQVAE and DummyModel against F11 = what the REAL R/syops produced on the reference models -- as shipped (its
    exact-type mapping only hooks the LIF layers) and with its conv / bn hooks registered for the spikingjelly layer types
    through custom_modules_hooks.  Per-module [overall, ACs, MACs, rate %], totals and parameter counts; the firing rates
    are counted on the device (spk_count_spikes)."""
    import io
    import syops.engine as seng
    import syops.ops as sops
    from syops import get_model_complexity_info
    from spikingjelly.activation_based import layer
    d = load(golden_dir, "f11_syops.npz")
    model, sdv = build_vae(synth.MNIST, dev)
    den, sdd = build_den(synth.MNIST, dev)
    assert synth.state_checksum(sdv) == str(d["weights_crc_vae"]) and synth.state_checksum(sdd) == str(d["weights_crc_den"])
    img = torch.from_numpy(d["images"]).to(dev)
    kw = {"vae": {"x": img.unsqueeze(0).repeat(16, 1, 1, 1, 1), "image": img},
          "den": {"x": torch.from_numpy(d["x_t"]).float().to(dev), "t": torch.from_numpy(d["t"]).to(dev)}}
    custom = {layer.Conv2d: sops.conv_syops_counter_hook, layer.ConvTranspose2d: sops.conv_syops_counter_hook,
              layer.BatchNorm2d: sops.bn_syops_counter_hook}
    worst = 0.0
    for mname, mod in (("vae", model), ("den", den)):
        for cname, cm in (("default", {}), ("custom", custom)):
            key = f"{mname}_{cname}"
            total, params = get_model_complexity_info(mod, (1, 28, 28), None, print_per_layer_stat=True, as_strings=False,
                                                      input_constructor=lambda res, k=kw[mname]: k, ost=io.StringIO(),
                                                      custom_modules_hooks=cm)
            per = seng.get_syops_pytorch.last_per_module
            assert list(per) == [str(n) for n in d[key + "_names"]], key
            got = np.stack(list(per.values()))
            want = d[key + "_per"]
            assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 2], want[:, 2]), key   # op counts: exact
            rel = np.abs(got - want) / (np.abs(want) + 1e-30)
            worst = max(worst, float(rel[want != 0].max()))
            assert float(rel[want != 0].max()) <= 1e-6, (key, float(rel.max()))                            # rates: fp32 vs count
            assert np.allclose(total, d[key + "_total"], rtol=1e-6, atol=0) and params == int(d[key + "_params"])
    strings, pstr = get_model_complexity_info(den, (1, 28, 28), None, print_per_layer_stat=False, as_strings=True,
                                              input_constructor=lambda res: kw["den"], custom_modules_hooks=custom)
    assert strings[1].endswith(" G Ops") and pstr == "3.1 M"
    # no hook is left behind: the next call is fused again and gives the fused path's logits
    from spkdiff.fused import has_hooks
    assert not has_hooks(den) and not has_hooks(model)
    parity("f11_syops_report", modules_compared=int(sum(len(d[k]) for k in d.files if k.endswith("_names"))),
           worst_relative_difference=worst)


def test_sampler_trajectory_vs_live_oracle_12_steps(dev):
    """12 reverse steps, B=8, noise drawn on the host in the reference's order under the same torch.manual_seed:
    the HIP sampler must reproduce the CPU oracle's tokens (the oracle is bit-identical to the reference, F6)."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 8
    ab.noise_source = 'host'
    torch.manual_seed(2024)
    tok = ab.sample(temp=0.9, sample_steps=12).cpu()
    torch.manual_seed(2024)
    want = ref.absorbing_sample(sd, 8, 128, 0.9, 12, 7, 16)
    n_bad = int((tok != want).sum())
    print(f"12-step trajectory B=8: token mismatches {n_bad}/{want.numel()}")
    parity("trajectory_12_steps_vs_live_oracle", token_mismatches=n_bad, tokens=int(want.numel()))
    assert n_bad == 0


def test_get_data_for_diff_matches_reference_fixture_f12(golden_dir, dev):
    """bulk encode of a loader to code indices.  F12 was produced by the REAL get_data_for_diff
    (R/snn_model/vq_diffusion.py:23-36), which carries the membrane state from batch to batch (no reset_net in its loop):
    409 of the 588 indices of batches 2-3 depend on it.  carry_state=True reproduces it index for index;
    carry_state=False equals the per-batch fresh-state encode."""
    from snn_model.vq_diffusion import get_data_for_diff, functional
    d = load(golden_dir, "f12_get_data_for_diff.npz")
    model, sd = build_vae(synth.MNIST, dev)
    assert synth.state_checksum(sd) == str(d["weights_crc"])
    loader = [(torch.from_numpy(im), torch.zeros(im.shape[0])) for im in d["images"]]
    got = get_data_for_diff(loader, model)
    functional.reset_net(model)
    fresh = get_data_for_diff(loader, model, carry_state=False)
    bad = sum(int((g != torch.from_numpy(w)).sum()) for g, w in zip(got, d["indices"]))
    bad_fresh = sum(int((g != torch.from_numpy(w)).sum()) for g, w in zip(fresh, d["indices_fresh_state"]))
    parity("f12_get_data_for_diff", index_mismatches_carried_state=bad, index_mismatches_fresh_state=bad_fresh,
           indices=int(d["indices"].size))
    assert bad == 0 and bad_fresh == 0
    assert got[0].shape == (6, 7, 7) and got[0].dtype == torch.int64
    live = ref.get_data_for_diff(loader, sd)
    assert all(torch.equal(a, b) for a, b in zip(got, live))


@pytest.mark.parametrize("tag", ["mnist", "mnist_trained"])
def test_f13_sample_100_steps_and_decode_vs_reference_fixture(golden_dir, dev, tag):
    """The benchmark's own length, end to end (VERDICT r1 missing #3): 100 reverse steps, B = 8, noise drawn on the host in
    the reference's order under torch.manual_seed(1313), then the decode glue of R/main.py:388-401 down to uint8.  F13
    holds what the REAL reference classes produced (R/snn_model/vq_diffusion.py:103-142): tokens must be equal, decoded
    pixels within 1e-4, uint8 equal away from truncation edges."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    d = load(golden_dir, "f13t_sample_trained.npz" if trained(tag) else "f13_sample_100_steps.npz")
    den, sdd = build_den(synth.MNIST, dev, weights='trained' if trained(tag) else 'synth')
    model, sdv = build_vae(synth.MNIST, dev, weights='trained' if trained(tag) else 'synth')
    assert synth.state_checksum(sdd) == str(d["weights_crc_den"]) and synth.state_checksum(sdv) == str(d["weights_crc_vae"])
    B, steps = int(d["B"]), int(d["steps"])
    out = {}
    for skip in (False, True):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples, ab.noise_source, ab.skip_untouched = B, 'host', skip
        torch.manual_seed(int(d["seed"]))
        out[skip] = ab.sample(temp=float(d["temp"]), sample_steps=steps)
    tok = out[False]
    want = torch.from_numpy(d["tokens"])
    n_bad = int((tok.cpu() != want).sum())
    pred, u8 = model.decode_tokens(tok.reshape(B, 7, 7))
    err = float((pred.cpu() - torch.from_numpy(d["pred"])).abs().max()) if n_bad == 0 else float('nan')
    safe = d["u8_edge_dist"] > 1e-3
    u8_bad = int((u8.cpu().numpy()[safe] != d["u8"][safe]).sum())
    u8_bad_all = int((u8.cpu().numpy() != d["u8"]).sum())
    parity("f13_sample_100_steps_decode_" + tag, token_mismatches=n_bad, tokens=int(want.numel()), pred_max_abs_err=err,
           u8_mismatches_away_from_edges=u8_bad, u8_mismatches_all=u8_bad_all, pixels=int(d["u8"].size),
           elimination_same_tokens=bool(torch.equal(out[True], tok)))
    assert n_bad == 0, "100-step trajectory differs from the reference"
    assert err <= 1e-4 and u8_bad == 0
    assert torch.equal(out[True], tok), "untouched-image elimination changes no token over the full trajectory"
    # and live against the oracle under another seed and temperature (the oracle is pinned to the reference by F6 and F13; 30 steps:
    # the full length is the fixture's part above -- the host oracle is what this suite's wall time is made of)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples, ab.noise_source = 4, 'host'
    torch.manual_seed(4242)
    tok2 = ab.sample(temp=0.8, sample_steps=30).cpu()
    torch.manual_seed(4242)
    ou8, otok = ref.sample_images(sdv, sdd, 4, 128, 0.8, 30, 7, 16)
    _, u82 = model.decode_tokens(tok2.reshape(4, 7, 7).to(dev))
    assert torch.equal(tok2, otok)
    assert int((u82.cpu().numpy().astype(int) - ou8.astype(int)).__abs__().max()) <= 1


@pytest.mark.parametrize("K", [100, 200, 256, 512])
def test_other_codebook_sizes_sample_and_train(dev, K):
    """The reference accepts any --codebook_size (R/main.py:58).  Round 6: every K stays on the matrix-core path -- conv2..conv5
    never see K; the logits layer's output channels are zero-padded to a multiple of 16 inside its packed weights and the fused
    reverse-step tail takes 1 <= K <= 512 (classes >= K masked as spk_psample_step masks them).  K = 100 / 200 (not multiples of
    16), 256, 512 (p_sample with 8 classes per lane; the cross-entropy tile no longer fits 64 KB of LDS) must run 'mfma-fp6v2',
    agree with the fp64 direct kernels and with the oracle (logits, tokens of host-noise trajectories through the fused tail,
    through the three-launch form and through the direct kernels), sample in range under Philox + graph replay, and train."""
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional
    import dataclasses
    cfg = dataclasses.replace(synth.MNIST, num_embeddings=K)
    sd = synth.synth_denoiser_state(cfg)
    den = DummyModel(1, K).cuda(0)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(sd)
    den.eval()
    assert den.impl_for(7, 7) == den.impl_for(8, 8) == 'mfma-fp6v2' and den.tail_fusable(7, 7)
    g = torch.Generator().manual_seed(K)
    x_t = torch.randint(0, K + 1, (3, 1, 7, 7), generator=g)
    t = torch.tensor([5, 50, 99])
    with torch.inference_mode():
        logits = den(x_t.float().to(dev), t=t.to(dev))
        functional.reset_net(den)
    want = ref.denoiser_forward(x_t.float(), t, sd, 16)
    lerr = float((logits.cpu() - want).abs().max())
    ab = AbsorbingDiffusion(den, mask_id=K)
    ab.n_samples, ab.noise_source = 3, 'host'
    torch.manual_seed(K)
    tok = ab.sample(temp=1.0, sample_steps=6).cpu()
    torch.manual_seed(K)
    otok = ref.absorbing_sample(sd, 3, K, 1.0, 6, 7, 16)
    n_bad = int((tok != otok).sum())
    # the same trajectory without the fused tail (conv6 on counts -> logits -> spk_psample_step) and on the fp64 direct kernels
    forms = {}
    for name, (tailf, req) in {"three_launches": (False, 'auto'), "direct_f64": (True, 'direct')}.items():
        den.use_step_tail, den.conv_impl_request = tailf, req
        assert den.impl_for(7, 7) == ('direct-f64' if req == 'direct' else 'mfma-fp6v2') and not den.tail_fusable(7, 7)
        torch.manual_seed(K)
        forms[name] = int((ab.sample(temp=1.0, sample_steps=6).cpu() != otok).sum())
        with torch.inference_mode():
            lg2 = den(x_t.float().to(dev), t=t.to(dev))
            functional.reset_net(den)
        forms[name + "_logits"] = float((lg2 - logits).abs().max())
    den.use_step_tail, den.conv_impl_request = True, 'auto'
    # Philox noise: graph replay of the fused-tail loop == the eager three-launch loop on the same key
    ab.noise_source = 'philox'
    torch.manual_seed(K + 1)
    tokp = ab.sample(temp=1.0, sample_steps=6)
    den.use_step_tail, ab.use_graph = False, False
    torch.manual_seed(K + 1)
    tokp3 = ab.sample(temp=1.0, sample_steps=6)
    den.use_step_tail, ab.use_graph = True, True
    parity(f"codebook_size_{K}", logits_max_abs_err=lerr, token_mismatches_6_steps=n_bad, impl=den.impl_for(7, 7),
           other_forms=forms, philox_tail_graph_vs_eager_three_launches=int((tokp != tokp3).sum()))
    assert logits.shape == (3, K, 7, 7) and lerr <= 1e-5
    assert n_bad == 0 and int(tokp.max()) < K and int(tokp.min()) >= 0
    assert forms["three_launches"] == 0 and forms["direct_f64"] == 0, forms
    assert forms["three_launches_logits"] == 0.0 and forms["direct_f64_logits"] <= 1e-5, forms
    assert torch.equal(tokp, tokp3)
    # one training step on the same (t, u) as the oracle (masked cross-entropy with a [K][49] tile; K = 512 takes the
    # no-LDS-tile kernel); library forward convolutions may flip a few spikes: the F9 tolerances
    den.train()
    x0 = torch.randint(0, K, (4, 1, 7, 7), generator=g).float()
    t4 = torch.tensor([3, 17, 30, 49])
    u4 = torch.rand(4, 1, 7, 7, generator=g)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
           for k, v in sd.items()}
    lo, (_, x_t4, x0i4, _, _) = ref.train_loss(x0, sdo, K, t=t4, u=u4)
    logits4 = den(x_t4.to(dev), t4.to(dev))
    loss = ab._loss_from_logits(logits4, x0i4.to(dev), t4.to(dev))
    loss.backward()
    functional.reset_net(den)
    parity(f"codebook_size_{K}_train", loss=float(loss), loss_oracle=float(lo))
    assert abs(float(loss) - float(lo)) <= 1e-3 * max(1.0, abs(float(lo)))
    assert den.conv6[0].weight.grad is not None and bool(torch.isfinite(den.conv6[0].weight.grad).all())


@pytest.mark.gpu
def test_constant_input_lif_lookup_equals_the_scan_on_every_float_around_1_2():
    """The time-invariant-input kernel (tinv_lif_kernel: encoder conv1, spike generator, denoiser conv1) replaces the sixteen
    LIF steps of a stateless call by a table look-up.  Here an identity 1x1 layer (weight 1, BN a = 1, b = 0) feeds it EVERY
    float from 0.9375 to 2.125 (2^23 + 2^21 + 2^19 values) plus specials, and the spikes must equal those of the LIF scan
    kernel (spk_lif_fwd, parity-pinned by F1) on the same inputs; and a stateful call (carried v) must still match too."""
    import torch
    from spkdiff import ops
    from spkdiff.ops import IN_TINV, MODE_LIF
    dev = torch.device("cuda")
    lo, hi = 0x3F700000, 0x40080000
    bits = torch.arange(lo, hi, dtype=torch.int64, device=dev)
    # (+inf is left out: the reference's reset v = v_reset * s + (1 - s) * h turns an infinite h into NaN, the fused kernels'
    #  v = s ? 0 : h does not -- csrc/spk_common.h; pre-activations are finite)
    special = torch.tensor([0.0, -0.0, 1.0, 2.0, -float("inf"), float("nan"), 3.4e38, 1e-45, -5.0, 0.5, 4.0, 1e9],
                           dtype=torch.float32, device=dev)
    x = torch.cat([bits.to(torch.int32).view(torch.float32), special])
    n = x.numel()
    B = (n + 4095) // 4096
    xp = torch.cat([x, torch.zeros(B * 4096 - n, device=dev)]).reshape(B, 1, 64, 64).contiguous()
    w = torch.ones(16, 1, 1, 1, device=dev)
    packed = ops.pack_conv_weight(w, False)
    one, zero = torch.ones(16, device=dev), torch.zeros(16, device=dev)
    r = ops.conv_fused(xp, packed, None, in_kind=IN_TINV, T=16, mode=MODE_LIF, k=1, stride=1, pad=0, bn_a=one, bn_b=zero,
                       want_ptc=True, chunk_out=16)
    got = r["ptc"][:, 0, :, :, :, 0].reshape(-1, 16)[:n]                       # [n, T] spikes of channel 0
    assert torch.equal(r["ptc"][:, 0, :, :, :, 5].reshape(-1, 16)[:n], got)    # every channel computes the same neuron
    ref = ops.lif_fwd(x.unsqueeze(0).repeat(16, 1), torch.zeros(n, device=dev), spike_dtype=ops.SPIKE_U8).t()
    bad = int((got != ref).any(dim=1).sum())
    parity("tinv_lookup_every_float", inputs=n, mismatching_inputs=bad)
    assert bad == 0
    # carried state: the kernel must run the steps (v0 != 0), and leave the same v behind as the scan
    m = 1 << 16
    xs = xp[:m // 4096]
    v0 = (torch.rand(m // 4096, 16, 64, 64, device=dev) * 0.9).contiguous()
    v_k = v0.clone()
    r2 = ops.conv_fused(xs, packed, None, in_kind=IN_TINV, T=16, mode=MODE_LIF, k=1, stride=1, pad=0, bn_a=one, bn_b=zero,
                        want_ptc=True, chunk_out=16, v=v_k)
    v_ref = v0[:, 3].reshape(-1).clone()
    ref2 = ops.lif_fwd(xs.reshape(1, -1).repeat(16, 1), v_ref, spike_dtype=ops.SPIKE_U8).t()
    assert torch.equal(r2["ptc"][:, 0, :, :, :, 3].reshape(-1, 16), ref2)
    assert torch.equal(v_k[:, 3].reshape(-1), v_ref)


@pytest.mark.gpu
def test_staged_time_invariant_layer_equals_the_stepwise_kernel():
    """tinv_lif_staged_kernel (round 5: a chunk's positions staged once in LDS, every thread keeps its channel; stateless calls of the 3x3
    first layers and of the 1x1 spike generator) against tinv_lif_kernel running the sixteen steps (the same call with a carried membrane
    state of zeros): ragged shapes -- partial chunks, both strides, 1 / 2 / 3 input channels, 16 / 32 / 64 output channels -- bit for bit."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("tinv_stress", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tinv_stress.py"))
    ts = importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
    g = torch.Generator().manual_seed(5)
    dev = torch.device("cuda")
    bad = tot = 0
    for _ in range(40):
        m, e, _f = ts.one_case(g, dev)
        bad += m; tot += e
    parity("tinv_staged_vs_stepwise", spikes=tot, mismatches=bad)
    assert bad == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name", ["mnist", "cifar"])
def test_spike_generator_by_token_table_equals_the_layer_by_layer_front_end(dev, cfg_name):
    """decode_tokens' front end as a per-token pattern table (spk_spikegen_tokens_s32: embedding + 1x1 conv + BN + LIF from the reset
    state + nibble packing in two launches) against the three launches it replaces (R/main.py:388-392, vae_model.py:54-56,66-71):
    the S32 spikes bit for bit -- every token of the codebook, an out-of-range token (NaN embedding: no spikes) -- and the decoded
    images of both paths equal."""
    from snn_model import vae_model
    from spkdiff import ops as O
    from spkdiff.ops import IN_TINV
    cfg = synth.MNIST if cfg_name == "mnist" else synth.CIFAR
    model, _sd = build_vae(cfg, dev)
    K, L = cfg.num_embeddings, cfg.latent
    g = torch.Generator().manual_seed(3)
    tokens = torch.randint(0, K, (5, L, L), generator=g)
    tokens.view(-1)[:K] = torch.arange(K)                          # (5 L^2 >= 245 > K = 128: every code occurs)
    tokens = tokens.to(dev)
    s32 = model.vq_layer.poisson.tokens_to_s32(tokens, model.vq_layer.embeddings.weight, T=16)
    assert s32 is not None
    zq = O.embedding(tokens, model.vq_layer.embeddings.weight, nchw_hw=(L, L))
    ptc = model.vq_layer.poisson.run(zq, IN_TINV, final='ptc', T=16, stateful=False)['ptc']
    assert torch.equal(s32.view(torch.uint8), O.ptc_to_s32(ptc).view(torch.uint8))
    bad_tok = tokens.clone(); bad_tok[0, 0, 0] = K; bad_tok[1, 1, 1] = -1
    s32b = model.vq_layer.poisson.tokens_to_s32(bad_tok, model.vq_layer.embeddings.weight, T=16)
    zqb = O.embedding(bad_tok, model.vq_layer.embeddings.weight, nchw_hw=(L, L))
    ptcb = model.vq_layer.poisson.run(zqb, IN_TINV, final='ptc', T=16, stateful=False)['ptc']
    assert torch.equal(s32b.view(torch.uint8), O.ptc_to_s32(ptcb).view(torch.uint8))
    assert vae_model.SPIKEGEN_BY_TOKEN
    f_t, u_t = model.decode_tokens(tokens, 16)
    try:
        vae_model.SPIKEGEN_BY_TOKEN = False
        f_l, u_l = model.decode_tokens(tokens, 16)
    finally:
        vae_model.SPIKEGEN_BY_TOKEN = True
    parity("spikegen_by_token_" + cfg_name, tokens=int(tokens.numel()), pixels_differing=int((u_t != u_l).sum()))
    assert torch.equal(u_t, u_l) and torch.equal(f_t, f_l)
    # the table is kept across calls while the parameters are unchanged: an in-place update of any of them must rebuild it
    gen_conv = model.vq_layer.poisson[0]
    with torch.no_grad():
        gen_conv.weight.mul_(1.37)
        model.vq_layer.embeddings.weight.add_(0.01)
        model.vq_layer.poisson[1].running_mean.add_(0.02)
    s32c = model.vq_layer.poisson.tokens_to_s32(tokens, model.vq_layer.embeddings.weight, T=16)
    zqc = O.embedding(tokens, model.vq_layer.embeddings.weight, nchw_hw=(L, L))
    ptcc = model.vq_layer.poisson.run(zqc, IN_TINV, final='ptc', T=16, stateful=False)['ptc']
    assert torch.equal(s32c.view(torch.uint8), O.ptc_to_s32(ptcc).view(torch.uint8))
    assert not torch.equal(s32c.view(torch.uint8), s32.view(torch.uint8))


@pytest.mark.gpu
def test_spike_generator_table_follows_invalidate_derived_and_is_not_shared(dev):
    """The spike-pattern table of decode_tokens' front end must never be stale (ADVICE r5, high + medium):
    (a) writes through ``.data`` bump no ``_version``; ``invalidate_derived`` is the contract for them
        (R/main.py's EMA / optimizer-in-graph flows) and must reach the table: codebook, generator weights and BN alike;
    (b) two models of the same shapes keep SEPARATE tables (an eager call of model A between two calls of model B must not make B read A's);
    (c) a call under stream capture records nothing: the eager call after it rebuilds, and graph replays between eager calls of another
        parameter state cannot leave a foreign table behind a matching key."""
    from spkdiff import ops as O
    from spkdiff.fused import invalidate_derived
    from spkdiff.ops import IN_TINV
    cfg = synth.MNIST
    K, L = cfg.num_embeddings, cfg.latent
    tokens = (torch.arange(3 * L * L) % K).reshape(3, L, L).to(dev)

    def layerwise(m):
        zq = O.embedding(tokens, m.vq_layer.embeddings.weight, nchw_hw=(L, L))
        return O.ptc_to_s32(m.vq_layer.poisson.run(zq, IN_TINV, final='ptc', T=16, stateful=False)['ptc']).view(torch.uint8)

    def table(m):
        from spkdiff.fused import derived_epoch
        return m.vq_layer.poisson.tokens_to_s32(tokens, m.vq_layer.embeddings.weight, T=16,
                                                epoch=derived_epoch(m.vq_layer)).view(torch.uint8)
    model, _ = build_vae(cfg, dev)
    s0 = table(model)
    assert torch.equal(s0, layerwise(model))
    stale = 0
    # (a) one parameter at a time through .data, then invalidate_derived
    for name, write in (("codebook", lambda: model.vq_layer.embeddings.weight.data.mul_(-1.0)),
                        ("generator_weight", lambda: model.vq_layer.poisson[0].weight.data.mul_(1.61)),
                        ("bn_running_mean", lambda: model.vq_layer.poisson[1].running_mean.data.add_(0.05))):
        before = table(model).clone()
        write()
        invalidate_derived(model)
        after = table(model)
        want = layerwise(model)
        stale += int(not torch.equal(after, want))
        assert not torch.equal(after, before), name + ": the write must change the spikes for this test to mean anything"
        assert torch.equal(after, want), name + ": stale table after a .data write + invalidate_derived"
    # (b) a second model with other weights, same shapes, calls interleaved
    other, _ = build_vae(cfg, dev)
    with torch.no_grad():
        other.vq_layer.embeddings.weight.mul_(0.5)
    invalidate_derived(other)
    for _ in range(2):
        a, b = table(model), table(other)
        stale += int(not torch.equal(a, layerwise(model))) + int(not torch.equal(b, layerwise(other)))
    assert not torch.equal(table(model), table(other))
    # (c) a capture of `other`'s decode, replays between eager calls
    f_ref, u_ref = other.decode_tokens(tokens, 16)
    from spkdiff.fused import derived_refs
    refs = derived_refs(other)                                    # (what a graph owner keeps alive: the addresses its launches bake)
    store = {}                                                    # (the captured launches' flag workspaces: theirs alone)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), O.flag_scope(store):
        other.decode_tokens(tokens, 16)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), O.flag_scope(store):
        f_g, u_g = other.decode_tokens(tokens, 16)
    u_eager_after_capture = other.decode_tokens(tokens, 16)[1]
    g.replay()
    stale += int(not torch.equal(u_g, u_ref)) + int(not torch.equal(u_eager_after_capture, u_ref))
    other.vq_layer.embeddings.weight.data.mul_(-1.0)            # new parameter state, same versions
    invalidate_derived(other)
    u_new = other.decode_tokens(tokens, 16)[1]
    g.replay()                                # a STALE graph (its owner should have re-captured): it must not reach the eager table
    torch.cuda.synchronize()
    del refs
    stale += int(not torch.equal(table(other), layerwise(other))) + int(not torch.equal(table(model), layerwise(model)))
    assert not torch.equal(u_new, u_ref)
    parity("spikegen_table_invalidation", stale_tables=stale)
    assert stale == 0


@pytest.mark.gpu
def test_two_live_sampler_graphs_on_one_model_replay_independently(dev):
    """Two samplers (dense and elimination forms) on ONE denoiser, both replaying captured hipGraphs, interleaved over several
    seeds: every replay must equal the eager loop.  (Regression: buffers a captured graph addresses by raw pointer -- the
    `unmasked` mask, the active list -- were released after the capture; their blocks went to the next sampler's allocations
    and the first graph's replays then wrote into the second one's state.)"""
    from snn_model.vq_diffusion import AbsorbingDiffusion, DummyModel, functional
    cfg = synth.MNIST
    den = DummyModel(1, cfg.num_embeddings).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(cfg))
    den.eval()

    def mk(skip, graph):
        ab = AbsorbingDiffusion(den, mask_id=cfg.num_embeddings)
        ab.n_samples = 24
        ab.skip_untouched, ab.use_graph = skip, graph
        return ab
    samplers = {"dense_graph": mk(False, True), "elim_graph": mk(True, True), "lists_off_graph": mk(True, True), "eager": mk(False, False)}
    samplers["lists_off_graph"].list_positions = False
    bad = 0
    for seed in range(4):
        out = {}
        for name, ab in samplers.items():
            torch.manual_seed(4321 + seed)
            out[name] = ab.sample(temp=1.0, sample_steps=40).cpu()
        bad += sum(int((v != out["eager"]).sum()) for v in out.values())
    parity("two_live_graphs_one_model", replays=12, token_mismatches=bad)
    assert bad == 0


# ------------------------------------------------------------------------------------------------- round 3: the timed configuration
def _philox_oracle_tokens(ops, dev, sd, key, B, steps, latent, temp=1.0, K=128, offset_of=None):
    """Run the CPU oracle on the noise the Philox-mode sampler drew: spk_philox_noise dumps (u, q) per reverse step with the
    (seed, offset) arguments the captured launches use (``offset_of(step index)`` = AbsorbingDiffusion._step_offset: the
    'global' layout's step * 2^40 + first image * h*w*K by default; R/snn_model/vq_diffusion.py:116,134-138)."""
    HW = latent * latent
    if offset_of is None:
        offset_of = lambda i: i * (1 << 40)

    def noise(t):
        u, q = ops.philox_noise(key, offset_of(steps - t), B, HW, K, dev)
        return u.cpu().view(B, 1, latent, latent), q.cpu()
    return ref.absorbing_sample(sd, B, K, temp, steps, latent, 16, noise=noise)


@pytest.mark.parametrize("B,steps", [(4, 100), (256, 6)])
def test_timed_configuration_philox_graph_vs_oracle_on_dumped_noise(dev, ops, B, steps):
    """The configuration bench.py times -- noise_source='philox', the whole reverse process replayed from ONE hipGraph -- pinned
    to the oracle bit for bit: the noise the device drew is dumped (spk_philox_noise) and fed to the CPU oracle, whose tokens
    must equal the sampler's in the dense form AND with the untouched-image elimination + position lists; decode within 1e-4."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.MNIST, dev)
    model, sd_v = build_vae(synth.MNIST, dev)
    got = {}
    key = None
    for name, skip, lists in (("dense", False, False), ("elim+lists", True, True), ("elim", True, False)):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples, ab.skip_untouched, ab.list_positions = B, skip, lists
        assert ab.noise_source == 'philox' and ab.use_graph
        torch.manual_seed(777)
        k = ab._philox_key()
        assert key is None or k == key
        key = k
        torch.manual_seed(777)
        got[name] = ab.sample(temp=1.0, sample_steps=steps)
        assert len(ab._graphs) == 1, "replayed from a captured hipGraph"
    want = _philox_oracle_tokens(ops, dev, sd, key, B, steps, 7)
    bad = {n: int((t.cpu() != want).sum()) for n, t in got.items()}
    print(f"philox+graph B={B} steps={steps}: token mismatches vs oracle on dumped noise {bad} of {want.numel()}")
    parity(f"timed_configuration_philox_graph_B{B}_{steps}steps", token_mismatches=bad, tokens=int(want.numel()))
    assert all(v == 0 for v in bad.values()), bad
    n = min(B, 16)
    pred, u8 = model.decode_tokens(got["dense"].reshape(B, 7, 7)[:n].contiguous())
    opred = ref.decode_tokens(want.reshape(B, 7, 7)[:n], sd_v, 16)
    err = float((pred.cpu() - opred).abs().max())
    parity(f"timed_configuration_decode_B{B}", pixel_max_abs_err=err)
    assert err <= 1e-4


def test_f15_bench_job_tokens_vs_fixture(golden_dir, dev):
    """The bench line's OWN job in the driver's suite: the FIRST and the LAST timed batch of the default
    ``python bench.py`` (seed 42, --steps 20 --warmup 5: draws #8 and #27 of torch's CPU generator are their keys; B = 256, 100
    reverse steps, Philox noise, one hipGraph replay, dense) against tests/golden/f15_bench_job_tokens.npz -- the fp32 CPU oracle
    on the dumped noise of exactly these batches (oracle/gen_f15_bench_job.py; R/snn_model/vq_diffusion.py:103-142).  Required:
    token equality on every image outside the fixture's recorded fragile set (images where the oracle's own fp32 convolution
    rounding flips a spike whose exact membrane potential sits on the threshold: margin recorded), and inside it equality with
    the oracle evaluated with exact convolutions.  GPU side: two graph replays."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    from spkdiff import dist as sdist
    z = load(golden_dir, "f15_bench_job_tokens.npz")
    seed, warmup, steps, B, sample_steps, T, setup = (int(v) for v in z["config"])
    den, sd = build_den(synth.MNIST, dev)
    assert synth.state_checksum(sd) == bytes(z["weights_checksum"]).decode(), "the fixture was generated for other weights"
    rep = {}
    for tag, skip in (("first", setup + warmup), ("last", setup + warmup + steps - 1)):
        ab = AbsorbingDiffusion(den, mask_id=128)
        ab.n_samples, ab.skip_untouched, ab.sync_key = B, False, False
        ab.set_shard(0, B)
        torch.manual_seed(seed)
        for _ in range(skip):
            ab._philox_key()
        hip = ab.sample(temp=1.0, sample_steps=sample_steps).cpu().reshape(B, -1)
        assert int(ab.last_key) == int(z[tag + "_key"][0]), "the key of the bench job's batch"
        want = torch.from_numpy(z[tag + "_tokens_oracle"].astype(np.int64))
        eq = (hip == want).all(1)
        frag = [int(b) for b in z[tag + "_fragile_images"]]
        exact = torch.from_numpy(z[tag + "_fragile_tokens_exact"].astype(np.int64))
        stray = [int(i) for i in torch.nonzero(~eq).flatten().tolist() if int(i) not in frag]
        frag_ok = [bool(torch.equal(hip[b], exact[j])) for j, b in enumerate(frag)]
        cs = [sdist.token_checksum(hip[i:i + 1], i) for i in range(B)]
        cs_eq = sum(int(c == int(w)) for c, w in zip(cs, z[tag + "_image_checksum"]))
        rep[tag] = dict(images_equal=int(eq.sum()), images=B, tokens_differing=int((hip != want).sum()), fragile=frag,
                        fragile_equal_exact_convolution_oracle=frag_ok, fragile_margins=[float(m) for m in z[tag + "_fragile_margin"]],
                        image_checksums_equal=cs_eq, stray=stray)
        assert not stray, f"{tag}: images {stray} differ from the fp32 oracle outside the recorded fragile set"
        assert all(frag_ok), f"{tag}: a fragile image differs from the exact-convolution oracle: {frag} {frag_ok}"
        assert int(eq.sum()) >= B - len(frag) and cs_eq == int(eq.sum())
    parity("f15_bench_job_tokens", **rep)


@pytest.mark.slow
def test_bench_line_job_full_size_vs_oracle(dev, ops):
    """The bench line's OWN job -- B = 256 x 100 reverse steps, Philox noise, one hipGraph replay, dense and elimination + lists --
    against the CPU oracle on the dumped noise: all 12 544 tokens (the other timed-form tests stop at
    B = 4 x 100 and B = 256 x 6).  Minutes of host work: SPKDIFF_RUN_SLOW=1, once per round through gpurun
    (tools/full_size_oracle.sh; log under profiles/).

    At this size -- 2.4e10 neuron-steps -- the oracle's own arithmetic shows: it convolves with oneDNN in fp32 (an accumulation
    order nobody controls), the HIP kernels return the correctly rounded EXACT dot product, and a handful of membrane potentials
    land within an ulp or two of the threshold, where the two decide a spike differently (SURVEY.md §7 "threshold discontinuity":
    the reference flips such spikes against its own fp64 evaluation).  Round 4 measured 2 of 12 544 tokens.  The test therefore
    does not stop at the count: every image whose tokens differ is traced to the FIRST reverse step at which the two
    trajectories part (samples are independent, the inputs of that step are identical on both sides), and there
      (1) the HIP logits equal the oracle's logits with every convolution evaluated exactly (fp64 sums, one rounding),
      (2) the fp32 oracle differs from that exact evaluation in at least one spike, and only at neuron-steps whose exact membrane
          potential is within 1e-5 of the threshold (its own fragile set),
      (3) the token update on the exact logits with the dumped noise gives the HIP tokens of that step.
    I.e. the HIP path is the exact arithmetic; the difference is the reference's rounding, not the kernels'."""
    import time
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.MNIST, dev)
    model, sd_v = build_vae(synth.MNIST, dev)
    B, steps, K = 256, 100, 128
    got, key = {}, None
    for name, skip in (("dense", False), ("elim+lists", True)):
        ab = AbsorbingDiffusion(den, mask_id=K)
        ab.n_samples, ab.skip_untouched, ab.list_positions = B, skip, skip
        torch.manual_seed(42)
        key = ab._philox_key()
        torch.manual_seed(42)
        got[name] = ab.sample(temp=1.0, sample_steps=steps)
        assert len(ab._graphs) == 1
    assert torch.equal(got["dense"], got["elim+lists"])
    # the same job step by step (eager launches, same Philox draws) with every step's state recorded
    ab = AbsorbingDiffusion(den, mask_id=K)
    ab.n_samples, ab.use_graph = B, False
    rec_h = []
    torch.manual_seed(42)
    tok_e = ab.sample(temp=1.0, sample_steps=steps, record=rec_h)
    assert torch.equal(tok_e, got["dense"]), "graph replay == eager launches"
    rec_h = [(t, x.cpu(), u.cpu(), lg.cpu()) for t, x, u, lg in rec_h]
    HW = 49
    noise_cache = {}

    def noise(t):
        u, q = ops.philox_noise(key, (steps - t) * (1 << 40), B, HW, K, dev)
        noise_cache[t] = (u.cpu().view(B, 1, 7, 7), q.cpu())
        return noise_cache[t]
    rec_o = []
    t0 = time.time()
    want = ref.absorbing_sample(sd, B, K, 1.0, steps, 7, 16, noise=noise, record=rec_o)
    dt = time.time() - t0
    bad = int((got["dense"].cpu() != want).sum())
    bad_images = sorted(set(torch.nonzero((got["dense"].cpu() != want).flatten(1).any(1)).flatten().tolist()))
    report = []
    for b in bad_images:
        k = next(i for i in range(steps) if not torch.equal(rec_h[i][1][b], rec_o[i][1][b]))
        t = rec_h[k][0]
        x_in = rec_h[k - 1][1][b:b + 1] if k > 0 else torch.full((1, 1, 7, 7), K, dtype=torch.int64)
        un_in = rec_h[k - 1][2][b:b + 1] if k > 0 else torch.zeros((1, 1, 7, 7), dtype=torch.bool)
        if k > 0:
            assert torch.equal(x_in, rec_o[k - 1][1][b:b + 1]), "identical inputs at the step where the trajectories part"
        tt = torch.full((1,), t, dtype=torch.long)
        with torch.inference_mode():
            lg32, lay32 = ref.denoiser_forward(x_in.float(), tt, sd, 16, return_layers=True)
            lgx, layx = ref.denoiser_forward(x_in.float(), tt, sd, 16, return_layers=True, exact_conv=True)
            if all(torch.equal(a[0], c[0]) for a, c in zip(lay32, layx)) and k > 0:
                # (oneDNN may block a batch of one differently from the batch of 256 the trajectory ran with: repeat the fp32
                #  evaluation on the whole batch of that step and take this image's slice)
                _, lay_all = ref.denoiser_forward(rec_o[k - 1][1].float(), torch.full((B,), t, dtype=torch.long), sd, 16,
                                                  return_layers=True)
                lay32 = [(a[:, b:b + 1].clone(), y[:, b:b + 1].clone()) for a, y in lay_all]
                del lay_all
        lg_hip = rec_h[k][3][b:b + 1]
        d_hip = float((lg_hip - lgx).abs().max())
        assert bool(((lg_hip - lgx).abs() <= 2e-6 * (1.0 + lgx.abs())).all()), f"image {b}: HIP logits != exact-convolution oracle"
        flips, worst = 0, 0.0
        for li, ((s32, _), (sx, yx)) in enumerate(zip(lay32, layx), 1):
            diff = s32 != sx
            if bool(diff.any()):
                v = torch.zeros_like(yx[0]); hs = []
                for ts in range(16):
                    h = v + (yx[ts] - v) * 0.5
                    hs.append(h); v = torch.where(h >= 1.0, torch.zeros_like(h), h)
                m = (torch.stack(hs) - 1.0).abs()[diff]
                first = int(torch.nonzero(diff.flatten(1).any(1)).min())           # earlier flips change later inputs: first step
                flips += int(diff.sum()); worst = max(worst, float((torch.stack(hs)[first] - 1.0).abs()[diff[first]].max()))
                assert float((torch.stack(hs)[first] - 1.0).abs()[diff[first]].max()) <= 1e-5, \
                    f"image {b} conv{li}: the fp32 oracle flips a spike that is NOT near the threshold"
                break                                                              # later layers see different inputs
        assert flips > 0, f"image {b}: tokens differ although the fp32 oracle and the exact evaluation agree on every spike"
        u, q = noise_cache[t]
        xs, _ = ref.p_sample_step(x_in, un_in, lgx.permute(0, 2, 3, 1), t, 1.0, u[b:b + 1], q[b * HW:(b + 1) * HW])
        assert torch.equal(xs, rec_h[k][1][b:b + 1]), f"image {b}: exact logits + dumped noise != HIP tokens of step {t}"
        report.append(dict(image=b, reverse_step=t, oracle_spike_flips_in_first_differing_layer=flips,
                           flipped_neuron_margin=worst, hip_vs_exact_logits=d_hip))
    # decode parity on the HIP tokens themselves (the oracle decodes the same tokens)
    pred, u8 = model.decode_tokens(got["dense"].reshape(B, 7, 7))
    opred = ref.decode_tokens(got["dense"].cpu().reshape(B, 7, 7), sd_v, 16)
    err = float((pred.cpu() - opred).abs().max())
    print(f"FULL SIZE B={B} x {steps} steps (Philox + hipGraph, dense == elimination + lists): {bad} of {want.numel()} tokens differ "
          f"from the fp32 CPU oracle, in {len(bad_images)} image(s); each traced to the oracle's own rounding: {report}; decode of the "
          f"HIP tokens vs oracle decode: max abs err {err:.2e}; oracle took {dt:.0f} s on {torch.get_num_threads()} threads")
    parity("bench_line_job_full_size_vs_oracle", token_mismatches_vs_fp32_oracle=bad, images_affected=len(bad_images),
           tokens=int(want.numel()), traced_to_oracle_rounding=report, decode_max_abs_err=err, oracle_seconds=dt)
    assert bad <= 16 and len(bad_images) <= 4, "a handful at most (2.4e10 neuron-steps; round 4 measured 2 tokens in 1-2 images)"
    assert err <= 1e-4


def test_philox_noise_entry_matches_what_psample_consumes(dev, ops):
    """spk_philox_noise == the draws of spk_psample_step / spk_select_active: a step run on injected (dumped) noise equals the
    same step in Philox mode, including through the philox_state indirection the captured graph uses."""
    B, K, HW = 64, 128, 49
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(B, K, 7, 7, generator=g) * 2).to(dev)
    un0 = (torch.rand(B, 1, 7, 7, generator=g) < 0.4).to(dev)
    x0 = torch.randint(0, K, (B, 1, 7, 7), generator=g).to(dev)
    state = torch.tensor([991, 4096], dtype=torch.int64, device=dev)
    for t in (1, 3, 50):
        off = 12345 * t
        u, q = ops.philox_noise(0, off, B, HW, K, dev, philox_state=state)
        u2, q2 = ops.philox_noise(991, off + 4096, B, HW, K, dev)
        assert torch.equal(u, u2) and torch.equal(q, q2)
        assert float(u.min()) >= 0 and float(u.max()) < 1 and float(q.min()) > 0
        xa, una = x0.clone(), un0.clone()
        ops.psample_step(logits, xa, una, t, 0.9, None, None, 0, off, philox_state=state)
        xb, unb = x0.clone(), un0.clone()
        ops.psample_step(logits, xb, unb, t, 0.9, u, q)
        assert torch.equal(xa, xb) and torch.equal(una, unb)
        a1 = ops.select_active(un0, t, None, 0, off, philox_state=state, K=K)
        a2 = ops.select_active(un0, t, u)
        n1, n2 = int(a1[1][0]), int(a2[1][0])
        assert n1 == n2 and torch.equal(a1[0][:n1], a2[0][:n2])


@pytest.mark.parametrize("steps", [12, 100])
def test_sampler_trajectory_8x8_vs_live_oracle(dev, steps):
    """BASELINE configs[3] shape: the reverse process on an 8x8 latent (R/snn_model/vq_diffusion.py:103-142 with the 7x7
    literals generalised), host noise in the reference's order, B = 4, dense and with the untouched-image elimination,
    against the CPU oracle run live under the same torch.manual_seed."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.CIFAR, dev)
    bad = {}
    torch.manual_seed(99 + steps)
    want = ref.absorbing_sample(sd, 4, 128, 1.0, steps, 8, 16)           # (one oracle run: both forms draw the same host noise)
    for name, skip in (("dense", False), ("elim", True)):
        ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(8, 8))
        ab.n_samples, ab.noise_source, ab.skip_untouched = 4, 'host', skip
        torch.manual_seed(99 + steps)
        tok = ab.sample(temp=1.0, sample_steps=steps).cpu()
        bad[name] = int((tok != want).sum())
    print(f"8x8 trajectory, {steps} steps, B=4: token mismatches {bad} of {want.numel()}")
    parity(f"trajectory_8x8_{steps}_steps_vs_live_oracle", token_mismatches=bad, tokens=int(want.numel()))
    assert all(v == 0 for v in bad.values()), bad


def test_timed_configuration_8x8_philox_graph_vs_oracle_on_dumped_noise(dev, ops):
    """The CIFAR-shaped sampler in the form bench.py times (Philox noise, hipGraph) against the oracle on the dumped noise."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.CIFAR, dev)
    B, steps = 4, 40
    got = {}
    for name, skip in (("dense", False), ("elim", True)):
        ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(8, 8))
        ab.n_samples, ab.skip_untouched = B, skip
        torch.manual_seed(31)
        key = ab._philox_key()
        torch.manual_seed(31)
        got[name] = ab.sample(temp=1.0, sample_steps=steps).cpu()
    want = _philox_oracle_tokens(ops, dev, sd, key, B, steps, 8)
    bad = {n: int((t != want).sum()) for n, t in got.items()}
    parity("timed_configuration_8x8_philox_graph", token_mismatches=bad, tokens=int(want.numel()))
    assert all(v == 0 for v in bad.values()), bad


def test_sample_does_not_depend_on_how_the_batch_is_split(dev, ops):
    """SURVEY.md §8e "parity mode => result independent of G" (the reference draws ONE batch from ONE stream,
    R/snn_model/vq_diffusion.py:103-142).  'global' noise layout: B = 32 generated as 1 x 32, 2 x 16 and 4 x 8 "virtual ranks"
    on one device (each a sampler with its own shard, graph-replayed, dense and elimination + lists) gives identical tokens and
    uint8 images, and equals the CPU oracle on the dumped noise; the 'rank' layout (rounds 1-3) is NOT split-invariant."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.MNIST, dev)
    model, _ = build_vae(synth.MNIST, dev)
    B, steps, seed = 32, 20, 2024

    def run(parts, skip, layout='global'):
        toks = []
        for r in range(parts):
            ab = AbsorbingDiffusion(den, mask_id=128)
            ab.noise_layout, ab.skip_untouched, ab.list_positions = layout, skip, skip
            ab.philox_stream = r
            ab.set_shard(r * (B // parts), B // parts)
            torch.manual_seed(seed)                       # every rank seeded alike
            toks.append(ab.sample(temp=1.0, sample_steps=steps))
            assert len(ab._graphs) == 1
        return torch.cat(toks, 0)
    ab0 = AbsorbingDiffusion(den, mask_id=128)
    torch.manual_seed(seed)
    key = ab0._philox_key()
    want = _philox_oracle_tokens(ops, dev, sd, key, B, steps, 7)
    bad = {}
    ref_u8 = None
    for parts in (1, 2, 4):
        for skip in (False, True):
            tok = run(parts, skip)
            bad[f"{parts}x{B // parts}{'_elim' if skip else ''}"] = int((tok.cpu() != want).sum())
            _, u8 = model.decode_tokens(tok.reshape(B, 7, 7))
            if ref_u8 is None:
                ref_u8 = u8
            assert torch.equal(u8, ref_u8)
    parity("sample_independent_of_split", token_mismatches_vs_oracle=bad, tokens=int(want.numel()))
    assert all(v == 0 for v in bad.values()), bad
    # a shard in the middle of a larger job, and a larger job's prefix: same images
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.set_shard(8, 8)
    torch.manual_seed(seed)
    assert torch.equal(ab.sample(temp=1.0, sample_steps=steps).cpu(), want[8:16])
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 48
    torch.manual_seed(seed)
    assert torch.equal(ab.sample(temp=1.0, sample_steps=steps).cpu()[:B], want)
    # the rank-folded layout of rounds 1-3 stays available -- and does depend on the split
    assert not torch.equal(run(2, False, 'rank'), run(1, False, 'rank'))


def test_rank_layout_timed_form_vs_oracle_on_dumped_noise(dev, ops):
    """The rounds 1-3 noise layout ('rank': local image index, step stride b*h*w*K, rank folded into the key) still pins to the
    oracle on its own dumped noise."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, sd = build_den(synth.MNIST, dev)
    B, steps = 8, 25
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.noise_layout, ab.philox_stream, ab.n_samples = 'rank', 3, B
    torch.manual_seed(5)
    key = ab._philox_key()
    torch.manual_seed(5)
    tok = ab.sample(temp=1.0, sample_steps=steps).cpu()
    want = _philox_oracle_tokens(ops, dev, sd, key, B, steps, 7, offset_of=lambda i: i * (B * 49 * 128))
    assert int((tok != want).sum()) == 0


def test_full_length_full_size_config3_and_config4_shapes(dev):
    """BASELINE configs[3] (CIFAR-shaped, B = 512) and the per-rank shape of configs[4] (B = 1024) at the benchmark's full
    length of 100 reverse steps: dense == elimination token for token (size-independent property of R/snn_model/vq_diffusion.py:
    113-124,140), every position unmasked, tokens in range, decode to uint8."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    for cfg, L, B in ((synth.CIFAR, 8, 512), (synth.MNIST, 7, 1024)):
        den, _ = build_den(cfg, dev)
        model, _ = build_vae(cfg, dev)
        toks = {}
        for name, skip in (("dense", False), ("elim", True)):
            ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(L, L))
            ab.n_samples, ab.skip_untouched = B, skip
            torch.manual_seed(8)
            toks[name] = ab.sample(temp=1.0, sample_steps=100)
            ab._graphs.clear()
        n_bad = int((toks["dense"] != toks["elim"]).sum())
        parity(f"full_length_B{B}_{L}x{L}", dense_vs_elimination_token_mismatches=n_bad, tokens=int(toks["dense"].numel()))
        assert n_bad == 0
        tok = toks["dense"]
        assert tok.shape == (B, 1, L, L) and int(tok.min()) >= 0 and int(tok.max()) < 128
        pred, u8 = model.decode_tokens(tok.reshape(B, L, L))
        assert u8.shape == (B, cfg.in_dim, 4 * L, 4 * L) and u8.dtype == torch.uint8 and float(pred.abs().max()) <= 1.0
        del den, model
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------- round 3: graph / cache lifetimes
def _eager_tokens(den, B, steps, seed, skip=True):
    from snn_model.vq_diffusion import AbsorbingDiffusion
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples, ab.use_graph, ab.skip_untouched = B, False, skip
    torch.manual_seed(seed)
    return ab.sample(temp=1.0, sample_steps=steps)


def test_sample_train_eval_sample_recaptures_the_graph(dev):
    """sample (graph) -> train() -> eval() -> sample: the transitions drop the derived tensors the captured launches address;
    the second sample must re-capture (derived epoch in the graph key) and equal the eager loop."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, _ = build_den(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 16
    torch.manual_seed(1); a = ab.sample(temp=1.0, sample_steps=20)
    k1 = set(ab._graphs)
    den.train(); den.eval()
    junk = [torch.randn(1 << 20, device=dev) for _ in range(8)]          # recycle whatever the invalidation freed
    torch.manual_seed(1); b = ab.sample(temp=1.0, sample_steps=20)
    assert set(ab._graphs) != k1, "a new graph was captured after the invalidation"
    assert torch.equal(a, b) and torch.equal(b, _eager_tokens(den, 16, 20, 1))
    del junk


def test_sample_after_graphed_training_steps_uses_the_new_weights(dev):
    """R/main.py:243-260 as this library runs it: sample -> N training iterations replayed from a hipGraph (no tensor's
    _version moves) -> eval -> sample.  The second sample must be the eager sampler's on the TRAINED weights."""
    from snn_model.vq_diffusion import AbsorbingDiffusion, functional
    from spkdiff.train import GraphedTrainStep
    den, _ = build_den(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 16
    torch.manual_seed(2); before = ab.sample(temp=1.0, sample_steps=20)
    den.train()
    opt = torch.optim.AdamW(den.parameters(), lr=5e-3, capturable=True)
    x0 = torch.randint(0, 128, (16, 1, 7, 7), generator=torch.Generator().manual_seed(3)).float().to(dev)
    step = GraphedTrainStep(ab, opt, x0)
    for _ in range(5):
        step(x0)
    den.eval()
    functional.reset_net(den)
    torch.manual_seed(2); after = ab.sample(temp=1.0, sample_steps=20)
    want = _eager_tokens(den, 16, 20, 2)
    assert torch.equal(after, want), "graph-replayed sampler == eager sampler on the trained weights"
    assert not torch.equal(after, before), "the training steps changed the weights enough to change the sample"


def test_data_write_is_noticed_by_the_weights_checksum(dev):
    """`p.data.copy_(...)` changes no (data_ptr, _version) pair; the sampler's content checksum must notice and rebuild."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    den, _ = build_den(synth.MNIST, dev)
    ab = AbsorbingDiffusion(den, mask_id=128)
    ab.n_samples = 16
    torch.manual_seed(4); a = ab.sample(temp=1.0, sample_steps=20)
    with torch.no_grad():
        for name, p in den.named_parameters():
            if name.endswith('0.weight'):
                p.data.mul_(1.5)
    torch.manual_seed(4); b = ab.sample(temp=1.0, sample_steps=20)
    den2, _ = build_den(synth.MNIST, dev)
    with torch.no_grad():
        for name, p in den2.named_parameters():
            if name.endswith('0.weight'):
                p.mul_(1.5)
    want = _eager_tokens(den2, 16, 20, 4)
    assert torch.equal(b, want) and not torch.equal(a, b)
    # ... also when the training path has re-laid the weights out channels-last (new storage, not default-contiguous)
    with torch.no_grad():
        for den_x in (den, den2):
            for name, p in den_x.named_parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
    torch.manual_seed(4); c = ab.sample(temp=1.0, sample_steps=20)
    assert torch.equal(c, b), "same values in another memory layout: same tokens"
    with torch.no_grad():
        for den_x in (den, den2):
            for name, p in den_x.named_parameters():
                if name.endswith('0.weight'):
                    p.data.mul_(0.8)
    torch.manual_seed(4); d = ab.sample(temp=1.0, sample_steps=20)
    assert torch.equal(d, _eager_tokens(den2, 16, 20, 4)) and not torch.equal(d, c)


def test_certified_kernel_workspaces_are_per_stream(dev, ops):
    """Two denoiser calls in flight on two streams must not share a flagged-neuron workspace (the repair pass of one would
    consume the other's ids): interleaved launches on two streams equal the serial results."""
    den, _ = build_den(synth.MNIST, dev)
    g = torch.Generator().manual_seed(6)
    xs = []
    for i in range(2):
        x = torch.randint(0, 128, (96, 1, 7, 7), generator=g)
        x[torch.rand(96, 1, 7, 7, generator=g) < 0.6] = 128
        xs.append(x.to(dev))
    want = [den.logits_from_tokens(x, 30 + i).clone() for i, x in enumerate(xs)]
    torch.cuda.synchronize()
    s = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    bad = 0
    for rep in range(6):
        got = [None, None]
        for i in range(2):
            with torch.cuda.stream(s[i]):
                got[i] = den.logits_from_tokens(xs[i], 30 + i)
        torch.cuda.synchronize()
        bad += sum(int((got[i] != want[i]).sum()) for i in range(2))
    keys = [k for k in ops._FLAG_DEFAULT if k[0] == "den"]
    assert len({k[2] for k in keys}) >= 3, "one workspace set per stream"
    assert bad == 0


# ------------------------------------------------------------------------------------------------- round 3: fused step tail
@pytest.mark.parametrize("cfg,L,B", [(synth.MNIST, 7, 37), (synth.CIFAR, 8, 19)])
def test_step_tail_equals_the_three_launches_it_replaces(dev, ops, cfg, L, B):
    """spk_den_step_tail == spk_den_conv3x3_counts_mfma + spk_psample_step + the first layer's spk_conv_fused_fwd, bit for bit:
    logits, tokens, unmasked, the next step's conv1 spikes (S32) and spike counts; Philox and injected noise; last step (no
    successor)."""
    den, _ = build_den(cfg, dev)
    g = torch.Generator().manual_seed(17)
    HW, K = L * L, 128
    for trial, t in enumerate((57, 3, 1)):
        x0 = torch.randint(0, K, (B, 1, L, L), generator=g)
        un0 = torch.rand(B, 1, L, L, generator=g) < 0.5
        x0[~un0] = K
        x0, un0 = x0.to(dev), un0.to(dev)
        inp = ops.den_build_input(x0, t)
        x5, cnt5, x1, cnt1, which, impl, collapse = den._trunk(inp, False)
        assert which == 'mfma-fp6v2' and collapse
        conv6, packed6 = den._conv6_params()
        logits = ops.den_conv3x3_counts(cnt5, packed6, K, 16, cnt1=cnt1)
        inject = trial == 1
        u = q = None
        if inject:
            u = torch.rand(B * HW, generator=g).to(dev)
            q = torch.empty(B * HW, K).exponential_(1, generator=g).to(dev)
        xa, una = x0.clone(), un0.clone()
        nxt = torch.empty((B, 2, L, L), dtype=torch.float32, device=dev)
        ops.psample_step(logits, xa, una, t, 0.9, u, q, seed=4242, offset=1000 * t, next_input=nxt if t > 1 else None)
        xb, unb = x0.clone(), un0.clone()
        conv1, bn1 = den.conv1[0], den.conv1[1]
        a1, b1 = bn1.affine_terms()
        c1 = (conv1._spk_params.get(conv1), conv1.bias.detach(), a1, b1) if t > 1 else None
        pre, lg = ops.den_step_tail(cnt5, cnt1, packed6, xb, unb, t, 0.9, T=16, K=K, u=u, q=q, seed=4242, offset=1000 * t,
                                    conv1=c1, want_logits=True)
        assert torch.equal(lg, logits), float((lg - logits).abs().max())
        assert torch.equal(xa, xb) and torch.equal(una, unb)
        assert int(unb.sum()) > int(un0.sum()) or t > 20
        if t > 1:
            r1 = den.conv1.run(nxt, ops.IN_TINV, final='ptc', T=16, stateful=False, chunk_out=ops.CHUNK_S32, want_counts=True)
            assert torch.equal(pre[0], r1['ptc']) and torch.equal(pre[1], r1['cnt'])
        else:
            assert pre is None


@pytest.mark.parametrize("K,L", [(1, 7), (17, 7), (129, 8), (385, 7), (512, 8)])
def test_step_tail_for_other_codebook_sizes_equals_the_three_launches(dev, ops, K, L):
    """Round 6: spk_den_step_tail takes 1 <= K <= 512 classes (one to four 16-channel groups per wave; conv6's output channels zero-padded
    to a multiple of 16 in the packed weights; classes >= K masked as spk_psample_step masks them; R/main.py:58 --codebook_size).  Edge sizes
    -- one class, one class past a group, one past 128 / 384 (the next channel-group count per wave), the maximum -- against counts-conv6 +
    spk_psample_step + the first layer's launch: logits, tokens, unmasked, the next step's conv1 spikes and counts; Philox and injected noise."""
    from snn_model.vq_diffusion import DummyModel, functional
    torch.manual_seed(K)
    den = DummyModel(1, K).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    with torch.no_grad():
        den.conv6[0].weight.mul_(20.0)                      # (random init: widen the logits so that the classes compete)
    den.eval()
    assert den.tail_fusable(L, L) and den.impl_for(L, L) == 'mfma-fp6v2'
    g = torch.Generator().manual_seed(100 + K)
    B, HW = 5, L * L
    bad = 0
    for trial, t in enumerate((41, 2, 1)):
        x0 = torch.randint(0, K, (B, 1, L, L), generator=g)
        un0 = torch.rand(B, 1, L, L, generator=g) < 0.5
        x0[~un0] = K
        x0, un0 = x0.to(dev), un0.to(dev)
        x5, cnt5, x1, cnt1, which, impl, collapse = den._trunk(ops.den_build_input(x0, t), False)
        conv6, packed6 = den._conv6_params()
        logits = ops.den_conv3x3_counts(cnt5, packed6, K, 16, cnt1=cnt1)
        assert logits.shape == (B, K, L, L)
        u = q = None
        if trial == 1:
            u = torch.rand(B * HW, generator=g).to(dev)
            q = torch.empty(B * HW, K).exponential_(1, generator=g).to(dev)
        xa, una = x0.clone(), un0.clone()
        nxt = torch.empty((B, 2, L, L), dtype=torch.float32, device=dev)
        ops.psample_step(logits, xa, una, t, 0.9, u, q, seed=99, offset=777 * t, next_input=nxt if t > 1 else None)
        xb, unb = x0.clone(), un0.clone()
        conv1, bn1 = den.conv1[0], den.conv1[1]
        a1, b1 = bn1.affine_terms()
        c1 = (conv1._spk_params.get(conv1), conv1.bias.detach(), a1, b1) if t > 1 else None
        pre, lg = ops.den_step_tail(cnt5, cnt1, packed6, xb, unb, t, 0.9, T=16, K=K, u=u, q=q, seed=99, offset=777 * t, conv1=c1,
                                    want_logits=True)
        bad += int(not torch.equal(lg, logits)) + int(not torch.equal(xa, xb)) + int(not torch.equal(una, unb))
        assert int(xb.max()) <= K and int(xb[unb].max()) < K
        if t > 1:
            r1 = den.conv1.run(nxt, ops.IN_TINV, final='ptc', T=16, stateful=False, chunk_out=ops.CHUNK_S32, want_counts=True)
            bad += int(not torch.equal(pre[0], r1['ptc'])) + int(not torch.equal(pre[1], r1['cnt']))
        else:
            assert pre is None and bool(unb.all())
    parity(f"step_tail_K{K}_{L}x{L}", differing_outputs=bad)
    assert bad == 0


@pytest.mark.parametrize("cfgname,L,B,t", [("mnist", 7, 37, 30), ("mnist", 7, 5, 3), ("cifar", 8, 19, 12)])
def test_step_tail_on_the_active_list_equals_the_two_launches(dev, ops, cfgname, L, B, t):
    """Round 6, elimination forms: spk_den_step_tail with the active list (workgroup = slot; counts and logits per slot, tokens / unmasked /
    noise per image active[s]) == spk_den_conv3x3_counts_mfma + spk_psample_step on the same list: logits, tokens, unmasked; images that are
    not on the list stay untouched; Philox and injected noise (R/snn_model/vq_diffusion.py:113-140 for the images that change at step t)."""
    cfg = synth.MNIST if cfgname == "mnist" else synth.CIFAR
    den, _ = build_den(cfg, dev)
    g = torch.Generator().manual_seed(55 + B)
    HW, K = L * L, 128
    for trial in range(2):
        x0 = torch.randint(0, K, (B, 1, L, L), generator=g)
        un0 = torch.rand(B, 1, L, L, generator=g) < 0.6
        un0[1 % B] = True                                   # (an image with nothing left to unmask: never on the list)
        x0[~un0] = K
        x0, un0 = x0.to(dev), un0.to(dev)
        inject = trial == 1
        u = torch.rand(B * HW, generator=g).to(dev) if inject else None
        q = torch.empty(B * HW, K).exponential_(1, generator=g).to(dev) if inject else None
        act = ops.select_active(un0, t, u.view(B, 1, L, L) if inject else None, 31, 500 * t, K=K)
        n = int(act[1][0].item())
        assert 0 < n < B
        out = {}
        for form in ("two_launches", "step_tail"):
            xa, una = x0.clone(), un0.clone()
            with ops.active_set(*act):
                x5, cnt5, x1, cnt1, which, impl, collapse = den._trunk(ops.den_build_input(xa, t), False)
                conv6, packed6 = den._conv6_params()
                if form == "two_launches":
                    lg = ops.den_conv3x3_counts(cnt5, packed6, K, 16, cnt1=cnt1)
                    ops.psample_step(lg, xa, una, t, 0.9, u, q, seed=31, offset=500 * t)
                else:
                    pre, lg = ops.den_step_tail(cnt5, cnt1, packed6, xa, una, t, 0.9, T=16, K=K, u=u, q=q, seed=31, offset=500 * t,
                                                conv1=None, want_logits=True)
                    assert pre is None
            out[form] = (lg[:n].clone(), xa, una)
        assert torch.equal(out["step_tail"][0], out["two_launches"][0])
        assert torch.equal(out["step_tail"][1], out["two_launches"][1]) and torch.equal(out["step_tail"][2], out["two_launches"][2])
        listed = torch.zeros(B, dtype=torch.bool, device=dev)
        listed[act[0][:n].long()] = True
        assert torch.equal(out["step_tail"][1][~listed], x0[~listed]) and torch.equal(out["step_tail"][2][~listed], un0[~listed])
        assert int(out["step_tail"][2].sum()) > int(un0.sum())
    parity(f"step_tail_active_list_{cfgname}_B{B}", images_on_the_list=n, differing=0)


@pytest.mark.parametrize("B", [1, 6])
def test_fp6v2_small_batch_split_with_no_active_image(dev, ops, B):
    """The small-batch split under a device-side image count of ZERO (a reverse step in which no image of a small batch changes): nothing
    is computed, the launch terminates, the workspace comes back clean, and the next call on it is right."""
    g = torch.Generator().manual_seed(8200 + B)
    Cout, Cin = 256, 128
    w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05).to(dev)
    bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.2).to(dev)
    a = ((torch.rand(Cout, generator=g) - 0.3) * 8.0).to(dev)
    b = ((torch.rand(Cout, generator=g) - 0.4) * 1.5).to(dev)
    sd = (torch.rand(16, B, Cin, 7, 7, generator=g) < 0.08).float().to(dev)
    pk, xs = ops.den_pack_weight_fp6v2(w, bias), ops.spikes_to_s32(sd)
    want, cw = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
    active = torch.arange(B, dtype=torch.int32, device=dev)
    with ops.active_set(active, torch.zeros(2, dtype=torch.int32, device=dev)):
        ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
    torch.cuda.synchronize()
    assert all(flag_ws_clean(v) for k, v in ops._FLAG_DEFAULT.items() if k[0] == "den")
    got, cg = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
    assert torch.equal(got, want) and torch.equal(cg, cw)


@pytest.mark.parametrize("cfgname,L", [("mnist", 7), ("cifar", 8)])
def test_sampler_with_and_without_the_fused_step_tail(dev, cfgname, L):
    """Dense sampling with the fused tail launch == with the three separate launches: graph replay, eager launches, host noise."""
    from snn_model.vq_diffusion import AbsorbingDiffusion
    cfg = synth.MNIST if cfgname == "mnist" else synth.CIFAR
    den, _ = build_den(cfg, dev)
    out = {}
    for tail in (True, False):
        den.use_step_tail = tail
        for mode in ("graph", "eager", "host"):
            ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(L, L))
            ab.n_samples, ab.skip_untouched = 21, False
            ab.use_graph = mode == "graph"
            ab.noise_source = 'host' if mode == "host" else 'philox'
            torch.manual_seed(5)
            out[(tail, mode)] = ab.sample(temp=1.0, sample_steps=30).cpu()
            if mode == "graph":
                assert len(ab._graphs) == 1
    den.use_step_tail = True
    for mode in ("graph", "eager", "host"):
        assert torch.equal(out[(True, mode)], out[(False, mode)]), mode
    assert torch.equal(out[(True, "graph")], out[(True, "eager")])
    rec = []
    ab = AbsorbingDiffusion(den, mask_id=128, latent_shape=(L, L))
    ab.n_samples, ab.noise_source = 5, 'host'
    torch.manual_seed(6)
    tok = ab.sample(temp=1.0, sample_steps=8, record=rec)
    assert len(rec) == 8 and rec[0][3].shape == (5, 128, L, L) and torch.equal(rec[-1][1], tok)
    lg = den.logits_from_tokens(rec[3][1], rec[4][0])                    # the logits recorded at a step == a plain denoiser call
    assert torch.equal(lg, rec[4][3])


# ------------------------------------------------------------------------------------------------- round 3: the other LIF forms
LIF_FORMS = [("soft_decay", dict(v_reset=None, decay_input=True, tau=2.0)),
             ("soft_nodecay", dict(v_reset=None, decay_input=False, tau=2.0)),
             ("hard_nodecay", dict(v_reset=0.0, decay_input=False, tau=2.0)),
             ("hard_decay_vseq", dict(v_reset=0.0, decay_input=True, tau=2.0)),
             ("soft_decay_tau3", dict(v_reset=None, decay_input=True, tau=3.0)),
             ("soft_nodecay_tau3", dict(v_reset=None, decay_input=False, tau=3.0)),
             ("hard_nodecay_tau5_vr", dict(v_reset=-0.25, decay_input=False, tau=5.0, v_threshold=0.8))]


@pytest.mark.parametrize("name,kw", LIF_FORMS)
def test_f14_lifnode_other_eval_forms_vs_reference_fixture(golden_dir, dev, name, kw):
    """neuron.LIFNode with soft reset (v_reset=None), decay_input=False and store_v_seq (SJ/activation_based/neuron.py:813-900,
    971-1011) on the HIP kernel spk_lif_fwd_ex against the REAL reference's outputs (fixture F14): spikes, v_seq and the
    state carried into a second call, bit for bit; and without store_v_seq the same spikes."""
    from spikingjelly.activation_based import neuron, surrogate
    d = load(golden_dir, "f14_lif_forms.npz")
    x = torch.from_numpy(d["x_seq"]).to(dev)
    node = neuron.LIFNode(surrogate_function=surrogate.ATan(), step_mode="m", store_v_seq=True, **kw).eval()
    with torch.inference_mode():
        s = node(x); vs = node.v_seq.clone()
        s2 = node(x.flip(0)); vs2 = node.v_seq.clone()
    bad = (int((s.cpu() != unpack(d[name + "_spikes"], d["spikes_shape"])).sum()) +
           int((s2.cpu() != unpack(d[name + "_spikes_carry"], d["spikes_shape"])).sum()))
    vbad = int((vs.cpu() != torch.from_numpy(d[name + "_v_seq"])).sum()) + int((vs2.cpu() != torch.from_numpy(d[name + "_v_seq_carry"])).sum())
    parity(f"f14_lif_{name}", spike_mismatches=bad, v_seq_mismatches=vbad, values=int(2 * s.numel()))
    assert bad == 0 and vbad == 0
    node.reset()
    assert node.v_seq is None and isinstance(node.v, float)
    node2 = neuron.LIFNode(surrogate_function=surrogate.ATan(), step_mode="m", **kw).eval()
    with torch.inference_mode():
        assert torch.equal(node2(x), s)
    # ragged size / other shape against the live oracle
    g = torch.Generator().manual_seed(9)
    xr = torch.randn(7, 3, 5, 11, generator=g) * 1.5
    node3 = neuron.LIFNode(surrogate_function=surrogate.ATan(), step_mode="m", store_v_seq=True, **kw).eval()
    with torch.inference_mode():
        s3 = node3(xr.to(dev))
    o3, ov3, ovs3 = ref.lif_multi_step_ex(xr, 0.0 if kw["v_reset"] is None else kw["v_reset"], kw.get("v_threshold", 1.0),
                                          kw["v_reset"], kw["tau"], kw["decay_input"])
    assert torch.equal(s3.cpu(), o3) and torch.equal(node3.v_seq.cpu(), ovs3) and torch.equal(node3.v.cpu(), ov3)


# ------------------------------------------------------------------------------------------------- round 3: native weight gradient
@pytest.mark.parametrize("N,Cout,Cin", [(7, 128, 64), (64, 256, 128), (37, 512, 256), (16, 256, 512), (5, 128, 320)])
@pytest.mark.parametrize("HH", [7, 8])
def test_conv3x3_weight_gradient_bf16_kernel_vs_fp64(dev, ops, N, Cout, Cin, HH):
    """spk_conv3x3_wgrad_bf16 (spike operand exact in bf16, gy split into three bf16 terms exactly, fp32 accumulation on the
    matrix cores) against the fp64 weight gradient of the same convolution: relative L2 error at fp32 round-off, and not worse
    than the framework's fp32 operator.  7x7 (MNIST-shaped latents) and 8x8 maps (CIFAR-shaped: round 4)."""
    g = torch.Generator().manual_seed(N + Cout)
    s = (torch.rand(N, Cin, HH, HH, generator=g) < 0.07).float()
    gy = torch.randn(N, Cout, HH, HH, generator=g) * torch.rand(Cout, generator=g).view(1, -1, 1, 1) * 1e-3
    gy[:, ::5] *= 64.0                                                    # mixed magnitudes across channels
    w = torch.zeros(Cout, Cin, 3, 3)
    _, want, _ = torch.ops.aten.convolution_backward(gy.double(), s.double(), w.double(), [Cout], [1, 1], [1, 1], [1, 1], False,
                                                     [0, 0], 1, [False, True, False])
    s_cl = s.to(dev).contiguous(memory_format=torch.channels_last)
    gy_cl = gy.to(dev).contiguous(memory_format=torch.channels_last)
    got, got_b = ops.conv3x3_wgrad(gy_cl, s_cl, Cout, Cin, want_bias=True)
    assert got.shape == (Cout, Cin, 3, 3)
    want_b = gy.double().sum(dim=(0, 2, 3))
    rel_b = float((got_b.cpu().double() - want_b).norm() / want_b.norm())
    assert rel_b <= 1e-6, rel_b                                          # the bias gradient from the same pass over gy
    rel = float((got.cpu().double() - want).norm() / want.norm())
    _, lib_gw, _ = torch.ops.aten.convolution_backward(gy_cl, s_cl, w.to(dev), [Cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])
    rel_lib = float((lib_gw.cpu().double() - want).norm() / want.norm())
    print(f"wgrad {HH}x{HH} N={N} {Cin}->{Cout}: rel L2 error {rel:.2e} (framework operator {rel_lib:.2e})")
    parity(f"wgrad_bf16_{HH}x{HH}_N{N}_{Cin}_{Cout}", rel_l2_err=rel, rel_l2_err_framework=rel_lib)
    assert rel <= 2e-6
    got2 = ops.conv3x3_wgrad(gy_cl, s_cl, Cout, Cin)
    assert torch.equal(got, got2), "deterministic (partial sums added in a fixed order)"


@pytest.mark.parametrize("HH", [7, 8])
def test_conv3x3_weight_gradient_impulses(dev, ops, HH):
    """One spike and one unit output gradient at a time: gw[co, ci, ky, kx] = 1 exactly where (y_s, x_s) = (y_g + ky - 1, x_g + kx - 1)
    and 0 elsewhere -- every (position, tap) pair of the map, every LDS row / column slot of the kernel's shifted spike copies
    and (8x8) swizzled gradient rows, channels on both sides of the swizzle periods."""
    Cout, Cin = 128, 64
    cases = [(yg, xg, co, ci) for yg, xg, co, ci in
             [(0, 0, 0, 0), (HH - 1, HH - 1, 127, 63), (0, HH - 1, 3, 9), (HH - 1, 0, 66, 40), (3, 4, 17, 8), (HH - 2, 1, 2, 15),
              (1, HH - 2, 5, 56), (4, 3, 127, 24)]]
    for yg, xg, co, ci in cases:
        for ys in range(max(0, yg - 1), min(HH, yg + 2)):
            for xs in range(max(0, xg - 1), min(HH, xg + 2)):
                s = torch.zeros(2, Cin, HH, HH); gy = torch.zeros(2, Cout, HH, HH)
                s[1, ci, ys, xs] = 1.0; gy[1, co, yg, xg] = 1.0
                got = ops.conv3x3_wgrad(gy.to(dev).contiguous(memory_format=torch.channels_last),
                                        s.to(dev).contiguous(memory_format=torch.channels_last), Cout, Cin).cpu()
                want = torch.zeros(Cout, Cin, 3, 3)
                want[co, ci, ys - yg + 1, xs - xg + 1] = 1.0
                assert torch.equal(got, want), (HH, yg, xg, ys, xs, co, ci, got.nonzero().tolist())


# ------------------------------------------------------------------------------------------------- round 3: native data gradient
@pytest.mark.parametrize("N,Cout,Cin", [(8, 128, 64), (64, 256, 128), (37, 512, 256), (16, 256, 512), (5, 128, 320), (320, 128, 320),
                                        (3, 16, 32)])
@pytest.mark.parametrize("form", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("HH", [7, 8])
def test_conv3x3_data_gradient_bf16_kernel_vs_fp64(dev, ops, N, Cout, Cin, form, HH):
    """spk_conv3x3_dgrad_bf16 (both operands split into three bf16 terms exactly, six cross products on the matrix cores, fp32
    accumulation) against the fp64 data gradient of the same convolution: relative L2 error at fp32 round-off, and not worse
    than twice the framework's fp32 operator; borders, ragged image groups, both column-tile forms."""
    g = torch.Generator().manual_seed(N + Cout + Cin)
    gy = torch.randn(N, Cout, HH, HH, generator=g) * torch.rand(Cout, generator=g).view(1, -1, 1, 1) * 1e-3
    gy[:, ::5] *= 64.0                                                    # mixed magnitudes across channels
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (0.5 / (Cin * 9) ** 0.5)
    w[::3] *= 17.0
    x = torch.zeros(N, Cin, HH, HH)
    want, _, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], [1, 1], [1, 1], [1, 1], False,
                                                     [0, 0], 1, [True, False, False])
    gy_cl = gy.to(dev).contiguous(memory_format=torch.channels_last)
    got = ops.conv3x3_dgrad(gy_cl, w.to(dev), Cin, form=form)
    assert got.shape == (N, Cin, HH, HH)
    rel = float((got.cpu().double() - want).norm() / want.norm())
    lib_gi, _, _ = torch.ops.aten.convolution_backward(gy_cl, x.to(dev).contiguous(memory_format=torch.channels_last),
                                                       w.to(dev).contiguous(memory_format=torch.channels_last), [Cout], [1, 1],
                                                       [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])
    rel_lib = float((lib_gi.cpu().double() - want).norm() / want.norm())
    worst = float((got.cpu().double() - want).abs().max() / want.abs().max())
    print(f"dgrad {form} {HH}x{HH} N={N} {Cout}->{Cin}: rel L2 error {rel:.2e} (framework operator {rel_lib:.2e}), max abs / max |gi| {worst:.2e}")
    parity(f"dgrad_{form}_{HH}x{HH}_N{N}_{Cout}_{Cin}", rel_l2_err=rel, rel_l2_err_framework=rel_lib)
    assert rel <= 1e-6 and rel <= 2.0 * rel_lib + 1e-7 and worst <= 2e-6
    got2 = ops.conv3x3_dgrad(gy_cl, w.to(dev), Cin, form=form)
    assert torch.equal(got, got2), "deterministic"


def test_data_gradient_forms_wide_dynamic_range_per_element(dev, ops):
    """whole-tensor relative L2 cannot see what the two-term fp16 form does to small elements.  One image's gy spans
    2^-34 .. 1 (rows of the map scaled by 2^-(6 y)), so the outputs of the lower rows are sums of terms far below the image's
    maximum.  Per ELEMENT, against fp64 and relative to the sum of |terms| of that output (what an fp32 operator's error is
    relative to): the three-term bf16 form (exact truncations at any magnitude) stays at fp32 round-off everywhere; the two-term
    fp16 form (one power-of-two scale per image / per input channel, default of the training backward) does so only within ~17
    binades of the image's maximum and keeps an ABSOLUTE 2^-38 of (image maximum x channel weight sum) below -- which is what
    csrc/conv_dgrad.hip and ops.conv3x3_dgrad now say."""
    N, Cout, Cin, HH = 64, 64, 64, 7
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(N, Cout, HH, HH, generator=g) * torch.exp2(-6.0 * torch.arange(HH).float()).view(1, 1, HH, 1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    x = torch.zeros(N, Cin, HH, HH)
    want, _, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], [1, 1], [1, 1], [1, 1], False,
                                                     [0, 0], 1, [True, False, False])
    mag, _, _ = torch.ops.aten.convolution_backward(gy.double().abs(), x.double(), w.double().abs(), [Cout], [1, 1], [1, 1], [1, 1],
                                                    False, [0, 0], 1, [True, False, False])          # sum of |terms| per output
    gy_cl = gy.to(dev).contiguous(memory_format=torch.channels_last)
    rep = {}
    for form in ("bf16x3", "f16x2"):
        got = ops.conv3x3_dgrad(gy_cl, w.to(dev), Cin, form=form).cpu().double()
        per_elem = ((got - want).abs() / mag.clamp_min(1e-300))                                      # [N, Cin, 7, 7]
        by_row = per_elem.amax(dim=(0, 1, 3))                                                        # worst per map row y
        rep[form] = [float(v) for v in by_row]
    print("per-element error / sum|terms| by map row (row y carries gy ~ 2^-(6 y)):", {k: [f"{v:.1e}" for v in r] for k, r in rep.items()})
    parity("dgrad_forms_wide_dynamic_range_per_element", **rep)
    assert max(rep["bf16x3"]) <= 1e-6, "three exact bf16 terms: fp32 round-off per element at every magnitude"
    assert max(rep["f16x2"][:3]) <= 1e-6, "two scaled fp16 terms: fp32 round-off within ~17 binades of the image's maximum"
    # below that the error is absolute: 2^-40 of the image's maximum per gy term (the remainder term has gone subnormal)
    wsum = float(w.abs().sum(dim=(0, 2, 3)).max())
    amax = float(gy.abs().amax())
    got = ops.conv3x3_dgrad(gy_cl, w.to(dev), Cin, form="f16x2").cpu().double()
    assert float((got - want).abs().max()) <= 2.0 ** -38 * amax * wsum + 1e-6 * float(mag.max())


def test_last_layer_and_time_mean_as_one_operator_backward_on_spike_counts(dev, ops):
    """ops.SpikeConvMeanTrainFunction (conv6 + mean over T; backward = one convolution of g / T with the spike COUNTS and one
    transposed convolution repeated over T) against the two-operator form (per-step convolution, sum / T, per-step backward):
    forward bit-equal, gradients equal to fp32 round-off."""
    T, B, Cin, Cout = 16, 6, 320, 128
    g = torch.Generator().manual_seed(11)
    s = (torch.rand(T, B, Cin, 7, 7, generator=g) < 0.08).float().to(dev)
    w0 = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=torch.channels_last)
    b0 = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    go = torch.randn(B, Cout, 7, 7, generator=g).to(dev)
    outs = []
    for fused in (True, False):
        sx = s.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        if fused:
            y = ops.SpikeConvMeanTrainFunction.apply(sx, w, b)
        else:
            y = torch.sum(ops.SpikeConvTrainFunction.apply(sx, w, b), dim=0) / T
        y.backward(go)
        outs.append((y.detach(), sx.grad.detach().clone(), w.grad.detach().clone(), b.grad.detach().clone()))
    (y1, gs1, gw1, gb1), (y2, gs2, gw2, gb2) = outs
    assert torch.equal(y1, y2)
    rel = lambda a, c: float((a.double() - c.double()).norm() / c.double().norm())
    r_s, r_w, r_b = rel(gs1, gs2), rel(gw1, gw2), rel(gb1, gb2)
    print(f"collapsed backward vs per-step backward: grad_spikes {r_s:.2e}, grad_weight {r_w:.2e}, grad_bias {r_b:.2e}")
    parity("conv6_mean_one_operator", grad_spikes_rel=r_s, grad_weight_rel=r_w, grad_bias_rel=r_b)
    assert gs1.shape == s.shape and r_s <= 1e-6 and r_w <= 1e-6 and r_b <= 1e-6


def test_conv3x3_data_gradient_two_term_form_scales(dev, ops):
    """spk_conv3x3_dgrad_f16x2 scales every image's gy and every input channel's weights by a power of two: images / channels
    of very different magnitude keep their own relative accuracy, all-zero images / channels give exact zeros, and values far
    below their image's maximum cost absolute, not relative, precision."""
    N, Cout, Cin = 8, 128, 64
    g = torch.Generator().manual_seed(5)
    gy = torch.randn(N, Cout, 7, 7, generator=g) * 1e-3
    gy[0] = 0.0
    gy[1] *= 1e-20
    gy[2] *= 1e+12
    gy[3, ::2] *= 2.0 ** -30                                               # half of the channels 30 binades below the others
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    w[:, 5] = 0.0
    w[:, 7] *= 1e-12
    w[:, 9] *= 1e+6
    x = torch.zeros(N, Cin, 7, 7)
    want, _, _ = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], [1, 1], [1, 1], [1, 1], False,
                                                     [0, 0], 1, [True, False, False])
    got = ops.conv3x3_dgrad(gy.to(dev).contiguous(memory_format=torch.channels_last), w.to(dev), Cin, form="f16x2").cpu().double()
    assert torch.isfinite(got).all()
    assert float(got[0].abs().max()) == 0.0 and float(got[:, 5].abs().max()) == 0.0
    worst = 0.0
    for n in range(1, N):
        for ci in (0, 7, 9, 33):
            e = float((got[n, ci] - want[n, ci]).norm() / want[n, ci].norm())
            worst = max(worst, e)
    print(f"two-term data gradient, per (image, channel) relative L2 error at most {worst:.2e}")
    parity("dgrad_f16x2_scales", worst_image_channel_rel_l2=worst)
    assert worst <= 2e-6



# ------------------------------------------------------------------------------------------------- round 4: training-step glue made native
@pytest.mark.parametrize("N,Cin,Cout,HH,cl", [(512, 2, 64, 7, True), (37, 3, 128, 8, False), (5, 1, 96, 7, True)])
def test_small_input_weight_gradient_kernel_vs_fp64(dev, ops, N, Cin, Cout, HH, cl):
    """spk_conv3x3_wgrad_small (the denoiser's first layer in the training step: two input channels, dense input) against the fp64
    weight / bias gradient of the same convolution; both weight memory formats; deterministic."""
    g = torch.Generator().manual_seed(N + Cout)
    gy = torch.randn(N, Cout, HH, HH, generator=g) * 1e-2
    x = torch.randn(N, Cin, HH, HH, generator=g) * 3.0
    w = torch.zeros(Cout, Cin, 3, 3)
    _, gw64, gb64 = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), [Cout], [1, 1], [1, 1], [1, 1], False,
                                                        [0, 0], 1, [False, True, True])
    wd = w.to(dev).contiguous(memory_format=torch.channels_last) if cl else w.to(dev)
    gw, gb = ops.conv3x3_wgrad_small(gy.to(dev), x.to(dev), wd, True)
    assert gw.shape == w.shape and (gw.is_contiguous(memory_format=torch.channels_last) if cl and Cin > 1 else True)
    rel = float((gw.cpu().double() - gw64).norm() / gw64.norm())
    relb = float((gb.cpu().double() - gb64).norm() / gb64.norm())
    parity(f"wgrad_small_N{N}_{Cin}_{Cout}", rel_l2_err=rel, bias_rel_l2_err=relb)
    assert rel <= 3e-7 and relb <= 3e-7
    gw2, gb2 = ops.conv3x3_wgrad_small(gy.to(dev), x.to(dev), wd, True)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_bn_lif_backward_reads_broadcast_and_pitched_gradients(dev, ops):
    """spk_bn_lif_train_bwd_strided: the gradient of the spikes in front of the denoiser's last layer arrives as ONE [B,C,H,W]
    tensor broadcast over T and as a channel slice of cat(x5, x1)'s wider gradient; reading it in place gives, bit for bit, what
    the expanded dense copy gives."""
    T, B, C, H, W, Cw = 16, 6, 64, 7, 7, 96
    g = torch.Generator().manual_seed(3)
    y = torch.randn(T, B, C, H, W, generator=g).to(dev)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    wide = torch.randn(B, Cw, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    outs = []
    for mode in ("view", "dense"):
        yy = y.clone().requires_grad_(True)
        ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        s, v_last = ops.BNLIFTrainFunction.apply(yy, ga, be, None, rm, rv, 0.1, 1e-5, 2.0, 1.0, 0.0, 2.0, False)
        gs = wide[:, 16:16 + C].unsqueeze(0).expand(T, B, C, H, W)          # stride 0 over T, row pitch 96 > 64
        assert gs.stride(0) == 0 and gs.stride(4) == Cw
        if mode == "dense":
            gs = gs.contiguous()
        s.backward(gs)
        outs.append((yy.grad.clone(), ga.grad.clone(), be.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fp6_weight_packing_reads_channels_last_weights(dev, ops):
    """spk_den_pack_weight_fp6_cl == spk_den_pack_weight_fp6 on the same values (the training path keeps its weights channels-last);
    the packing pass that also counts spikes == the plain one + sum over T."""
    g = torch.Generator().manual_seed(9)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.randn(128, generator=g).to(dev)
    p0 = ops.den_pack_weight_fp6(w, b)
    p1 = ops.den_pack_weight_fp6(w.contiguous(memory_format=torch.channels_last), b)
    assert torch.equal(p0[1], p1[1]) and torch.equal(p0[2], p1[2])             # scales, biases (the slabs carry unwritten padding)
    s = (torch.rand(16, 5, 64, 7, 7, generator=g) < 0.07).float().to(dev)
    s = s.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)            # channels-last memory
    c4, cnt = ops.spikes_cl_to_c4_counts(s)
    assert torch.equal(c4, ops.spikes_cl_to_c4(s)) and torch.equal(cnt, s.sum(0))
    y0, y1 = ops.den_conv3x3_fp6_raw(c4, p0, 128), ops.den_conv3x3_fp6_raw(c4, p1, 128)
    assert torch.equal(y0, y1) and float(y0.abs().max()) > 0


def test_training_weight_prep_is_the_per_layer_packing(dev, ops):
    """ops.train_weight_prep (spk_den_pack_weight_fp6_cl_multi + spk_conv3x3_dgrad_f16x2_pack_multi: two launches for all layers)
    against the per-layer calls: the exact forward and the two-term data gradient give the same bits; a workspace packed for
    another image count is refused with NaN, not mis-read."""
    g = torch.Generator().manual_seed(17)
    shapes = [(128, 64), (256, 128), (512, 256), (256, 512), (128, 320)]
    Ns = [128, 128, 512, 512, 8]
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * (0.03 + 0.02 * i)).to(dev).contiguous(memory_format=torch.channels_last)
          .requires_grad_(True) for i, (co, ci) in enumerate(shapes)]
    ws[1].data[:, 5] *= 2.0 ** -9                                                # per-channel maxima that differ by binades
    ws[2].data[:, 7] = 0.0
    bs = [torch.randn(co, generator=g).to(dev) if i % 2 == 0 else None for i, (co, ci) in enumerate(shapes)]
    preps = ops.train_weight_prep([(w, b, n, (7, 7)) for w, b, n in zip(ws, bs, Ns)])
    assert preps is not None and len(preps) == 5
    for i, ((co, ci), w, b, n, pr) in enumerate(zip(shapes, ws, bs, Ns, preps)):
        assert pr.matches(w) and pr.dg is not None and pr.dg[2] == n
        p0 = ops.den_pack_weight_fp6(w, b)
        assert torch.equal(p0[1], pr.fp6[1]) and torch.equal(p0[2], pr.fp6[2])
        s = (torch.rand(16, 2, ci, 7, 7, generator=g) < 0.1).float().to(dev)
        s = s.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        c4 = ops.spikes_cl_to_c4(s)
        assert torch.equal(ops.den_conv3x3_fp6_raw(c4, p0, co), ops.den_conv3x3_fp6_raw(c4, pr.fp6, co))
        gy = (torch.randn(n, co, 7, 7, generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
        gi0 = ops.conv3x3_dgrad(gy, w, ci, form="f16x2")
        gi1 = ops.conv3x3_dgrad(gy, w, ci, form="f16x2", prep=pr)
        assert torch.equal(gi0, gi1) and float(gi0.abs().max()) > 0
        # the prepacked entry point itself, on a workspace packed for a DIFFERENT tile width: NaN
        if ci % 64 == 0:
            other = 8 if n >= 64 else 512
            pr2 = ops.train_weight_prep([(w, b, other, (7, 7))])[0]
            same_width = torch.equal(pr2.dg[0][:co * 9 * ci * 4], pr.dg[0][:co * 9 * ci * 4])
            gi2 = torch.empty((n, 7, 7, ci), dtype=torch.float32, device=dev)
            ops.check(ops.lib.spk_conv3x3_dgrad_f16x2_prepacked(gy.data_ptr(), pr2.dg[0].data_ptr(), pr2.dg[1], gi2.data_ptr(), n, 7, 7,
                                                                co, ci, torch.cuda.current_stream().cuda_stream), "prepacked")
            torch.cuda.synchronize()
            assert torch.equal(gi2.permute(0, 3, 1, 2), gi0) if same_width else bool(torch.isnan(gi2).all())
    w_after = ws[0]
    with torch.no_grad():
        w_after.add_(1e-3)                                                       # an optimizer step: the prep no longer matches
    assert not preps[0].matches(w_after)


def test_bn_lif_forward_with_packed_spikes(dev, ops):
    """spk_bn_lif_train_fwd_c4: the apply launch also leaves the spikes as C4 records.  Same spikes, state and statistics as
    spk_bn_lif_train_fwd, bit for bit; the C4 records are what the conversion kernel makes of the fp32 spikes."""
    import ctypes
    g = torch.Generator().manual_seed(23)
    for (T, B, C, H, W) in ((16, 4, 64, 7, 7), (16, 3, 128, 7, 7), (16, 2, 512, 7, 7), (16, 2, 256, 8, 8)):
        y = torch.randn(T, B, C, H, W, generator=g).to(dev)
        y = y.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        outs = []
        for rep in range(2):
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            s, v, c4 = ops.BNLIFTrainFunction.apply(y, gamma, beta, None, rm, rv, 0.1, 1e-5, 2.0, 1.0, 0.0, 2.0, False, True)
            outs.append((s.clone(), v.clone(), c4.clone(), rm.clone(), rv.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        s, v, c4, rm, rv = outs[0]
        assert torch.equal(c4, ops.spikes_cl_to_c4(s))
        # the three-launch form through the C-ABI
        HW = H * W
        nb = int(ops.lib.spk_bn_lif_train_ws_bytes(B, C, HW))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        s0 = torch.empty_like(s); v0 = torch.empty_like(v)
        mean0, inv0 = torch.empty(C, device=dev), torch.empty(C, device=dev)
        rm0, rv0 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        ops.check(ops.lib.spk_bn_lif_train_fwd(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm0.data_ptr(), rv0.data_ptr(),
                                               ctypes.c_float(0.1), ctypes.c_float(1e-5), None, s0.data_ptr(), v0.data_ptr(),
                                               mean0.data_ptr(), inv0.data_ptr(), ws.data_ptr(), nb, T, B, C, HW, ctypes.c_float(2.0),
                                               ctypes.c_float(1.0), ctypes.c_float(0.0), torch.cuda.current_stream().cuda_stream),
                  "spk_bn_lif_train_fwd")
        torch.cuda.synchronize()
        assert torch.equal(rm, rm0) and torch.equal(rv, rv0) and torch.equal(s, s0) and torch.equal(v, v0)


def test_training_iteration_with_and_without_weight_prep(dev, ops):
    """One diffusion training iteration (train_iter + backward) with the per-iteration weight preparation and the packed-spike
    hand-over between blocks, against the same iteration with every layer packing for itself: same loss, same gradients, bit
    for bit (the prepared operands are the same bytes)."""
    from spkdiff import synth
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion, functional
    res = []
    for prep in (True, False):
        torch.manual_seed(5)
        den = DummyModel(1, 128, n_steps=16).to(dev)
        functional.set_step_mode(net=den, step_mode='m')
        den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
        den.train()
        den.train_weight_prep = prep
        ab = AbsorbingDiffusion(den, mask_id=128)
        x0 = torch.randint(0, 128, (8, 1, 7, 7), generator=torch.Generator().manual_seed(42)).float().to(dev)
        torch.manual_seed(11)
        loss = ab.train_iter(x0)['loss']
        loss.backward()
        functional.reset_net(den)
        res.append((loss.detach().clone(), [p.grad.clone() for p in den.parameters()],
                    [b.clone() for n, b in den.named_buffers() if 'running' in n]))
    assert torch.equal(res[0][0], res[1][0]) and float(res[0][0]) > 0
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)


def test_packed_spike_hand_over_between_training_blocks(dev, ops):
    """FusedSequential.train_forward(want_c4=True) leaves the block's spikes as C4 records next to the fp32 tensor (tagged with
    the tensor's version) for the next block's exact forward: they are what the conversion kernel makes of the fp32 spikes, the
    next block gives the same result with and without them, and the fp32 tensor cannot be rewritten in place behind them
    (autograd refuses in-place writes to this operator's outputs)."""
    from snn_model.vq_diffusion import DummyModel, functional
    torch.manual_seed(3)
    den = DummyModel(1, 128, n_steps=16).to(dev)
    functional.set_step_mode(net=den, step_mode='m')
    den.load_state_dict(synth.synth_denoiser_state(synth.MNIST))
    den.train()
    h = torch.randn(16, 4, 2, 7, 7, device=dev)
    x1 = den.conv1.train_forward(h, want_c4=True)
    c4, ver = x1._spk_c4
    assert ver == x1._version and torch.equal(c4, ops.spikes_cl_to_c4(x1.detach()))
    y_fast = den.conv2.train_forward(x1, binary_input=True).detach().clone()
    functional.reset_net(den)
    x1c = den.conv1.train_forward(h, want_c4=False)
    assert getattr(x1c, '_spk_c4', None) is None
    y_ref = den.conv2.train_forward(x1c, binary_input=True).detach()
    assert torch.equal(y_fast, y_ref) and float(y_ref.abs().max()) > 0
    with pytest.raises(RuntimeError):
        x1.mul_(0.0)
    functional.reset_net(den)
