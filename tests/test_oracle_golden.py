"""CPU tests: the oracle (oracle/snn_ref.py) against the golden fixtures captured from the REAL
reference by oracle/gen_golden.py.  Bit-exact: same ATen ops, same order.

These fixtures are what pins the oracle (the reference has no tests of its own).
If the host CPU picks a different oneDNN kernel than the build container's, a convolution may
differ in the last ulp; spikes are then compared outside the recorded fragile set.
"""
import os

import numpy as np
import pytest
import torch

from oracle import snn_ref as ref
from spkdiff import synth


def unpack(bits, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(bits)[:n].reshape(tuple(shape)).astype(np.float32))


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.fixture(scope="module")
def vae_mnist():
    return synth.synth_vqvae_state(synth.MNIST)


@pytest.fixture(scope="module")
def den_mnist():
    return synth.synth_denoiser_state(synth.MNIST)


def test_f1_lif_exact(golden_dir):
    d = load(golden_dir, "f1_lif.npz")
    x = torch.from_numpy(d["x_seq"])
    s, v = ref.lif_multi_step(x)
    assert torch.equal(s, unpack(d["spikes"], d["spikes_shape"]))
    assert torch.equal(v, torch.from_numpy(d["v"]))
    s2, v2 = ref.lif_multi_step(x.flip(0), v)           # state carried across calls (no reset)
    assert torch.equal(s2, unpack(d["spikes_carry"], d["spikes_shape"]))
    assert torch.equal(v2, torch.from_numpy(d["v_carry"]))


LIF_FORMS = [("soft_decay", dict(v_reset=None, decay_input=True, tau=2.0)),
             ("soft_nodecay", dict(v_reset=None, decay_input=False, tau=2.0)),
             ("hard_nodecay", dict(v_reset=0.0, decay_input=False, tau=2.0)),
             ("hard_decay_vseq", dict(v_reset=0.0, decay_input=True, tau=2.0)),
             ("soft_decay_tau3", dict(v_reset=None, decay_input=True, tau=3.0)),
             ("soft_nodecay_tau3", dict(v_reset=None, decay_input=False, tau=3.0)),
             ("hard_nodecay_tau5_vr", dict(v_reset=-0.25, decay_input=False, tau=5.0, v_threshold=0.8))]


@pytest.mark.parametrize("name,kw", LIF_FORMS)
def test_f14_lif_other_eval_forms(golden_dir, name, kw):
    """F14: soft reset / decay_input=False / v_seq forms of the reference's eval LIF (SJ/activation_based/neuron.py:813-900),
    state carried across two calls: the oracle against the real reference's outputs, bit for bit."""
    d = load(golden_dir, "f14_lif_forms.npz")
    x = torch.from_numpy(d["x_seq"])
    vth = kw.get("v_threshold", 1.0)
    s, v, vs = ref.lif_multi_step_ex(x, 0.0 if kw["v_reset"] is None else kw["v_reset"], vth, kw["v_reset"], kw["tau"],
                                     kw["decay_input"])
    assert torch.equal(s, unpack(d[name + "_spikes"], d["spikes_shape"]))
    assert torch.equal(vs, torch.from_numpy(d[name + "_v_seq"]))
    s2, _, vs2 = ref.lif_multi_step_ex(x.flip(0), v, vth, kw["v_reset"], kw["tau"], kw["decay_input"])
    assert torch.equal(s2, unpack(d[name + "_spikes_carry"], d["spikes_shape"]))
    assert torch.equal(vs2, torch.from_numpy(d[name + "_v_seq_carry"]))


@pytest.mark.parametrize("det", [False, True])
def test_f8_lif_training_forward_and_bptt(golden_dir, det):
    """the oracle's training-mode LIF (surrogate-gradient autograd) against the reference's
    torch-backend LIFNode in train mode: spikes, carried state and dL/dx of a fixed linear functional."""
    d = load(golden_dir, f"f8_lif_train_{'detach' if det else 'nodetach'}.npz")
    x = torch.from_numpy(d["x_seq"]).requires_grad_(True)
    w1, w2, w3 = (torch.from_numpy(d[k]) for k in ("w1", "w2", "w3"))
    sa, v = ref.lif_multi_step_train(x, detach_reset=det)
    sb, v = ref.lif_multi_step_train(x.flip(0), v, detach_reset=det)
    ((sa * w1).sum() + (sb * w2).sum() + (v * w3).sum()).backward()
    assert torch.equal(sa.detach(), unpack(d["spikes_a"], d["spikes_shape"]))
    assert torch.equal(sb.detach(), unpack(d["spikes_b"], d["spikes_shape"]))
    assert torch.equal(v.detach(), torch.from_numpy(d["v"]))
    assert torch.equal(x.grad, torch.from_numpy(d["grad_x"]))


def test_f9_diffusion_train_step(golden_dir):
    """one AbsorbingDiffusion._train_loss + backward of the reference (DummyModel in train() mode,
    batch-statistics BN, surrogate-gradient LIF) against the oracle's restatement under the same torch.manual_seed:
    the RNG order (randint, then rand_like), x_t / mask, the loss, every recorded gradient and the updated running
    statistics are bit-identical."""
    d = load(golden_dir, "f9_train_step.npz")
    sdd = synth.synth_denoiser_state(synth.MNIST)
    assert str(d["weights_crc"]) == synth.state_checksum(sdd)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v.clone())
          for k, v in sdd.items()}
    x0 = torch.from_numpy(d["x0"])
    torch.manual_seed(int(d["seed"]))
    stats = {}
    loss, (t, x_t, x0_ignore, mask, logits) = ref.train_loss(x0, sd, 128, stats_out=stats)
    loss.backward()
    assert torch.equal(t, torch.from_numpy(d["t"])) and torch.equal(x_t, torch.from_numpy(d["x_t"]))
    assert torch.equal(mask, torch.from_numpy(d["mask"])) and torch.equal(x0_ignore, torch.from_numpy(d["x0_ignore"]))
    assert torch.equal(loss.detach(), torch.from_numpy(d["loss"]))
    assert torch.equal(logits.detach(), torch.from_numpy(d["logits"]))
    for k in d.files:
        if k.startswith("grad."):
            assert torch.equal(sd[k[5:]].grad, torch.from_numpy(d[k])), k
        if k.startswith("stat."):
            assert torch.equal(stats[k[5:]], torch.from_numpy(d[k])), k
    norms = dict(zip(d["grad_names"].tolist(), d["grad_norms"].tolist()))
    for k, n in norms.items():
        assert abs(float(sd[k].grad.norm()) - n) <= 1e-6 * max(1.0, n), k


def test_f10_vqvae_train_step(golden_dir):
    """SURVEY §8f item 2 (second half): SNN_VQVAE.forward in train() mode + (loss_eq + loss_rec).backward() of the
    reference against the oracle's restatement: the three losses, the code indices and the running statistics are
    bit-identical; gradients agree to fp32 round-off (decoder / codebook / alpha ones bit-identically; where the
    read-out and PSP gradient paths meet at the encoder output autograd's accumulation order differs in the last bit)."""
    d = load(golden_dir, "f10_vqvae_train_step.npz")
    sdv = synth.synth_vqvae_state(synth.MNIST)
    assert str(d["weights_crc"]) == synth.state_checksum(sdv)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k and "coef" not in k
              else v.clone()) for k, v in sdv.items()}
    img = torch.from_numpy(d["images"])
    stats = {}
    (leq, lrec, lreal), idx = ref.snn_vqvae_train_forward(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img, sd,
                                                          torch.from_numpy(d["data_variance"]), stats_out=stats)
    (leq + lrec).backward()
    assert torch.equal(leq.detach(), torch.from_numpy(d["loss_eq"]))
    assert torch.equal(lrec.detach(), torch.from_numpy(d["loss_rec"]))
    assert torch.equal(lreal.detach(), torch.from_numpy(d["real_loss_rec"]))
    assert torch.equal(idx, torch.from_numpy(d["indices"]))
    for k in d.files:
        if k.startswith("grad."):
            want = torch.from_numpy(d[k])
            assert float((sd[k[5:]].grad - want).abs().max()) <= 1e-7 + 1e-6 * float(want.abs().max()), k
        if k.startswith("stat."):
            assert torch.equal(stats[k[5:]], torch.from_numpy(d[k])), k
    for k in ("decoder.snn_convs.3.weight", "decoder.snn_convs.6.bias", "vq_layer.embeddings.weight", "vq_layer.alpha"):
        assert torch.equal(sd[k].grad, torch.from_numpy(d["grad." + k])), k
    for k, n in zip(d["grad_names"].tolist(), d["grad_norms"].tolist()):
        got = 0.0 if sd[k].grad is None else float(sd[k].grad.norm())
        assert abs(got - n) <= 1e-6 * max(1.0, n) + 1e-7, k


def test_memout_coef():
    # coef = 0.8 ** arange(15..0) fp32, shape (16,1,1,1,1)
    c = ref.memout_coef(16)
    assert c.shape == (16, 1, 1, 1, 1)
    assert torch.equal(c.flatten(), torch.pow(0.8, torch.arange(15, -1, -1)))
    assert torch.equal(c, synth.memout_coef(16))


def test_f7_bn_exact_and_fma_form(golden_dir):
    d = load(golden_dir, "f7_bn.npz")
    sd = {"p." + k: torch.from_numpy(d[k]) for k in ("weight", "bias", "running_mean", "running_var")}
    x = torch.from_numpy(d["x"])
    y = torch.from_numpy(d["y"])
    assert torch.equal(ref.seq_bn_eval(x, sd, "p"), y)
    a, b = ref.bn_affine_terms(sd, "p")
    assert torch.equal(ref.bn_apply_fma(x, a, b), y), "the pinned fp32 fma form of eval BN no longer holds"


@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR)])
def test_f3_encode(golden_dir, tag, cfg):
    d = load(golden_dir, f"f3_encode_{tag}.npz")
    sd = synth.synth_vqvae_state(cfg)
    assert synth.state_checksum(sd) == str(d["weights_crc"]), "synthetic weight generator changed: regenerate goldens"
    images = torch.from_numpy(d["images"])
    x = images.unsqueeze(0).repeat(16, 1, 1, 1, 1)
    e, xr, idx = ref.snn_vqvae_forward(x, sd)
    assert torch.equal(idx, torch.from_numpy(d["indices"]))              # bit-exact code indices
    assert torch.equal(e, unpack(d["e_bits"], d["e_shape"]))
    assert float((xr - torch.from_numpy(d["x_recon"])).abs().max()) <= 1e-6
    assert idx.unique().numel() > 32                                     # non-degenerate codebook use
    # encode_indices == get_data_for_diff body
    assert torch.equal(ref.encode_indices(images + 0.5, sd).flatten(), idx)


@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR)])
def test_f4_decode_glue(golden_dir, tag, cfg):
    d = load(golden_dir, f"f4_decode_{tag}.npz")
    sd = synth.synth_vqvae_state(cfg)
    pred = ref.decode_tokens(torch.from_numpy(d["tokens"]), sd, 16)
    assert float((pred - torch.from_numpy(d["pred"])).abs().max()) <= 1e-6
    u8 = ref.to_uint8(pred)
    safe = d["u8_edge_dist"] > 1e-3
    assert np.array_equal(u8[safe], d["u8"][safe])


def test_f2_layers_teacher_forced(golden_dir, vae_mnist):
    d = load(golden_dir, "f2_layers_mnist.npz")
    sd = vae_mnist
    specs = {
        "enc1": ("encoder.snn_convs.0", "encoder.snn_convs.1", 2, 1, False, 0),
        "enc2": ("encoder.snn_convs.3", "encoder.snn_convs.4", 2, 1, False, 0),
        "enc3": ("encoder.snn_convs.6", "encoder.snn_convs.7", 1, 0, False, 0),
        "poisson": ("vq_layer.poisson.0", "vq_layer.poisson.1", 1, 0, False, 0),
        "dec1": ("decoder.snn_convs.0", "decoder.snn_convs.1", 2, 1, True, 1),
        "dec2": ("decoder.snn_convs.3", "decoder.snn_convs.4", 2, 1, True, 1),
    }
    for name, (cp, bp, st, pad, tr, op) in specs.items():
        if name + "_in" in d:
            inp = torch.from_numpy(d[name + "_in"]).unsqueeze(0).repeat(16, 1, 1, 1, 1)
        else:
            inp = unpack(d[name + "_in_bits"], d[name + "_in_shape"])
        s, y = ref.conv_bn_lif(inp, sd, cp, bp, st, pad, tr, op)
        want = unpack(d[name + "_out_bits"], d[name + "_out_shape"])
        frag = unpack(d[name + "_frag_bits"], d[name + "_out_shape"]).bool()
        assert torch.equal(s[~frag], want[~frag]), name
        ny = d[name + "_y_b0"].shape[0]
        assert float((y[:ny, 0] - torch.from_numpy(d[name + "_y_b0"])).abs().max()) <= 2e-6, name


@pytest.mark.parametrize("tag,cfg", [("mnist", synth.MNIST), ("cifar", synth.CIFAR)])
def test_f5_denoiser(golden_dir, tag, cfg):
    d = load(golden_dir, f"f5_denoiser_{tag}.npz")
    sdd = synth.synth_denoiser_state(cfg)
    assert synth.state_checksum(sdd) == str(d["weights_crc"])
    logits, layers = ref.denoiser_forward(torch.from_numpy(d["x_t"]).float(), torch.from_numpy(d["t"]), sdd, 16,
                                          return_layers=True)
    for i, (s, _) in enumerate(layers, 1):
        want = unpack(d[f"s{i}_bits"], d[f"s{i}_shape"])
        assert torch.equal(s, want), f"conv{i} spikes"
        assert int(s.sum()) == int(d[f"count{i}"])
    assert torch.equal(logits, torch.from_numpy(d["logits"]))


def test_f6_psample_steps_and_rng_order(golden_dir, den_mnist):
    d = load(golden_dir, "f6_psample.npz")
    B, steps = int(d["B"]), int(d["steps"])
    x = torch.ones(B, 1, 7, 7).long() * 128
    un = torch.zeros_like(x).bool()
    for i, t in enumerate(d["ts"]):
        x, un = ref.p_sample_step(x, un, torch.from_numpy(d["logits"][i]), int(t), 1.0,
                                  torch.from_numpy(d["u"][i]), torch.from_numpy(d["q"][i]))
        assert torch.equal(x, torch.from_numpy(d["x_after"][i]))
        assert torch.equal(un, torch.from_numpy(d["unmasked_after"][i]))
    # t == 1 unmasks everything: rand < 1/1 always
    assert bool(un.all()) and int(x.max()) < 128
    # full trajectory with the reference's own RNG consumption order (manual_seed, u then q per step)
    torch.manual_seed(int(d["seed"]))
    tok = ref.absorbing_sample(den_mnist, B, 128, 1.0, steps, 7, 16)
    assert torch.equal(tok, torch.from_numpy(d["final_tokens"]))


def test_conv_shape_errors():
    # error behaviour of the step-mode wrappers: SJ/activation_based/layer.py:170,322,464
    with pytest.raises(ValueError):
        ref.seq_conv2d(torch.zeros(2, 1, 4, 4), torch.zeros(1, 1, 3, 3), None)
    with pytest.raises(ValueError):
        ref.seq_bn_eval(torch.zeros(2, 1, 4, 4), {}, "x")


def test_generalised_T4_runs(vae_mnist):
    # BASELINE config 1 (T=4): unrunnable on the unmodified reference (coef fixed at 16 steps) -> parity unpinned;
    # the restatement must at least be self-consistent and shape-correct.
    sd = dict(vae_mnist)
    sd["vq_layer.memout.coef"] = ref.memout_coef(4)
    sd["memout.coef"] = ref.memout_coef(4)
    img = torch.rand(3, 1, 28, 28, generator=torch.Generator().manual_seed(0)) - 0.5
    e, xr, idx = ref.snn_vqvae_forward(img.unsqueeze(0).repeat(4, 1, 1, 1, 1), sd)
    assert e.shape == (4, 3, 16, 7, 7) and xr.shape == (3, 1, 28, 28) and idx.shape == (147,)


def test_f12_get_data_for_diff_carried_state(golden_dir, vae_mnist):
    """F12: the REAL get_data_for_diff over three batches (no reset_net in its loop: membrane state carries)."""
    d = load(golden_dir, "f12_get_data_for_diff.npz")
    assert synth.state_checksum(vae_mnist) == str(d["weights_crc"])
    loader = [(torch.from_numpy(im), None) for im in d["images"]]
    with torch.inference_mode():
        got = ref.get_data_for_diff(loader, vae_mnist)
        fresh = [ref.encode_indices(im, vae_mnist) for im, _ in loader]
    for g, w, f, wf in zip(got, d["indices"], fresh, d["indices_fresh_state"]):
        assert torch.equal(g, torch.from_numpy(w)) and torch.equal(f, torch.from_numpy(wf))
    assert not np.array_equal(d["indices"][1], d["indices_fresh_state"][1]), "the carried state matters in the fixture"


def test_f13_sample_100_steps_and_decode(golden_dir, vae_mnist, den_mnist):
    """F13: 100 reverse steps (B = 8) + decode to uint8 by the real reference; the oracle under the same seed."""
    d = load(golden_dir, "f13_sample_100_steps.npz")
    assert synth.state_checksum(den_mnist) == str(d["weights_crc_den"])
    torch.manual_seed(int(d["seed"]))
    with torch.inference_mode():
        u8, tok = ref.sample_images(vae_mnist, den_mnist, int(d["B"]), 128, float(d["temp"]), int(d["steps"]), 7, 16)
    assert torch.equal(tok, torch.from_numpy(d["tokens"]))
    assert np.array_equal(u8, d["u8"])


# ------------------------------------------------------------------------------------------------- round 4: TRAINED weights
@pytest.fixture(scope="module")
def trained_sd():
    return synth.trained_state("vqvae"), synth.trained_state("denoiser")


def test_trained_checkpoint_is_a_reference_checkpoint(trained_sd):
    """checkpoints/mnist_strokes_*.npz: plain state_dicts with the reference's key set (R/main.py:199,286) --
    oracle/gen_golden_trained.py loads them strictly into the real classes -- and what makes them different from the synthetic
    N(0, sigma) weights: heavy tails and a wide dynamic range inside a channel."""
    sdv, sdd = trained_sd
    assert set(sdv) == set(synth.synth_vqvae_state(synth.MNIST)) and set(sdd) == set(synth.synth_denoiser_state(synth.MNIST))
    for k, v in synth.synth_denoiser_state(synth.MNIST).items():
        assert sdd[k].shape == v.shape and sdd[k].dtype == v.dtype, k
    w = sdd["conv4.0.weight"].flatten(1)
    kurt = float(((w - w.mean()) ** 4).mean() / w.var() ** 2)
    assert kurt > 4.0, "trained conv4 weights are heavy-tailed (uniform init: 1.8)"
    below = float((w.abs() < w.abs().amax(1, keepdim=True) / 64).float().mean())
    assert below > 0.05, "a good share of the weights sits below 2^-6 of its channel's maximum (the fixed-point rounding regime)"


def test_f3t_f4t_trained_encode_decode(golden_dir, trained_sd):
    sdv, _ = trained_sd
    d = load(golden_dir, "f3t_encode_mnist_trained.npz")
    assert synth.state_checksum(sdv) == str(d["weights_crc"])
    images = torch.from_numpy(d["images"])
    assert torch.equal(images, synth.stroke_images(images.shape[0], seed=777) - 0.5), "the fixture's images are stroke images"
    x = images.unsqueeze(0).repeat(16, 1, 1, 1, 1)
    with torch.inference_mode():
        e, xr, idx = ref.snn_vqvae_forward(x, sdv)
    assert torch.equal(idx, torch.from_numpy(d["indices"])) and torch.equal(xr, torch.from_numpy(d["x_recon"]))
    assert torch.equal(e, unpack(d["e_bits"], d["e_shape"]))
    assert float(d["recon_mse"]) < 0.02, "the trained VQ-VAE reconstructs its data"
    d4 = load(golden_dir, "f4t_decode_mnist_trained.npz")
    with torch.inference_mode():
        pred = ref.decode_tokens(torch.from_numpy(d4["tokens"]), sdv, 16)
    assert torch.equal(pred, torch.from_numpy(d4["pred"])) and np.array_equal(ref.to_uint8(pred), d4["u8"])


def test_f5t_trained_denoiser_and_exact_convolution_form(golden_dir, trained_sd):
    _, sdd = trained_sd
    d = load(golden_dir, "f5t_denoiser_mnist_trained.npz")
    assert synth.state_checksum(sdd) == str(d["weights_crc"])
    x_t, t = torch.from_numpy(d["x_t"]).float(), torch.from_numpy(d["t"])
    with torch.inference_mode():
        logits, layers = ref.denoiser_forward(x_t, t, sdd, 16, return_layers=True)
        lx, layers_x = ref.denoiser_forward(x_t, t, sdd, 16, return_layers=True, exact_conv=True)
    assert torch.equal(logits, torch.from_numpy(d["logits"]))
    for i, (s, _) in enumerate(layers, 1):
        assert torch.equal(s, unpack(d[f"s{i}_bits"], d[f"s{i}_shape"]))
    # the exact-convolution form (the HIP kernels' contract) decides every spike of this fixture like oneDNN's fp32 does
    assert all(torch.equal(a[0], b[0]) for a, b in zip(layers, layers_x))
    assert bool(((logits - lx).abs() <= 2e-6 * (1 + lx.abs())).all())


def test_f13t_trained_sample_100_steps(golden_dir, trained_sd):
    sdv, sdd = trained_sd
    d = load(golden_dir, "f13t_sample_trained.npz")
    torch.manual_seed(int(d["seed"]))
    with torch.inference_mode():
        u8, tok = ref.sample_images(sdv, sdd, int(d["B"]), 128, float(d["temp"]), int(d["steps"]), 7, 16)
    assert torch.equal(tok, torch.from_numpy(d["tokens"])) and np.array_equal(u8, d["u8"])
