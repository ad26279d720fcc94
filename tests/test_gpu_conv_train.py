"""Native training convolutions (csrc/conv_train.hip): forward, data gradient and weight / bias gradient of the
spiking VQ-VAE's layer shapes (R/snn_model/vae_model.py:101-159) and of ragged / odd shapes, against torch's fp64 convolution
and its autograd on the CPU (the operator the reference's loss.backward() runs, R/main.py:136-142).  Everything goes through
ops.NativeConvTrainFunction / ops.ExactConvTrainFunction -> ctypes -> the C-ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from spkdiff import ops as o
    return o


def _rel_l2(got, want):
    return float((got.double() - want.double()).norm() / (want.double().norm() + 1e-30))


# (Cin, Cout, k, stride, pad, transposed, out_pad, H, N, binary input)
VQVAE_LAYERS = [
    (1, 32, 3, 2, 1, False, 0, 28, 16, False),     # encoder conv1 (image input; one reduced channel: vector kernel)
    (32, 64, 3, 2, 1, False, 0, 14, 16, True),     # encoder conv2
    (64, 16, 1, 1, 0, False, 0, 7, 16, True),      # encoder conv3 (1x1, 16 output channels: half a column tile)
    (16, 64, 3, 2, 1, True, 1, 7, 16, False),      # decoder convT1 (dense input: the quantised latent)
    (64, 32, 3, 2, 1, True, 1, 14, 16, True),      # decoder convT2
    (32, 1, 3, 1, 1, True, 0, 28, 16, True),       # decoder convT3 (one output channel: vector kernels)
]
ODD_SHAPES = [
    (8, 24, 3, 1, 1, False, 0, 5, 3, False),       # 75 rows: a ragged last 64-row group; 24 of 32 columns
    (24, 40, 3, 2, 1, False, 0, 9, 2, False),      # two column tiles, the second one ragged; odd map
    (16, 16, 3, 2, 1, True, 1, 5, 3, False),       # sub-pixel classes of different sizes (10x10 output of a 5x5 input)
    (40, 8, 2, 2, 0, True, 0, 6, 2, False),        # k = 2 / s = 2: one tap per class
    (16, 32, 3, 1, 1, True, 0, 6, 5, False),       # stride-1 transposed layer on the matrix path
    (64, 64, 3, 1, 1, False, 0, 7, 4, True),       # 144 KB of weight taps in LDS (one workgroup per CU)
    (1, 8, 3, 1, 1, False, 0, 6, 3, False),        # one reduced channel, stride 1
    (16, 16, 3, 2, 1, True, 0, 5, 3, False),       # 9x9 output of a 5x5 input: the odd sub-pixel classes are one row / column short
    (8, 8, 1, 2, 0, False, 0, 6, 2, False),        # 1x1 / stride 2: three of the four classes of its data gradient have no tap (zeros)
    (32, 32, 3, 2, 1, False, 0, 28, 2, True),      # 28x28 maps: the weight gradient's operand rows do not fit LDS four rows at a time
    (3, 32, 3, 2, 1, False, 0, 16, 4, False),      # the first layer on RGB (CIFAR-shaped model): three reduced channels, vector kernels
    (32, 3, 3, 1, 1, True, 0, 16, 4, True),        # the read-out layer on RGB: three output channels
    (2, 16, 3, 1, 1, False, 0, 7, 5, False),       # two reduced channels at stride 1 (the denoiser's first layer shape)
]


@pytest.mark.parametrize("cfg", VQVAE_LAYERS + ODD_SHAPES, ids=lambda c: "-".join(str(int(v)) for v in c))
@pytest.mark.parametrize("w_cl", [True, False], ids=["wCL", "wNCHW"])
def test_native_conv_train_forward_and_backward_vs_fp64_autograd(dev, ops, cfg, w_cl):
    cin, cout, k, st, pd, tr, op, H, N, binary = cfg
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k + st + H)
    x = (torch.rand(N, cin, H, H, generator=g) < 0.15).float() if binary else torch.randn(N, cin, H, H, generator=g)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(*wshape, generator=g) * 0.2
    b = torch.randn(cout, generator=g) * 0.1
    xo, wo, bo = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yo = F.conv_transpose2d(xo, wo, bo, st, pd, op) if tr else F.conv2d(xo, wo, bo, st, pd)
    gy = torch.randn(yo.shape, generator=g)
    (yo * gy.double()).sum().backward()

    need_gi = not (cin <= 4 and st > 1)            # (the first layer's image input takes no gradient)
    assert ops.conv_train_supported(x.shape, w.to(dev), st, pd, tr, op, need_gi, forward=True), "shape not taken by the native kernels"
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(need_gi)
    wd = w.to(dev)
    if w_cl:
        wd = wd.contiguous(memory_format=torch.channels_last)
    wd = wd.requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.NativeConvTrainFunction.apply(xd, wd, bd, st, pd, tr, op)
    assert tuple(y.shape) == tuple(yo.shape) and y.permute(0, 2, 3, 1).is_contiguous()
    gyd = gy.to(dev).contiguous(memory_format=torch.channels_last)
    (y * gyd).sum().backward()
    torch.cuda.synchronize()
    e_y = _rel_l2(y.detach().cpu(), yo.detach())
    e_gi = _rel_l2(xd.grad.cpu(), xo.grad) if need_gi else 0.0
    e_gw, e_gb = _rel_l2(wd.grad.cpu(), wo.grad), _rel_l2(bd.grad.cpu(), bo.grad)
    assert wd.grad.stride() == wd.stride()
    assert max(e_y, e_gi, e_gw, e_gb) <= 2e-6, (e_y, e_gi, e_gw, e_gb)
    # element-wise as well: nothing misplaced that an L2 norm would average away
    assert float((y.detach().cpu().double() - yo.detach()).abs().max()) <= 1e-5 * (1 + float(yo.abs().max()))
    if need_gi:
        assert float((xd.grad.cpu().double() - xo.grad).abs().max()) <= 1e-5 * (1 + float(xo.grad.abs().max()))
    assert float((wd.grad.cpu().double() - wo.grad).abs().max()) <= 1e-5 * (1 + float(wo.grad.abs().max()))


def test_native_conv_train_is_deterministic_and_layout_agnostic(dev, ops):
    """Fixed-order partial sums: two runs are bit-identical; an NCHW input / gradient gives the same numbers as a channels-last one."""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(40, 64, 14, 14, generator=g) < 0.1).float().to(dev)
    w = (torch.randn(64, 32, 3, 3, generator=g) * 0.1).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(40, 32, 28, 28, generator=g).to(dev)
    outs = []
    for xin, gin in ((x, gy), (x.contiguous(memory_format=torch.channels_last), gy.contiguous(memory_format=torch.channels_last)),
                     (x, gy)):
        xr, wr = xin.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.NativeConvTrainFunction.apply(xr, wr, None, 2, 1, True, 1)
        (y * gin).sum().backward()
        outs.append((y.detach().clone(), xr.grad.clone(), wr.grad.clone()))
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)
    for a, b_ in zip(outs[0], outs[2]):
        assert torch.equal(a, b_)


def test_exact_forward_with_native_backward(dev, ops):
    """ops.ExactConvTrainFunction (the parity runs' forward: exact direct kernel) now takes the native gradients where they fit."""
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(8, 32, 14, 14, generator=g) < 0.2).float()
    w = torch.randn(64, 32, 3, 3, generator=g) * 0.1
    b = torch.randn(64, generator=g) * 0.1
    gy = torch.randn(8, 64, 7, 7, generator=g)
    xo, wo, bo = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    (F.conv2d(xo, wo, bo, 2, 1) * gy.double()).sum().backward()
    xd, wd, bd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.ExactConvTrainFunction.apply(xd, wd, bd, 2, 1, False, 0)
    (y * gy.to(dev)).sum().backward()
    for got, want in ((xd.grad, xo.grad), (wd.grad, wo.grad), (bd.grad, bo.grad)):
        assert _rel_l2(got.cpu(), want) <= 2e-6


def test_unsupported_shapes_fall_back_to_the_framework(dev, ops):
    """Five input channels, 128 output channels: not taken -- the modules then call the framework's operator and training still runs.
    (Three input channels -- CIFAR's first layer -- ARE taken: the few-channel vector kernels.)"""
    from spikingjelly.activation_based import layer
    assert ops.conv_train_supported((4, 3, 32, 32), torch.empty(32, 3, 3, 3, device=dev), 2, 1, False, 0, False, forward=True)
    assert not ops.conv_train_supported((4, 5, 32, 32), torch.empty(32, 5, 3, 3, device=dev), 2, 1, False, 0, False, forward=True)
    assert not ops.conv_train_supported((4, 64, 8, 8), torch.empty(128, 64, 3, 3, device=dev), 1, 1, False, 0, True, forward=True)
    m = layer.Conv2d(5, 32, 3, 2, 1, step_mode='m').to(dev).train()
    x = torch.randn(2, 4, 5, 32, 32, device=dev)
    y = m(x)
    y.sum().backward()
    assert tuple(y.shape) == (2, 4, 32, 16, 16) and m.weight.grad is not None


def test_graphed_vqvae_training_step_equals_the_eager_loop(dev):
    """spkdiff.train.GraphedVQVAETrainStep: one captured iteration of the reference's VQ-VAE training loop (R/main.py:118-146:
    forward in train() mode, loss_eq + loss_rec, backward, AdamW, reset_net) replayed per batch gives the losses and the weights
    of the same iterations run launch by launch (every kernel of the iteration is deterministic), and the losses fall."""
    from snn_model.vae_model import SNN_VQVAE, functional
    from spkdiff import synth
    from spkdiff.train import GraphedVQVAETrainStep
    imgs = [(synth.stroke_images(8, 5 + i) - 0.5).to(dev) for i in range(6)]

    def make():
        m = SNN_VQVAE(1, 16, 128, 0.08).to(dev)
        functional.set_step_mode(net=m, step_mode='m')
        m.load_state_dict(synth.synth_vqvae_state(synth.MNIST))
        m.train()
        return m, torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.001, fused=True, capturable=True)

    m1, o1 = make()
    eager = []
    for x in [imgs[0]] * 3 + imgs:                     # (the three warm-up iterations of the captured form, then six batches)
        a, b, c = m1(x.unsqueeze(0).repeat(16, 1, 1, 1, 1), x)
        o1.zero_grad(); (a + b).backward(); o1.step(); functional.reset_net(m1)
        eager.append((float(a.detach()), float(b.detach()), float(c.detach())))
    m2, o2 = make()
    step = GraphedVQVAETrainStep(m2, o2, imgs[0], 16, warmup=3)
    got = [tuple(float(v.detach()) for v in step(x)) for x in imgs]
    for g, w in zip(got, eager[3:]):
        assert all(abs(gv - wv) <= 1e-5 * (1 + abs(wv)) for gv, wv in zip(g, w)), (g, w)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert float((p1.detach() - p2.detach()).abs().max()) <= 1e-6 * (1 + float(p1.detach().abs().max())), k
    with pytest.raises(RuntimeError):
        GraphedVQVAETrainStep(m2, torch.optim.AdamW(m2.parameters(), lr=1e-3), imgs[0])
    m2.eval()
    with torch.inference_mode():                       # the captured optimizer steps are visible to the inference path
        _, xr, idx = m2(imgs[0].unsqueeze(0).repeat(16, 1, 1, 1, 1), imgs[0])
    functional.reset_net(m2)
    assert xr.shape == (8, 1, 28, 28) and bool(torch.isfinite(xr).all())


def test_native_and_framework_convolutions_give_the_same_training_step(dev, ops):
    """ops.NATIVE_TRAIN_CONV = False restores the framework's convolutions (rounds 3-4): one VQ-VAE training step either way -- same
    losses to fp32 round-off, gradients within 1e-4 relative L2 (a neuron-step may flip between two fp32 summation orders)."""
    from snn_model.vae_model import SNN_VQVAE, functional
    from spkdiff import synth
    img = (synth.stroke_images(4, 9) - 0.5).to(dev)
    res = []
    for native in (True, False):
        keep = ops.NATIVE_TRAIN_CONV
        ops.NATIVE_TRAIN_CONV = native
        try:
            m = SNN_VQVAE(1, 16, 128, 0.08).to(dev)
            functional.set_step_mode(net=m, step_mode='m')
            m.load_state_dict(synth.synth_vqvae_state(synth.MNIST))
            m.train()
            a, b, c = m(img.unsqueeze(0).repeat(16, 1, 1, 1, 1), img)
            (a + b).backward()
            res.append(((float(a.detach()), float(b.detach()), float(c.detach())),
                        {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
            functional.reset_net(m)
        finally:
            ops.NATIVE_TRAIN_CONV = keep
    (l1, g1), (l2, g2) = res
    assert all(abs(x - y) <= 1e-3 * (1 + abs(y)) for x, y in zip(l1, l2)), (l1, l2)
    assert set(g1) == set(g2)
    for k in g1:
        if float(g2[k].norm()) > 1e-4:
            assert _rel_l2(g1[k].cpu(), g2[k].cpu()) <= 2e-2, (k, _rel_l2(g1[k].cpu(), g2[k].cpu()))
        else:                                   # (a convolution bias in front of a batch-statistics BN: round-off on either side)
            assert float((g1[k] - g2[k]).abs().max()) <= 5e-5, k
