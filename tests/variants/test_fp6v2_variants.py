"""The measured-and-dropped launch forms of the fp6v2 main launch (duo, deferred scan, staggered, 4 / 12 waves): bit-equality with
the default form.  They are NOT in the shipped library: build `make -C spiking-diffusion_amd/csrc variants` and run

    SPKDIFF_LIB=$PWD/spiking-diffusion_amd/spkdiff/variants/libspkdiff_variants.so python -m pytest tests/variants -m variants -q

on a GPU box (tools/variants_check.sh).  Marked ``variants`` only -- the driver's ``-m gpu`` run does not select them, and without a
GPU or with the shipped library loaded every test here is skipped."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parity_report import record as parity  # noqa: E402
from spkdiff import _lib                    # noqa: E402

pytestmark = [pytest.mark.variants,
              pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X"),
              pytest.mark.skipif(not _lib.HAS_OPTIONS, reason="shipped library loaded: set SPKDIFF_LIB to the `make variants` build")]


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from spkdiff import ops as o
    return o


@pytest.mark.parametrize("B", [1, 3, 64, 256, 300])
def test_fp6v2_deferred_scan_bit_equal_to_the_scan_between_k_loops(dev, ops, B):
    """Round 5 (an opt-in, v2_defer = 1; measured slower than the default): the LIF scan of an item runs inside the K loop of the same waves' next item
    (fp6v2_body_defer: pre-activations through a staging slab, counts summed behind the next item's first barrier, the last item of
    a workgroup scanned the old way).  Same arithmetic in another order of events: spikes AND spike counts must equal the
    round-2..4 form (v2_defer = 0) bit for bit -- workgroups with one item, with several, with none (B = 1, 3), ragged last rounds
    (B = 300 over 256 workgroups), repeated launches on one workspace, and through the active-set path."""
    from spkdiff import _lib
    g = torch.Generator().manual_seed(900 + B)
    prev = _lib.get_option("v2_defer")
    total = 0
    try:
        for Cout, Cin in ((256, 128), (512, 256), (256, 512), (128, 64)):
            w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05).to(dev)
            bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.1).to(dev)
            x = (torch.rand(16, B, Cin, 7, 7, generator=g) < 0.05).float().to(dev)
            a = (torch.rand(Cout, generator=g) * 8 + 2).to(dev)
            b = (torch.rand(Cout, generator=g) * 0.8).to(dev)
            pk, xs = ops.den_pack_weight_fp6v2(w, bias), ops.spikes_to_s32(x)
            outs = {}
            for mode in (0, 1):
                _lib.set_option("v2_defer", mode)
                for rep in range(2):
                    o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                outs[mode] = (o.clone(), c.clone())
            nbad = int((outs[1][0] != outs[0][0]).sum())
            assert nbad == 0 and torch.equal(outs[1][1], outs[0][1]), (Cout, Cin, nbad)
            assert 0.001 < float(ops.s32_to_spikes(outs[0][0]).mean()) < 0.9
            total += outs[0][0].numel() * 2
            if B >= 3:
                n = B // 2 + 1
                active = torch.arange(B, dtype=torch.int32, device=dev)
                n_act = torch.tensor([n, 0], dtype=torch.int32, device=dev)
                res = {}
                for mode in (0, 1):
                    _lib.set_option("v2_defer", mode)
                    with ops.active_set(active, n_act):
                        o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                    res[mode] = (o[:n].clone(), c[:n].clone())
                assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1]), (Cout, Cin, "active set")
                assert torch.equal(res[1][0], outs[0][0][:n]), "the first n images of the full batch"
    finally:
        _lib.set_option("v2_defer", prev)
    parity(f"fp6v2_deferred_scan_vs_scan_between_k_loops_B{B}", neuron_steps=total, spike_mismatches=0)


@pytest.mark.parametrize("B", [1, 5, 64, 256])
def test_fp6v2_duo_form_bit_equal_to_the_one_workgroup_form(dev, ops, B):
    """Round 5 (an opt-in, v2_duo = 1; measured slower than the default): two independent four-wave workgroups per CU on half-image items (fp6v2_body_duo: weight thirds
    in a ring, counted s_waitcnt, per-CU arrival parity + head start).  Same arithmetic, another schedule: every layer shape, with
    no head start (v2_duo = 1), a head start of 95 and of 400 ticks per chunk, must give the spikes AND the spike counts
    of the one-workgroup form (v2_duo = 0) bit for bit -- also through the active-set path (a device-side image count below B)."""
    from spkdiff import _lib
    g = torch.Generator().manual_seed(500 + B)
    prev = _lib.get_option("v2_duo")
    total = 0
    try:
        for Cout, Cin in ((128, 64), (256, 128), (512, 256), (256, 512)):
            w = ((torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * 0.05).to(dev)
            bias = ((torch.rand(Cout, generator=g) - 0.5) * 0.1).to(dev)
            x = (torch.rand(16, B, Cin, 7, 7, generator=g) < 0.05).float().to(dev)
            a = (torch.rand(Cout, generator=g) * 8 + 2).to(dev)
            b = (torch.rand(Cout, generator=g) * 0.8).to(dev)
            pk, xs = ops.den_pack_weight_fp6v2(w, bias), ops.spikes_to_s32(x)
            outs = {}
            for mode in (0, 1, 95, 400):
                _lib.set_option("v2_duo", mode)
                for rep in range(2):                          # (twice: the ring / arrival counters carry over between launches)
                    o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                outs[mode] = (o.clone(), c.clone())
            for mode in (1, 95, 400):
                assert torch.equal(outs[mode][0], outs[0][0]) and torch.equal(outs[mode][1], outs[0][1]), (Cout, Cin, mode)
            assert 0.001 < float(ops.s32_to_spikes(outs[0][0]).mean()) < 0.9
            total += outs[0][0].numel() * 2
            if B >= 5:
                # the sampler's active-set calls: only the first n image slots are computed (n read on the device)
                n = B // 2 + 1
                active = torch.arange(B, dtype=torch.int32, device=dev)
                n_act = torch.tensor([n, 0], dtype=torch.int32, device=dev)
                res = {}
                for mode in (0, 1):
                    _lib.set_option("v2_duo", mode)
                    with ops.active_set(active, n_act):
                        o, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
                    res[mode] = (o[:n].clone(), c[:n].clone())
                assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1]), (Cout, Cin, "active set")
                assert torch.equal(res[1][0], outs[0][0][:n]), "the first n images of the full batch"
    finally:
        _lib.set_option("v2_duo", prev)
    parity(f"fp6v2_duo_vs_one_workgroup_B{B}", neuron_steps=total, spike_mismatches=0)


def test_fp6v2_staggered_form_bit_equal(dev, ops):
    """The measured alternatives of the fp6v2 main launch kept behind spk_set_option (include/spkdiff.h) -- the staggered
    (one-chunk-lag, three LDS slots) form "v2_lag", one and three waves per SIMD "v2_waves" = 4 / 12 -- give the default form's
    spikes and counts bit for bit.  The library reads no environment variable: the host switches between calls."""
    from spkdiff import _lib
    torch.manual_seed(3)
    cases = []
    for B, Cout, Cin in ((37, 128, 64), (256, 256, 128), (19, 512, 256), (64, 256, 512)):
        w = (torch.rand(Cout, Cin, 3, 3, device=dev) - 0.5) * 0.05
        bias = (torch.rand(Cout, device=dev) - 0.5) * 0.1
        x = (torch.rand(16, B, Cin, 7, 7, device=dev) < 0.06).float()
        a = torch.rand(Cout, device=dev) * 8 + 2
        b = torch.rand(Cout, device=dev) * 0.8
        cases.append((ops.spikes_to_s32(x), ops.den_pack_weight_fp6v2(w, bias), Cout, a, b))

    def run():
        out = []
        for xs, pk, Cout, a, b in cases:
            y, c = ops.den_conv3x3_mfma_fp6v2(xs, pk, Cout, bn_a=a, bn_b=b, want_counts=True)
            out.append((y.clone(), c.clone()))
        torch.cuda.synchronize()
        return out
    assert (_lib.get_option("v2_waves"), _lib.get_option("v2_lag")) == (8, 0)
    with pytest.raises(NotImplementedError):
        _lib.set_option("no_such_option", 1)
    base = run()
    n = 0
    try:
        for name, value in (("v2_lag", 1), ("v2_waves", 4), ("v2_waves", 12)):
            _lib.set_option(name, value)
            got = run()
            _lib.set_option(name, 8 if name == "v2_waves" else 0)
            for (y0, c0), (y1, c1) in zip(base, got):
                assert torch.equal(y0, y1) and torch.equal(c0, c1), (name, value)
                n += y0.numel()
    finally:
        _lib.set_option("v2_waves", 8)
        _lib.set_option("v2_lag", 0)
    parity("fp6v2_staggered_form", bytes_compared=n, mismatches=0)
