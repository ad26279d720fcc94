"""CPU tests (no GPU, no compute calls): the C-ABI library loads and exports every symbol ``include/spkdiff.h``
declares; the drop-in Python surface has the reference's names, constructor signatures, state_dict keys and
state-lifetime semantics; and the product path refuses CPU tensors instead of falling back."""
import os
import re

import pytest
import torch

from spkdiff import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "spkdiff.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(spk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from spkdiff import _lib
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/spkdiff.h but not exported by libspkdiff.so"
        assert n in _lib.EXPORTS, f"{n} has no ctypes signature in spkdiff/_lib.py"
    assert set(_lib.EXPORTS) == set(names)
    assert _lib.version() == _lib.EXPECTED_VERSION == 105
    # the shipped library keeps no process-wide state: the option entry points exist only in a `make variants` build
    # (include/spkdiff_variants.h), and its launch-shape options read as their compiled-in defaults
    if not os.environ.get("SPKDIFF_LIB"):
        assert not _lib.HAS_OPTIONS and not hasattr(_lib.lib, "spk_set_option") and not hasattr(_lib.lib, "spk_get_option")
        assert _lib.get_option("v2_waves") == 8 and _lib.get_option("v2_duo") == 0
        with pytest.raises(NotImplementedError):
            _lib.set_option("v2_duo", 1)
    assert _lib.lib.spk_error_string(-1).decode().startswith("spkdiff: invalid argument")


def test_argument_errors_map_to_python_exceptions():
    from spkdiff import _lib
    # null pointers / bad sizes are rejected on the host before any launch (no GPU needed)
    assert _lib.lib.spk_lif_fwd(None, None, None, 16, 10, 2.0, 1.0, 0.0, 0, None) == -1
    assert _lib.lib.spk_conv_out_size(28, 3, 2, 1, 0, 0) == 14
    assert _lib.lib.spk_conv_out_size(7, 3, 2, 1, 1, 1) == 14
    with pytest.raises(ValueError):
        _lib.check(-1, "x")
    with pytest.raises(NotImplementedError):
        _lib.check(-2, "x")
    with pytest.raises(_lib.SpkdiffError):
        _lib.check(1, "x")


def test_constant_input_lif_table_matches_the_fp32_recurrence_on_every_float():
    """The time-invariant-input layers replace the sixteen LIF steps of a stateless call by a table look-up
    (csrc/spk_common.h: spk_lif_const_input_bits16).  The table is checked here against the neuron's fp32 recurrence
    (SJ/activation_based/neuron.py:799-811: h = v + (x - v) / tau, spike = h >= v_th, hard reset to 0; tau 2, v_th 1) on EVERY
    float in [0.5, 4) -- 25 M inputs -- with the device function's selection logic restated in numpy; outside that range the
    train is trivial (x <= 1 or NaN: silent; x >= 2: a spike at every step), checked on samples."""
    import ctypes
    import numpy as np
    from spkdiff import _lib
    th = np.zeros(16, dtype=np.float32)
    pat = np.zeros(18, dtype=np.uint32)
    assert _lib.lib.spk_lif_const_input_table(th.ctypes.data_as(ctypes.c_void_p), pat.ctypes.data_as(ctypes.c_void_p)) == 0
    assert _lib.lib.spk_lif_const_input_table(None, None) == -1
    assert np.all(np.diff(th) < 0) and th[0] == 2.0

    def recurrence_bits(x):
        v = np.zeros_like(x)
        bits = np.zeros(x.shape, dtype=np.uint32)
        for t in range(16):
            h = v + (x - v) * np.float32(0.5)
            s = h >= np.float32(1.0)
            bits |= s.astype(np.uint32) << np.uint32(t)
            v = np.where(s, np.float32(0), h)
        return bits

    def table_bits(x):
        with np.errstate(invalid="ignore", over="ignore"):
            t = x - np.float32(1.0)
            _, e = np.frexp(t)
            k = np.clip(1 - e, 1, 16)
            p = np.where(x >= th[k - 1], k, k + 1)
            p = np.where(x >= np.float32(2.0), 1, p)
            p = np.where(x > np.float32(1.0), p, 17)
        return pat[p]

    n_bad = 0
    lo, hi = 0x3F000000, 0x40800000                       # [0.5, 4.0)
    for a in range(lo, hi, 1 << 22):
        x = np.arange(a, min(a + (1 << 22), hi), dtype=np.uint32).view(np.float32)
        n_bad += int(np.count_nonzero(recurrence_bits(x) != table_bits(x)))
    assert n_bad == 0
    with np.errstate(invalid="ignore", over="ignore"):
        rng = np.random.default_rng(0)
        x = rng.integers(0, 1 << 32, size=1 << 20, dtype=np.uint64).astype(np.uint32).view(np.float32)   # any bit pattern
        x = np.concatenate([x, np.array([0.0, -0.0, 1.0, 2.0, np.inf, -np.inf, np.nan, 3.4e38, 1e-45], dtype=np.float32)])
        assert np.array_equal(recurrence_bits(x), table_bits(x))


def test_workspace_sizes_and_shape_support_are_host_side():
    """The size / support queries of the round-2 entry points answer on the host (no GPU): position-list buffer layout,
    flag workspaces, packed-weight sizes, and which VQ-VAE layer shapes the fp6 kernel family takes."""
    from spkdiff import _lib, ops
    lib = _lib.lib
    B, R = 256, 3
    lists = 64 + R * 64 + R * 6 * B * 4
    assert lib.spk_select_needed_bytes(B, R) == (lists + 63) // 64 * 64 + R * B * 64
    assert lib.spk_select_needed_bytes(0, 3) == -1 and lib.spk_select_needed_bytes(4, 9) == -1
    cap = 1 << 20
    # (count, published count, id list, overflow bitmap, hand-over ticket; a `make variants` library adds the duo form's
    #  counters and the deferred-scan form's staging slabs behind the ticket)
    extra = (2048 + 128 + 256 * 24576) if _lib.HAS_OPTIONS else 0
    assert lib.spk_den_fp6v2_flag_words(2, 64, 7, 7) == 2 + cap + (2 * 64 * 49 + 31) // 32 + 1 + extra
    assert lib.spk_vae_fp6_flag_words(2, 32, 28, 28) == 2 + cap + (2 * 32 * 784 + 31) // 32 + 1
    assert lib.spk_vae_fp6_packed_bytes(32, 64) == 9 * 5 * 1536 and lib.spk_vae_fp6_packed_bytes(64, 16) == 2 * 9 * 3 * 1536
    assert lib.spk_vae_fp6_packed_bytes(32, 128) == -1 and lib.spk_vae_fp6_packed_bytes(48, 64) == -1
    # null pointers are argument errors before any launch
    assert lib.spk_vae_fp6_fwd(None, None, None, None, None, None, None, None, None, 0, None, 16, 1, 14, 14, 64, 32, 1, -1, None) == -1
    assert lib.spk_select_needed(None, 1, None, 0, 0, None, None, None, None, 1, 7, 7, 3, 128, None) == -1
    assert lib.spk_readout_collapsed_fwd(None, None, None, 1.0, None, None, 0, 1, 28, 28, 32, 1, 3, 1, 1, None) == -1
    kind = ops.vae_fp6_kind
    assert kind(64, 32, 3, 2, 1, 1, True, 16, 14, 14) == ops.VAE_OUT_COLLAPSED        # decoder convT2
    assert kind(16, 64, 3, 2, 1, 1, True, 16, 7, 7) == ops.VAE_OUT_S32                # decoder convT1
    assert kind(32, 64, 3, 2, 1, 0, False, 16, 14, 14) == ops.VAE_OUT_PTC             # encoder conv2
    assert kind(64, 32, 3, 2, 1, 1, True, 4, 14, 14) is None and kind(64, 32, 3, 1, 1, 0, True, 16, 14, 14) is None
    assert kind(32, 64, 3, 2, 1, 0, False, 16, 12, 12) is None
    assert ops.readout_collapsed_supported(32, 1, 3) and not ops.readout_collapsed_supported(30, 1, 3)


def test_state_dict_keys_match_reference():
    from snn_model.vae_model import SNN_VQVAE, functional
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion
    for cfg in (synth.MNIST, synth.CIFAR):
        m = SNN_VQVAE(cfg.in_dim, 16, 128, torch.tensor(1.0))
        sd = synth.synth_vqvae_state(cfg)
        assert list(m.state_dict().keys()) == [k for k in m.state_dict().keys()]
        assert set(m.state_dict().keys()) == set(sd.keys())
        for k, v in m.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k].shape), k
        m.load_state_dict(sd)
        assert m.decoder.snn_convs[0].weight.shape == (16, 64, 3, 3)      # ConvT weights are [Cin, Cout, k, k]
    d = DummyModel(1, 128)
    sdd = synth.synth_denoiser_state(synth.MNIST)
    assert set(d.state_dict().keys()) == set(sdd.keys())
    d.load_state_dict(sdd)
    functional.set_step_mode(net=d, step_mode='m')
    assert all(m.step_mode == 'm' for m in d.modules() if hasattr(m, 'step_mode'))
    ab = AbsorbingDiffusion(d, mask_id=128)
    assert (ab.num_classes, ab.shape, ab.num_timesteps, ab.n_samples, ab.mask_id) == (128, [7, 7], 49, 16, 128)
    assert tuple(m.memout.coef.shape) == (16, 1, 1, 1, 1)


def test_memory_module_semantics():
    from spikingjelly.activation_based import neuron, functional, surrogate, base
    n = neuron.LIFNode(surrogate_function=surrogate.ATan())
    assert (n.tau, n.v_threshold, n.v_reset, n.decay_input, n.step_mode, n.backend) == (2.0, 1.0, 0.0, True, 's', 'torch')
    assert n.v == 0.0 and isinstance(n.v, float)
    assert "v" not in n.state_dict()
    n.v = torch.ones(3)
    assert torch.is_tensor(n.v)
    n.double()
    assert n.v.dtype == torch.float64                              # memories follow _apply
    functional.reset_net(torch.nn.Sequential(n))
    assert n.v == 0.0 and isinstance(n.v, float)
    assert n.supported_backends == ('torch', 'hip')
    functional.set_backend(n, 'hip')
    assert n.backend == 'hip'
    with pytest.raises(NotImplementedError):
        n.backend = 'lava'
    with pytest.raises(ValueError):
        n.step_mode = 'q'
    with pytest.raises(AssertionError):
        neuron.LIFNode(tau=1.0)
    m = base.MemoryModule()
    m.register_memory('s', [1, 2])
    m.s.append(3)
    m.reset()
    assert m.s == [1, 2]


def test_no_cpu_fallback_and_training_branches_raise():
    from snn_model.vae_model import SNN_VQVAE, functional
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion
    m = SNN_VQVAE(1, 16, 128, torch.tensor(1.0))
    functional.set_step_mode(net=m, step_mode='m')
    m.eval()
    x = torch.zeros(16, 2, 1, 28, 28)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, x[0])
    m.train()
    with pytest.raises(RuntimeError, match="CPU"):
        m(x, x[0])
    with torch.no_grad(), pytest.raises(NotImplementedError):
        m(x, x[0])
    d = DummyModel(1, 128)
    functional.set_step_mode(net=d, step_mode='m')
    d.eval()
    with pytest.raises(RuntimeError):
        AbsorbingDiffusion(d, 128).sample(sample_steps=2)
    d.train()                                     # the training graph has no CPU path either
    with pytest.raises(RuntimeError, match="CPU"):
        AbsorbingDiffusion(d, 128).train_iter(torch.zeros(2, 1, 7, 7))


def test_training_noise_follows_the_reference_draw_order():
    """sample_time / q_sample (R/snn_model/vq_diffusion.py:56-74) are host logic over torch's generator: under the
    fixture's torch.manual_seed they reproduce the reference's t, x_t, x_0_ignore and mask exactly (fixture F9)."""
    import numpy as np
    from snn_model.vq_diffusion import DummyModel, AbsorbingDiffusion
    d = np.load(os.path.join(ROOT, "tests", "golden", "f9_train_step.npz"))
    ab = AbsorbingDiffusion(DummyModel(1, 128), mask_id=128)
    x0 = torch.from_numpy(d["x0"])
    torch.manual_seed(int(d["seed"]))
    t, pt = ab.sample_time(x0.shape[0], x0.device)
    x_t, x0_ignore, mask = ab.q_sample(x_0=x0, t=t)
    assert torch.equal(t, torch.from_numpy(d["t"])) and abs(float(pt[0]) - 1 / 49) < 1e-8
    assert torch.equal(x_t, torch.from_numpy(d["x_t"])) and torch.equal(mask, torch.from_numpy(d["mask"]))
    assert torch.equal(x0_ignore, torch.from_numpy(d["x0_ignore"]))


def test_oracle_is_not_imported_by_the_product():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import snn_model.vq_diffusion, spkdiff.ops, spkdiff.fused; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle leaked'"
            % os.path.join(ROOT, "spiking-diffusion_amd"))
    subprocess.run([sys.executable, "-c", code], check=True, cwd="/tmp")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spiking-diffusion_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_synthetic_checkpoint_cache_is_safe_under_concurrent_ranks(tmp_path):
    """bench.py's ranks all ask for the same synthetic checkpoints at start-up: the file cache (synth.cached_state) must hand every
    process the generator's own tensors, whoever wins the race to generate, and leave no lock behind."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path[:0] = [os.path.join(%r, 'spiking-diffusion_amd'), %r]\n"
            "from spkdiff import synth\n"
            "print(synth.state_checksum(synth.cached_state('denoiser', synth.MNIST)), "
            "synth.state_checksum(synth.cached_state('vqvae', synth.CIFAR)))\n" % (root, root))
    env = dict(os.environ, SPKDIFF_SYNTH_CACHE=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True) for _ in range(4)]
    outs = [p.communicate(timeout=600)[0].split() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    from spkdiff import synth
    want = [synth.state_checksum(synth.synth_denoiser_state(synth.MNIST)), synth.state_checksum(synth.synth_vqvae_state(synth.CIFAR))]
    assert all(o == want for o in outs), (outs, want)
    left = sorted(os.listdir(tmp_path))
    assert len([f for f in left if f.endswith(".pt")]) == 2 and not any(f.endswith((".lock", ".tmp")) for f in left), left


def test_header_is_plain_c(tmp_path):
    """include/spkdiff.h is the boundary a C / cgo / JNI binding includes: it must compile as C without any HIP header."""
    import shutil, subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "inc.c"
    # (spkdiff_variants.h: the two option entry points of the separate `make variants` library -- plain C as well)
    src.write_text('#include "spkdiff.h"\n#include "spkdiff_variants.h"\n'
                   'int main(void) { return spk_conv_out_size(7, 3, 1, 1, 0, 0) == 7 ? 0 : 1; }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr



def test_shipped_library_has_no_settable_state_and_variant_sources_stay_out_of_it():
    """the measured-and-dropped launch forms and the process-wide options are NOT in libspkdiff.so.  The forms' source is
    an include file only a -DSPK_V2_VARIANTS=1 build pulls in; the default Makefile target compiles csrc/*.hip and nothing under
    csrc/variants/; the variants header declares exactly the two option entry points."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "spiking-diffusion_amd", "csrc")
    assert os.path.exists(os.path.join(csrc, "variants", "fp6v2_forms.inc"))
    assert not [f for f in os.listdir(os.path.join(csrc, "variants")) if f.endswith(".hip")]
    main = open(os.path.join(csrc, "den_mfma_fp6v2.hip")).read()
    assert '#if SPK_V2_VARIANTS\n#include "variants/fp6v2_forms.inc"' in main
    assert "duo_kernel" not in main.split('#include "variants/fp6v2_forms.inc"')[0]
    assert len(main.splitlines()) < 2000
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "spkdiff_variants.h")).read(), flags=re.S)
    assert sorted(set(re.findall(r"\b(spk_[a-z0-9_]+)\s*\(", txt))) == ["spk_get_option", "spk_set_option"]
    assert "spk_set_option" not in re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "spkdiff.h")).read(), flags=re.S)


def test_synth_cache_takes_over_a_stale_lock(tmp_path, monkeypatch):
    """a generator killed in mid-run left its .lock behind and every later process (all eight bench ranks) waited
    the full timeout.  The lock now holds its owner's pid: a dead owner is taken over at once."""
    import hashlib, time, zlib
    from spkdiff import synth
    monkeypatch.setenv("SPKDIFF_SYNTH_CACHE", str(tmp_path / "cache"))
    os.makedirs(tmp_path / "cache", mode=0o700)
    with open(synth.__file__, "rb") as f:
        ver = zlib.crc32(f.read())
    import torch
    key = hashlib.sha256(repr(("vqvae", synth.MNIST, [], ver, torch.__version__)).encode()).hexdigest()[:20]
    with open(tmp_path / "cache" / (key + ".lock"), "w") as f:
        f.write("999999999")                      # no such process
    t0 = time.time()
    sd = synth.cached_state("vqvae")
    assert time.time() - t0 < 60 and "encoder.snn_convs.0.weight" in sd
    assert not os.path.exists(tmp_path / "cache" / (key + ".lock")) and os.path.exists(tmp_path / "cache" / (key + ".pt"))
    # a cache directory that others may write to is not trusted (nothing is read from or written to it)
    os.chmod(tmp_path / "cache", 0o777)
    os.unlink(tmp_path / "cache" / (key + ".pt"))
    sd2 = synth.cached_state("vqvae")
    assert not os.path.exists(tmp_path / "cache" / (key + ".pt"))
    assert synth.state_checksum(sd) == synth.state_checksum(sd2)


def test_flag_scope_is_thread_local():
    """a flag_scope opened by the capturing thread must not be seen by other threads' certified-kernel calls."""
    import threading
    from spkdiff import ops
    seen = []
    store = {}
    with ops.flag_scope(store):
        assert getattr(ops._FLAG_TLS, "store", None) is store
        th = threading.Thread(target=lambda: seen.append(getattr(ops._FLAG_TLS, "store", None)))
        th.start(); th.join()
    assert seen == [None] and getattr(ops._FLAG_TLS, "store", None) is None


def test_stroke_images_are_deterministic_and_image_like():
    import torch
    from spkdiff import synth
    a, b = synth.stroke_images(48, seed=7), synth.stroke_images(48, seed=7)
    assert torch.equal(a, b) and a.shape == (48, 1, 28, 28) and a.dtype == torch.float32
    assert 0.0 <= float(a.min()) and float(a.max()) <= 1.0 and 0.03 < float(a.mean()) < 0.3
    assert not torch.equal(a, synth.stroke_images(48, seed=8))
    assert synth.stroke_images(4, seed=7, img=32, channels=3).shape == (4, 3, 32, 32)


def test_round4_training_entry_points_check_their_arguments_on_the_host():
    """The training entry points added in round 4 reject bad arguments before any launch (no GPU): the multi-layer weight
    packing (null tables, more than eight layers, a layer whose channel counts the kernels do not take), the prepacked data
    gradient, the BatchNorm+LIF forward that also writes packed spikes, the small-input weight gradient, q_sample."""
    import ctypes
    from spkdiff import _lib
    lib = _lib.lib
    nine = (ctypes.c_void_p * 9)(*([1] * 9))
    i9 = (ctypes.c_int * 9)(*([64] * 9))
    l9 = (ctypes.c_longlong * 9)(*([1 << 30] * 9))
    assert lib.spk_den_pack_weight_fp6_cl_multi(None, None, None, None, None, None, None, 1, None) == -1
    assert lib.spk_den_pack_weight_fp6_cl_multi(nine, None, nine, nine, nine, i9, i9, 0, None) == -1
    assert lib.spk_den_pack_weight_fp6_cl_multi(nine, None, nine, nine, nine, i9, i9, 9, None) == -2       # more than eight layers
    bad = (ctypes.c_int * 1)(48)                                                                            # Cin % 64 != 0
    assert lib.spk_den_pack_weight_fp6_cl_multi(nine, None, nine, nine, nine, i9, bad, 1, None) == -2
    assert lib.spk_conv3x3_dgrad_f16x2_pack_multi(None, None, None, None, None, None, 1, None) == -1
    assert lib.spk_conv3x3_dgrad_f16x2_pack_multi(nine, nine, l9, i9, i9, i9, 9, None) == -2
    assert lib.spk_conv3x3_dgrad_f16x2_pack_multi(nine, nine, l9, i9, i9, bad, 1, None) == -2               # Cin % 32 != 0
    small = (ctypes.c_longlong * 1)(16)
    assert lib.spk_conv3x3_dgrad_f16x2_pack_multi(nine, nine, small, i9, i9, i9, 1, None) == -1             # workspace too small
    assert lib.spk_conv3x3_dgrad_f16x2_prepacked(None, None, 0, None, 8, 7, 7, 64, 64, None) == -1
    assert lib.spk_conv3x3_dgrad_f16x2_prepacked(1, 1, 1 << 30, 1, 8, 7, 9, 64, 64, None) == -2             # not a 7x7 / 8x8 map
    assert lib.spk_conv3x3_wgrad_bf16(1, 1, 1, 1 << 40, 1, None, 8, 9, 9, 128, 64, None) == -2              # 9x9 maps
    assert lib.spk_conv3x3_wgrad_bf16(1, 1, 1, 16, 1, None, 8, 8, 8, 128, 64, None) == -1                   # workspace too small
    assert lib.spk_conv3x3_wgrad_ws_bytes(512, 128, 64) > 0 and lib.spk_conv3x3_wgrad_ws_bytes(512, 100, 64) == -1
    f = ctypes.c_float
    assert lib.spk_bn_lif_train_fwd_c4(None, None, None, None, None, f(0.1), f(1e-5), None, None, None, None, None, None, None, 0,
                                       16, 2, 64, 49, f(2.0), f(1.0), f(0.0), None) == -1
    # packed spikes need whole 64-channel records: refused before anything is launched
    ws = lib.spk_bn_lif_train_ws_bytes(2, 48, 49)
    assert ws > 0
    assert lib.spk_bn_lif_train_fwd_c4(16, None, None, None, None, f(0.1), f(1e-5), None, 16, None, 16, 16, 16, 16, ws,
                                       16, 2, 48, 49, f(2.0), f(1.0), f(0.0), None) == -2
    assert lib.spk_q_sample(None, None, None, None, None, None, 4, 49, 100, 128, None) == -1
    assert lib.spk_conv3x3_wgrad_small(None, None, None, 0, None, None, 8, 7, 7, 64, 2, 1, None) == -1


def test_training_convolution_entry_points_reject_and_accept_on_the_host():
    """csrc/conv_train.hip: the support queries and the argument checks run on the host (no launch): the six layers of the MNIST
    model and the first / read-out layers of the RGB model are taken in every direction they are used, shapes outside the limits
    are refused with SPK_ERR_UNSUPPORTED (the host then calls the framework's operator), null pointers / short workspaces with
    SPK_ERR_ARG."""
    from spkdiff import _lib
    lib = _lib.lib
    sup, wsb = lib.spk_conv_train_gather_supported, lib.spk_conv_train_wgrad_ws_bytes
    # (Cin, Cout, k, stride, transposed, N, Hi, Ho): forward form, backward form, weight-gradient operands
    layers = [(1, 32, 3, 2, False, 512, 28, 14), (32, 64, 3, 2, False, 512, 14, 7), (64, 16, 1, 1, False, 512, 7, 7),
              (16, 64, 3, 2, True, 512, 7, 14), (64, 32, 3, 2, True, 512, 14, 28), (32, 1, 3, 1, True, 512, 28, 28),
              (3, 32, 3, 2, False, 512, 32, 16), (32, 3, 3, 1, True, 512, 32, 32)]
    for cin, cout, k, s, tr, n, hi, ho in layers:
        assert sup(cin, cout, k, s, 1 if tr else 0) == 1, (cin, cout)
        if cin > 4:                                          # (the image input of a first layer takes no gradient)
            assert sup(cout, cin, k, s, 0 if tr else 1) == 1, (cin, cout)
        if tr:
            assert wsb(n, hi, hi, cout, cin, k) > 0, (cin, cout)
        else:
            assert wsb(n, ho, ho, cin, cout, k) > 0, (cin, cout)
    assert sup(5, 32, 3, 2, 0) == 0 and sup(64, 128, 3, 1, 0) == 0 and sup(128, 64, 3, 1, 0) == 0      # 5 / 128 channels
    assert sup(32, 32, 5, 1, 0) == 0                         # 25 taps
    assert sup(32, 32, 3, 3, 1) == 0 and sup(32, 32, 3, 3, 0) == 1                                     # transposed form: stride <= 2
    assert sup(32, 1, 3, 2, 1) == 0                          # one output channel, transposed form at stride 2
    assert wsb(512, 7, 7, 128, 64, 3) == -1 and wsb(0, 7, 7, 32, 32, 3) == -1 and wsb(512, 7, 7, 64, 1, 3) > 0
    assert wsb(8, 6, 6, 1, 64, 4) > 0 and wsb(8, 6, 6, 2, 64, 4) == -1 and wsb(8, 6, 6, 2, 20, 3) == -1   # few-channel form: k <= 3 beyond one channel, Cv / 4 a power of two
    import ctypes
    ll = ctypes.c_longlong
    assert lib.spk_conv_train_gather(None, None, None, None, 8, 7, 7, 32, 7, 7, 32, 3, 1, 1, 0, ll(32), ll(1), ll(288), None) == -1
    assert lib.spk_conv_train_gather(16, 16, None, 16, 8, 7, 7, 5, 7, 7, 32, 3, 1, 1, 0, ll(5), ll(1), ll(45), None) == -2
    assert lib.spk_conv_train_wgrad(None, None, None, ll(0), None, None, 8, 7, 7, 32, 7, 7, 32, 3, 1, 1, ll(32), ll(1), ll(288), 1,
                                    None) == -1
    assert lib.spk_conv_train_wgrad(16, 16, 16, ll(16), 16, None, 8, 7, 7, 32, 7, 7, 32, 3, 1, 1, ll(32), ll(1), ll(288), 0,
                                    None) == -1      # workspace too small
    assert lib.spk_conv_train_wgrad(16, 16, 16, ll(1 << 40), 16, None, 8, 7, 7, 128, 7, 7, 32, 3, 1, 1, ll(32), ll(1), ll(288), 0,
                                    None) == -2
