"""Deterministic synthetic checkpoints for the Spiking-Diffusion inference path.

The reference ships no checkpoints, and there is no network, so
tests, ``bench.py`` and ``smoke()`` use weights produced here: every tensor is
drawn from a CPU ``torch.Generator`` seeded by (seed, crc32(key)), and the
BatchNorm running statistics are then calibrated layer by layer on a small
synthetic batch so that the LIF layers fire at a few percent (un-calibrated
random weights are degenerate: every latent position maps to one code).

The result is a plain ``state_dict`` with exactly the key names and shapes of
the reference modules:
  * ``SNN_VQVAE``  -- R/snn_model/vae_model.py:22-196 (keys listed SURVEY.md §8b)
  * ``DummyModel`` -- R/snn_model/vq_diffusion.py:150-187

This is set-up code (it runs once, on the host, before anything is timed); it
is NOT part of the HIP hot path and NOT the oracle.  It only uses torch CPU ops.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class PathConfig:
    """Shape parameters the reference hard-codes (SURVEY.md §5 'Config')."""
    in_dim: int = 1          # image channels (1 MNIST-like, 3 CIFAR-shaped)
    img: int = 28            # image side
    latent_dim: int = 16     # embedding_dim, R/main.py:69
    num_embeddings: int = 128  # --codebook_size, R/main.py:58
    T: int = 16              # time steps

    @property
    def latent(self) -> int:
        return self.img // 4

    @property
    def tokens(self) -> int:
        return self.latent * self.latent


MNIST = PathConfig()
CIFAR = PathConfig(in_dim=3, img=32)


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFFFFFFFFFF)
    return g


def _uniform(shape, bound, seed, key):
    return (torch.rand(shape, generator=_gen(seed, key), dtype=torch.float32) * 2.0 - 1.0) * bound


def _conv_params(sd, prefix, cout, cin, k, seed, transposed=False):
    fan_in = (cout if transposed else cin) * k * k
    bound = 1.0 / (fan_in ** 0.5)
    shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    sd[prefix + ".weight"] = _uniform(shape, bound, seed, prefix + ".weight")
    sd[prefix + ".bias"] = _uniform((cout,), bound, seed, prefix + ".bias")


def _bn_params(sd, prefix, c, seed):
    # gamma in [0.8, 1.2], beta in [-0.1, 0.1]: not the identity, so the affine is exercised
    sd[prefix + ".weight"] = 1.0 + _uniform((c,), 0.2, seed, prefix + ".weight")
    sd[prefix + ".bias"] = _uniform((c,), 0.1, seed, prefix + ".bias")
    sd[prefix + ".running_mean"] = torch.zeros(c)
    sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)


def memout_coef(T: int) -> torch.Tensor:
    """``MembraneOutputLayer.coef`` buffer, R/snn_model/snn_layers.py:31-34 with T a parameter."""
    return torch.pow(0.8, torch.arange(T - 1, -1, -1))[:, None, None, None, None]


def _lif(x_seq):
    v = torch.zeros_like(x_seq[0])
    out = torch.empty_like(x_seq)
    for t in range(x_seq.shape[0]):
        v = v + (x_seq[t] - v) * 0.5
        s = (v >= 1.0).to(x_seq.dtype)
        v = (1.0 - s) * v
        out[t] = s
    return out


def _calibrate(sd, bn_prefix, y_seq):
    """Set running stats of ``bn_prefix`` to the batch statistics of y_seq [T,B,C,H,W]; return BN(y)."""
    c = y_seq.shape[2]
    flat = y_seq.transpose(0, 2).reshape(c, -1)
    mean = flat.mean(1)
    var = flat.var(1, unbiased=True).clamp_min(1e-6)
    sd[bn_prefix + ".running_mean"] = mean
    sd[bn_prefix + ".running_var"] = var
    g, b = sd[bn_prefix + ".weight"], sd[bn_prefix + ".bias"]
    a = g / torch.sqrt(var + 1e-5)
    return y_seq * a.view(1, 1, c, 1, 1) + (b - mean * a).view(1, 1, c, 1, 1)


def _seq(fn, x_seq):
    T, B = x_seq.shape[:2]
    y = fn(x_seq.flatten(0, 1))
    return y.view(T, B, *y.shape[1:])


def synth_vqvae_state(cfg: PathConfig = MNIST, seed: int = 1234, calib_batch: int = 16) -> dict:
    """Synthetic, BN-calibrated ``SNN_VQVAE`` state_dict (keys: SURVEY.md §8b)."""
    sd: dict = {}
    D, K = cfg.latent_dim, cfg.num_embeddings
    _conv_params(sd, "encoder.snn_convs.0", 32, cfg.in_dim, 3, seed)
    _bn_params(sd, "encoder.snn_convs.1", 32, seed)
    _conv_params(sd, "encoder.snn_convs.3", 64, 32, 3, seed)
    _bn_params(sd, "encoder.snn_convs.4", 64, seed)
    _conv_params(sd, "encoder.snn_convs.6", D, 64, 1, seed)
    _bn_params(sd, "encoder.snn_convs.7", D, seed)
    sd["vq_layer.alpha"] = torch.tensor(0.5)
    sd["vq_layer.memout.coef"] = memout_coef(cfg.T)
    sd["vq_layer.embeddings.weight"] = torch.zeros(K, D)
    _conv_params(sd, "vq_layer.poisson.0", D, D, 1, seed)
    _bn_params(sd, "vq_layer.poisson.1", D, seed)
    _conv_params(sd, "decoder.snn_convs.0", 64, D, 3, seed, transposed=True)
    _bn_params(sd, "decoder.snn_convs.1", 64, seed)
    _conv_params(sd, "decoder.snn_convs.3", 32, 64, 3, seed, transposed=True)
    _bn_params(sd, "decoder.snn_convs.4", 32, seed)
    _conv_params(sd, "decoder.snn_convs.6", cfg.in_dim, 32, 3, seed, transposed=True)
    sd["memout.coef"] = memout_coef(cfg.T)

    # ---- calibration pass (host, once) ----
    T = cfg.T
    img = torch.rand(calib_batch, cfg.in_dim, cfg.img, cfg.img, generator=_gen(seed, "calib.images")) - 0.5
    x = img.unsqueeze(0).repeat(T, 1, 1, 1, 1)
    p = "encoder.snn_convs."
    x = _lif(_calibrate(sd, p + "1", _seq(lambda y: F.conv2d(y, sd[p + "0.weight"], sd[p + "0.bias"], 2, 1), x)))
    x = _lif(_calibrate(sd, p + "4", _seq(lambda y: F.conv2d(y, sd[p + "3.weight"], sd[p + "3.bias"], 2, 1), x)))
    z = _lif(_calibrate(sd, p + "7", _seq(lambda y: F.conv2d(y, sd[p + "6.weight"], sd[p + "6.bias"]), x)))
    coef = sd["vq_layer.memout.coef"]
    xm = 0.5 * (z * coef).sum(0) + 0.5 * z.sum(0) / T                      # [B,D,h,w]
    flat = xm.permute(0, 2, 3, 1).reshape(-1, D)
    # codebook: K rows drawn around real latent vectors so that the argmin is non-degenerate
    pick = torch.randint(0, flat.shape[0], (K,), generator=_gen(seed, "codebook.pick"))
    noise = torch.randn(K, D, generator=_gen(seed, "codebook.noise")) * flat.std(0, keepdim=True) * 0.5
    sd["vq_layer.embeddings.weight"] = (flat[pick] + noise).contiguous()
    d = (flat ** 2).sum(1, keepdim=True) + (sd["vq_layer.embeddings.weight"] ** 2).sum(1) \
        - 2.0 * flat @ sd["vq_layer.embeddings.weight"].t()
    q = sd["vq_layer.embeddings.weight"][d.argmin(1)].view(calib_batch, cfg.latent, cfg.latent, D).permute(0, 3, 1, 2)
    x = q.unsqueeze(0).repeat(T, 1, 1, 1, 1).contiguous()
    p = "vq_layer.poisson."
    x = _lif(_calibrate(sd, p + "1", _seq(lambda y: F.conv2d(y, sd[p + "0.weight"], sd[p + "0.bias"]), x)))
    p = "decoder.snn_convs."
    x = _lif(_calibrate(sd, p + "1", _seq(
        lambda y: F.conv_transpose2d(y, sd[p + "0.weight"], sd[p + "0.bias"], 2, 1, 1), x)))
    x = _lif(_calibrate(sd, p + "4", _seq(
        lambda y: F.conv_transpose2d(y, sd[p + "3.weight"], sd[p + "3.bias"], 2, 1, 1), x)))
    return {k: v.contiguous() for k, v in sd.items()}


def synth_denoiser_state(cfg: PathConfig = MNIST, seed: int = 4321, calib_batch: int = 8,
                         sample_steps: int = 100) -> dict:
    """Synthetic, BN-calibrated ``DummyModel`` state_dict (R/snn_model/vq_diffusion.py:158-187)."""
    sd: dict = {}
    K = cfg.num_embeddings
    chans = [2, 64, 128, 256, 512, 256]
    for i in range(5):
        _conv_params(sd, f"conv{i + 1}.0", chans[i + 1], chans[i], 3, seed)
        _bn_params(sd, f"conv{i + 1}.1", chans[i + 1], seed)
    _conv_params(sd, "conv6.0", K, 256 + 64, 3, seed)

    T, L = cfg.T, cfg.latent
    tok = torch.randint(0, K + 1, (calib_batch, 1, L, L), generator=_gen(seed, "calib.tokens")).float()
    # half of the calibration batch is heavily masked, as during sampling
    mask = torch.rand(calib_batch, 1, L, L, generator=_gen(seed, "calib.mask")) < \
        torch.linspace(0.1, 1.0, calib_batch).view(-1, 1, 1, 1)
    tok[mask] = float(K)
    t = torch.randint(1, sample_steps + 1, (calib_batch,), generator=_gen(seed, "calib.t")).float()
    x = torch.cat((tok, torch.ones_like(tok) * t.view(-1, 1, 1, 1)), 1).unsqueeze(0).repeat(T, 1, 1, 1, 1)
    for i in range(5):
        p = f"conv{i + 1}."
        x = _lif(_calibrate(sd, p + "1", _seq(
            lambda y: F.conv2d(y, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1), x)))
    return {k: v.contiguous() for k, v in sd.items()}


def stroke_images(n: int, seed: int = 2024, img: int = 28, channels: int = 1) -> torch.Tensor:
    """``n`` procedurally generated digit-like images [n, channels, img, img] in [0, 1] (the value range of the reference's
    ``ToTensor`` loaders, R/load_dataset_snn.py): two to four pen strokes -- straight segments and quadratic arcs through random
    control points -- drawn with a soft-edged pen of random width on a black ground.  There is no network and no data set on
    the build machines: these stand in for MNIST when the models are TRAINED (tools/train_on_strokes.py; the trained weights
    under checkpoints/ and the ``*_trained`` fixtures come from them).  Deterministic in (n, seed, img): CPU generator, fp32."""
    g = _gen(seed, f"strokes.{img}")
    S, P = 4, 12                                              # strokes per image (some switched off), points per stroke
    ctr = torch.rand(n, S, 3, 2, generator=g) * (img * 0.64) + img * 0.18          # three control points per stroke
    on = torch.rand(n, S, generator=g) < torch.tensor([1.0, 1.0, 0.6, 0.3])
    width = 0.7 + 1.1 * torch.rand(n, 1, 1, generator=g)
    gain = 0.75 + 0.25 * torch.rand(n, 1, 1, generator=g)
    straight = (torch.rand(n, S, 1, 1, generator=g) < 0.4).float()
    tt = torch.linspace(0, 1, P).view(1, 1, P, 1)
    a, b, c = ctr[:, :, 0:1], ctr[:, :, 1:2], ctr[:, :, 2:3]
    bez = (1 - tt) ** 2 * a + 2 * (1 - tt) * tt * b + tt ** 2 * c
    lin = (1 - tt) * a + tt * c
    pts = straight * lin + (1 - straight) * bez               # [n, S, P, 2] polyline vertices
    p0, p1 = pts[:, :, :-1], pts[:, :, 1:]                    # [n, S, P-1, 2] segments
    yy, xx = torch.meshgrid(torch.arange(img, dtype=torch.float32), torch.arange(img, dtype=torch.float32), indexing="ij")
    px = torch.stack((xx, yy), -1).view(1, 1, 1, img * img, 2) + 0.5
    d = (p1 - p0).unsqueeze(3)
    rel = px - p0.unsqueeze(3)
    u = ((rel * d).sum(-1) / (d * d).sum(-1).clamp_min(1e-6)).clamp(0, 1)
    dist = (rel - u.unsqueeze(-1) * d).norm(dim=-1)           # [n, S, P-1, img*img]
    dist = dist.masked_fill(~on.view(n, S, 1, 1), 1e9).amin(dim=(1, 2))
    out = ((width.view(n, 1) + 0.8 - dist) / 1.2).clamp(0, 1) * gain.view(n, 1)
    out = out.view(n, 1, img, img)
    if channels > 1:
        tint = 0.5 + 0.5 * torch.rand(n, channels, 1, 1, generator=g)
        out = out * tint
    return out.contiguous()


CHECKPOINT_DIR = None     # default: <package root>/checkpoints


def trained_state(kind: str, name: str = "mnist_strokes") -> dict:
    """The TRAINED checkpoint committed under ``checkpoints/`` (kind 'vqvae' / 'denoiser'): weights obtained by running the
    reference's two training loops (R/main.py:118-146,202-252) with this build's training path on ``stroke_images`` from the
    synthetic initialisation (tools/train_on_strokes.py has the recipe and the log).  Same keys as the reference's
    ``model.pth`` / ``diff_model.pth`` (R/main.py:199,286); fp32, bit for bit what the ``*_trained`` fixtures were generated
    from with the real reference (oracle/gen_golden.py)."""
    import os
    import numpy as np
    root = CHECKPOINT_DIR or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "checkpoints")
    path = os.path.join(root, f"{name}_{kind}.npz")
    with np.load(path) as z:
        return {k: torch.from_numpy(z[k].copy()) for k in z.files}


def cached_state(kind: str, cfg: PathConfig = MNIST, **kw) -> dict:
    """``synth_vqvae_state`` / ``synth_denoiser_state`` (kind 'vqvae' / 'denoiser') through a file cache: the calibration pass
    runs ONCE per node -- the first process to claim the lock generates and publishes (atomic rename), the others wait for
    the file -- so that the eight ranks of a bench run do not each calibrate on an eighth of the host cores.  The cache key
    holds every argument and the generator version (crc of this file); a stale or unreadable file is regenerated."""
    import hashlib
    import os
    import tempfile
    import time
    fn = {"vqvae": synth_vqvae_state, "denoiser": synth_denoiser_state}[kind]
    with open(__file__, "rb") as f:
        ver = zlib.crc32(f.read())
    key = hashlib.sha256(repr((kind, cfg, sorted(kw.items()), ver, torch.__version__)).encode()).hexdigest()[:20]
    root = os.environ.get("SPKDIFF_SYNTH_CACHE") or os.path.join(tempfile.gettempdir(), f"spkdiff_synth_{os.getuid()}")
    os.makedirs(root, mode=0o700, exist_ok=True)
    st = os.stat(root)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        # a directory somebody else owns or may write to (the default path is predictable): do not trust files found there
        return fn(cfg, **kw)
    path, lock = os.path.join(root, key + ".pt"), os.path.join(root, key + ".lock")

    def load():
        try:
            return torch.load(path, map_location="cpu", weights_only=True)
        except Exception:
            return None
    sd = load() if os.path.exists(path) else None
    if sd is not None:
        return sd
    def lock_is_stale():
        # the lock holds its owner's pid: stale when that process is gone, or when the file is older than any generation takes
        try:
            with open(lock) as f:
                pid = int(f.read().strip() or 0)
            if pid > 0:
                try:
                    os.kill(pid, 0)
                except ProcessLookupError:
                    return True
                except PermissionError:
                    pass
            # (a live owner touches its lock every 10 s -- see the heartbeat below -- so the age bound is independent of how
            #  long a generation takes: 8 ranks calibrating on cores / 8 threads each can exceed any fixed bound)
            return time.time() - os.stat(lock).st_mtime > 120.0
        except (OSError, ValueError):
            return False

    def claim():
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)
            os.write(fd, str(os.getpid()).encode())
            os.close(fd)
            return True
        except FileExistsError:
            return False

    def owns():
        # after unlink + claim two waiters may have raced (one unlinking the other's fresh lock): the pid in the file decides
        try:
            with open(lock) as f:
                return int(f.read().strip() or 0) == os.getpid()
        except (OSError, ValueError):
            return False
    mine = claim()
    if not mine:
        t0 = time.time()
        while time.time() - t0 < 600.0:
            if os.path.exists(path):
                sd = load()
                if sd is not None:
                    return sd
            if lock_is_stale():                          # (a killed generator left its lock behind: take over at once)
                try:
                    os.unlink(lock)
                except OSError:
                    pass
                mine = claim()
                if mine:
                    time.sleep(0.05)
                    mine = owns()
                if mine:
                    break
            time.sleep(0.2)
    stop = None
    if mine:
        import threading
        stop = threading.Event()

        def heartbeat():
            while not stop.wait(10.0):
                try:
                    os.utime(lock, None)
                except OSError:
                    return
        threading.Thread(target=heartbeat, daemon=True).start()
    try:
        sd = fn(cfg, **kw)
    finally:
        if stop is not None:
            stop.set()
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save(sd, tmp)
    os.replace(tmp, path)
    if mine:
        try:
            os.unlink(lock)
        except OSError:
            pass
    return sd


def state_checksum(sd: dict) -> str:
    """crc32 over all tensors' bytes in key order (pins fixtures to the generator version)."""
    c = 0
    for k in sorted(sd):
        c = zlib.crc32(sd[k].detach().cpu().contiguous().numpy().tobytes(), c)
    return f"{c:08x}"
