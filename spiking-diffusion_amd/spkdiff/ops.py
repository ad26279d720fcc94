"""Tensor-level wrappers over the C-ABI: validate, allocate outputs with torch, launch on torch's current stream.

PyTorch is plumbing here (device memory, streams); every operation is executed by ``libspkdiff.so``.
All tensors must live on a ROCm device ("cuda"); CPU tensors are rejected -- there is no fallback.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, lib

MODE_LIF, MODE_RAW, MODE_MEMOUT, MODE_MEAN = 0, 1, 2, 3
SPIKE_F32, SPIKE_U8, SPIKE_BITS = 0, 1, 2
MAX_T = 16


# ---------------------------------------------------------------------------------------------- active-set batch
# Set by the sampler around one denoiser call (``with ops.active_set(active, n_active)``): the per-step kernels then
# process only the images listed by spk_select_active -- ``active`` int32 [B] (slot -> image), ``n_active`` int32 [2] (count, work word),
# both on the device, so that a captured hipGraph replays with fresh lists.  None = every image (dense).
# ``need`` (optional, spk_select_needed): the positions of each active image the step will read, per layer depth; the
# layers that take position lists then compute only those.
ACTIVE = None
NEED = None


class active_set:
    def __init__(self, active, n_active, need=None):
        self.pair = None if active is None else (active, n_active)
        self.need = need if active is not None else None

    def __enter__(self):
        global ACTIVE, NEED
        self.prev, ACTIVE = ACTIVE, self.pair
        self.prev_need, NEED = NEED, self.need
        return self

    def __exit__(self, *exc):
        global ACTIVE, NEED
        ACTIVE = self.prev
        NEED = self.prev_need
        return False


def _n_dyn():
    return None if ACTIVE is None else ACTIVE[1].data_ptr()


# ---------------------------------------------------------------------------------------------- in-situ kernel timing
# bench.py switches this on to bracket named launches with HIP events recorded on the SAME stream the kernels
# are enqueued on (torch's current stream): TIMERS[tag] = [(start_event, end_event), ...].
TIMERS = None


class timed:
    __slots__ = ("tag", "e1")

    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        if TIMERS is not None and self.tag is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            TIMERS.setdefault(self.tag, []).append((e0, self.e1))
        return self

    def __exit__(self, *exc):
        if TIMERS is not None and self.tag is not None:
            self.e1.record()
        return False


def _stream(t: torch.Tensor):
    """HIP stream the launch goes to: torch's current stream of the tensor's device.  A launch is issued in the calling
    thread's current device context, so a tensor on another GPU is refused rather than launched into the wrong context
    (this package runs one process per GPU; ``with torch.cuda.device(t.device):`` around the call is the way to address
    a second device from one process)."""
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"spkdiff: tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "wrap the call in `with torch.cuda.device(tensor.device):`")
    return torch.cuda.current_stream(t.device).cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"spkdiff: {name} is on '{t.device}'. The HIP kernels are the implementation; "
                           "there is no CPU path (move the module / tensors to a ROCm device).")
    if dtype is not None and t.dtype != dtype:
        raise NotImplementedError(f"spkdiff: {name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def conv_out_size(n, k, stride, pad, transposed=False, out_pad=0):
    return lib.spk_conv_out_size(n, k, stride, pad, int(transposed), out_pad)


def clock_probe(device, target_ms: float = 4.0):
    """Shader clock [GHz] this device holds under a block-scaled fp6 x fp4 MFMA load (spk_clock_probe): median over
    the workgroups of d(s_memtime) / d(s_memrealtime) * 0.1.  Synchronises.  Measurement aid for bench.py."""
    nblk = torch.cuda.get_device_properties(device).multi_processor_count
    out = torch.zeros((nblk, 4), dtype=torch.int64, device=device)
    iters = max(1000, int(target_ms * 1e-3 / (4 * 33 / 2.0e9)))
    check(lib.spk_clock_probe(_p(out), nblk, iters, _stream(out)), "spk_clock_probe")
    o = out.cpu().double()
    ghz = (o[:, 0] / o[:, 1].clamp(min=1)) * 0.1
    cyc_per_mfma = o[:, 0] / o[:, 2].clamp(min=1)
    return {"ghz_median": float(ghz.median()), "ghz_min": float(ghz.min()), "ghz_max": float(ghz.max()),
            "cycles_per_mfma": float(cyc_per_mfma.median()), "workgroups": int(nblk), "mfma_per_wave": int(4 * iters)}


# ---------------------------------------------------------------------------------------------- neuron
def lif_fwd(x_seq: torch.Tensor, v: torch.Tensor, tau=2.0, v_threshold=1.0, v_reset=0.0, spike_dtype=SPIKE_F32):
    """x_seq [T, ...] fp32, v [...] fp32 (updated in place) -> spikes. Mirrors the cupy-plugin contract
    SJ/activation_based/neuron.py:954-966 (x_seq.flatten(1), v.flatten(0))."""
    x_seq = _dev(x_seq, "x_seq", torch.float32)
    v = _dev(v, "v", torch.float32)
    T = x_seq.shape[0]
    N = x_seq[0].numel()
    if v.numel() != N:
        raise ValueError(f"v has {v.numel()} elements, x_seq[0] has {N}")
    if spike_dtype == SPIKE_F32:
        out = torch.empty_like(x_seq)
    elif spike_dtype == SPIKE_U8:
        out = torch.empty(x_seq.shape, dtype=torch.uint8, device=x_seq.device)
    elif spike_dtype == SPIKE_BITS:
        out = torch.empty((T, (N + 63) // 64), dtype=torch.int64, device=x_seq.device)
    else:
        raise NotImplementedError(spike_dtype)
    check(lib.spk_lif_fwd(_p(x_seq), _p(v), _p(out), T, N, float(tau), float(v_threshold), float(v_reset),
                          spike_dtype, _stream(x_seq)), "spk_lif_fwd")
    return out


def lif_fwd_ex(x_seq: torch.Tensor, v: torch.Tensor, tau=2.0, v_threshold=1.0, v_reset=0.0, soft_reset=False, decay_input=True,
               want_v_seq=False):
    """The reference neuron's other eval forms (spk_lif_fwd_ex): soft reset, decay_input=False, v_seq.  Returns
    (spikes fp32 like x_seq, v_seq or None); v updated in place."""
    x_seq = _dev(x_seq, "x_seq", torch.float32)
    v = _dev(v, "v", torch.float32)
    T, N = x_seq.shape[0], x_seq[0].numel()
    if v.numel() != N:
        raise ValueError(f"v has {v.numel()} elements, x_seq[0] has {N}")
    out = torch.empty_like(x_seq)
    v_seq = torch.empty_like(x_seq) if want_v_seq else None
    check(lib.spk_lif_fwd_ex(_p(x_seq), _p(v), _p(out), _p(v_seq), T, N, float(tau), float(v_threshold),
                             0.0 if v_reset is None else float(v_reset), int(bool(soft_reset)), int(bool(decay_input)),
                             _stream(x_seq)), "spk_lif_fwd_ex")
    return out, v_seq


# ---------------------------------------------------------------------------------------------- stateless layers
def lif_train_fwd(x_seq: torch.Tensor, v_init: torch.Tensor, tau=2.0, v_threshold=1.0, v_reset=0.0):
    """Training-mode multi-step LIF (hard reset, decay_input). Returns (spike_seq, h_seq, v_last), all fp32."""
    x = _dev(x_seq, "x_seq", torch.float32)
    v0 = _dev(v_init, "v", torch.float32)
    T = x.shape[0]
    N = x[0].numel()
    if v0.numel() != N:
        raise ValueError(f"v has {v0.numel()} elements, x_seq[0] has {N}")
    h = torch.empty_like(x)
    s = torch.empty_like(x)
    v_last = torch.empty_like(v0)
    check(lib.spk_lif_train_fwd(_p(x), _p(v0), _p(h), _p(s), _p(v_last), T, N, float(tau), float(v_threshold),
                                float(v_reset), _stream(x)), "spk_lif_train_fwd")
    return s, h, v_last


def lif_train_bwd(grad_spike_seq, grad_v_last, h_seq, tau=2.0, v_threshold=1.0, v_reset=0.0, alpha=2.0,
                  detach_reset=False, need_grad_v=True):
    """BPTT of lif_train_fwd with the ATan surrogate. Returns (grad_x_seq, grad_v_init or None)."""
    gs = _dev(grad_spike_seq, "grad_spike_seq", torch.float32)
    h = _dev(h_seq, "h_seq", torch.float32)
    gv = None if grad_v_last is None else _dev(grad_v_last, "grad_v_last", torch.float32)
    T = h.shape[0]
    N = h[0].numel()
    gx = torch.empty_like(h)
    gv0 = torch.empty(h.shape[1:], dtype=torch.float32, device=h.device) if need_grad_v else None
    check(lib.spk_lif_train_bwd(_p(gs), _p(gv), _p(h), _p(gx), _p(gv0), T, N, float(tau), float(v_threshold),
                                float(v_reset), float(alpha), int(bool(detach_reset)), _stream(h)), "spk_lif_train_bwd")
    return gx, gv0


class LIFTrainFunction(torch.autograd.Function):
    """spike_seq, v_last = f(x_seq, v_init): the HIP counterpart of the reference's LIFNodeATGF
    (SJ/activation_based/auto_cuda/neuron_kernel.py:496-540), ATan surrogate."""

    @staticmethod
    def forward(ctx, x_seq, v_init, tau, v_threshold, v_reset, alpha, detach_reset):
        s, h, v_last = lif_train_fwd(x_seq, v_init, tau, v_threshold, v_reset)
        ctx.save_for_backward(h)
        ctx.cfg = (tau, v_threshold, v_reset, alpha, detach_reset)
        return s, v_last

    @staticmethod
    def backward(ctx, grad_s, grad_v_last):
        (h,) = ctx.saved_tensors
        tau, v_threshold, v_reset, alpha, detach_reset = ctx.cfg
        gs = grad_s if grad_s is not None else torch.zeros_like(h)
        gx, gv0 = lif_train_bwd(gs.contiguous(), None if grad_v_last is None else grad_v_last.contiguous(), h, tau,
                                v_threshold, v_reset, alpha, detach_reset, need_grad_v=ctx.needs_input_grad[1])
        return gx, gv0, None, None, None, None, None


def _cl5(x, name):
    """[T,B,C,H,W] fp32 device tensor -> the same values with channels-last memory ([T][B][H][W][C]); no copy if it
    already is (the output of a channels-last library convolution viewed as [T,B,...])."""
    if not x.is_cuda:
        raise RuntimeError(f"spkdiff: {name} is on '{x.device}'; there is no CPU path")
    if x.dtype != torch.float32:
        raise NotImplementedError(f"spkdiff: {name} must be float32, got {x.dtype}")
    if x.permute(0, 1, 3, 4, 2).is_contiguous():
        return x
    return x.flatten(0, 1).contiguous(memory_format=torch.channels_last).view(x.shape)


def _cl4(x, name):
    if x is None:
        return None
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError(f"spkdiff: {name} must be a float32 device tensor")
    return x if x.permute(0, 2, 3, 1).is_contiguous() else x.contiguous(memory_format=torch.channels_last)


def _empty_cl(shape, device):
    """Uninitialised fp32 [..., C, H, W] tensor whose memory is [...][H][W][C]."""
    perm = tuple(range(len(shape) - 3)) + (len(shape) - 2, len(shape) - 1, len(shape) - 3)
    inv = tuple(range(len(shape) - 3)) + (len(shape) - 1, len(shape) - 3, len(shape) - 2)
    return torch.empty(tuple(shape[i] for i in perm), dtype=torch.float32, device=device).permute(inv)


class BNLIFTrainFunction(torch.autograd.Function):
    """spike_seq, v_last = f(y_seq, gamma, beta, v_init): training-mode BatchNorm2d (batch statistics over T*B*H*W,
    running statistics updated in place) fused with the surrogate-gradient LIF -- the tail of one denoiser block in
    train() mode (SJ/activation_based/layer.py:458-465 + neuron.py:739-749,133-135).  y_seq [T,B,C,H,W] fp32.
    Only y and the two per-channel statistics are kept for the backward, which recomputes the membrane potentials.
    The kernels work on channels-last memory ([T][B][H][W][C]): inputs in another layout are converted, outputs and
    gradients are returned channels-last (what the library's NHWC convolutions consume without a transpose)."""

    @staticmethod
    def forward(ctx, y_seq, gamma, beta, v_init, running_mean, running_var, momentum, eps, tau, v_threshold, v_reset,
                alpha, detach_reset, want_c4=False):
        # want_c4: a THIRD output, the spikes as C4 records (the next block's exact MFMA forward reads them: no conversion launch)
        if y_seq.dim() != 5:
            raise ValueError(f'expected y_seq with shape [T, N, C, H, W], but got {tuple(y_seq.shape)}')
        y = _cl5(y_seq, "y_seq")
        T, B, C = int(y.shape[0]), int(y.shape[1]), int(y.shape[2])
        HW = int(y.shape[3] * y.shape[4])
        g = None if gamma is None else _dev(gamma, "gamma", torch.float32)
        b = None if beta is None else _dev(beta, "beta", torch.float32)
        v0 = _cl4(v_init, "v")
        if v0 is not None and tuple(v0.shape) != tuple(y.shape[1:]):
            raise RuntimeError(f"LIFNode state has shape {tuple(v0.shape)} but the input implies {tuple(y.shape[1:])}; "
                               "call functional.reset_net first")
        for name, r in (("running_mean", running_mean), ("running_var", running_var)):
            if r is not None and (not r.is_cuda or r.dtype != torch.float32 or not r.is_contiguous()):
                raise ValueError(f"{name} must be a contiguous fp32 device tensor (updated in place)")
        ws = torch.empty(int(lib.spk_bn_lif_train_ws_bytes(B, C, HW)), dtype=torch.uint8, device=y.device)
        s = _empty_cl(y.shape, y.device)
        v_last = _empty_cl(y.shape[1:], y.device)
        mean = torch.empty(C, dtype=torch.float32, device=y.device)
        invstd = torch.empty(C, dtype=torch.float32, device=y.device)
        c4 = None
        if want_c4 and C % 64 == 0 and C // 4 <= 256 and 256 % (C // 4) == 0 and all(
                t is None or t.data_ptr() % 16 == 0 for t in (y, s, v0, v_last)):
            c4 = torch.empty((B, C // 64, int(y.shape[3]), int(y.shape[4]), T, 32), dtype=C4_DTYPE, device=y.device)
        with timed("train.bn_lif_fwd"):
            check(lib.spk_bn_lif_train_fwd_c4(_p(y), _p(g), _p(b), _p(running_mean), _p(running_var), float(momentum),
                                              float(eps), _p(v0), _p(s), _p(v_last), _p(mean), _p(invstd), _p(c4), _p(ws),
                                              ws.numel(), T, B, C, HW, float(tau), float(v_threshold), float(v_reset),
                                              _stream(y)), "spk_bn_lif_train_fwd_c4")
        ctx.save_for_backward(y, g, b, mean, invstd, v0)
        ctx.cfg = (tau, v_threshold, v_reset, alpha, detach_reset)
        # v_last usually ends in lif.v and nowhere in the loss: without this autograd materialises a zero gradient for it on every
        # backward (a fill + a layout copy per block and iteration); the kernel takes a null pointer for "no gradient"
        ctx.set_materialize_grads(False)
        if not want_c4:
            return s, v_last
        if c4 is not None:
            ctx.mark_non_differentiable(c4)
        return s, v_last, c4

    @staticmethod
    def backward(ctx, grad_s, grad_v_last, *_):
        y, g, b, mean, invstd, v0 = ctx.saved_tensors
        tau, v_threshold, v_reset, alpha, detach_reset = ctx.cfg
        T, B, C = int(y.shape[0]), int(y.shape[1]), int(y.shape[2])
        HW = int(y.shape[3] * y.shape[4])
        # grad_s as autograd left it: a dense channels-last tensor, or -- in front of the denoiser's last layer -- ONE [B,C,H,W]
        # gradient broadcast over T (stride 0) and / or a channel slice of a wider channels-last tensor (cat(x5, x1) backward):
        # the kernel takes the step stride and the row pitch instead of an expanded copy (25.7 MB for conv5's spikes at B = 32)
        gs_ts, gs_pitch = B * HW * C, C
        gs = None
        if grad_s is None and grad_v_last is None:
            return (None,) * 14
        if grad_s is not None and grad_s.is_cuda and grad_s.dtype == torch.float32 and grad_s.dim() == 5:
            st = grad_s.stride()
            Hh, Ww = int(y.shape[3]), int(y.shape[4])
            pitch = st[4]
            if (st[2] == 1 and st[3] == Ww * pitch and st[1] == Hh * Ww * pitch and pitch >= C and pitch % 4 == 0 and
                    st[0] in (0, B * Hh * Ww * pitch) and st[0] % 4 == 0 and grad_s.data_ptr() % 16 == 0 and
                    (st[0] == 0 or pitch != C)):
                gs, gs_ts, gs_pitch = grad_s, int(st[0]), int(pitch)
        if gs is None:
            gs = _empty_cl(y.shape, y.device).zero_() if grad_s is None else _cl5(grad_s, "grad_spike_seq")
        gv = _cl4(grad_v_last, "grad_v_last")
        ws = torch.empty(int(lib.spk_bn_lif_train_ws_bytes(B, C, HW)), dtype=torch.uint8, device=y.device)
        gy = _empty_cl(y.shape, y.device)
        gg = torch.empty(C, dtype=torch.float32, device=y.device)
        gb = torch.empty(C, dtype=torch.float32, device=y.device)
        gv0 = _empty_cl(y.shape[1:], y.device) if (v0 is not None and ctx.needs_input_grad[3]) else None
        with timed("train.bn_lif_bwd"):
            check(lib.spk_bn_lif_train_bwd_strided(_p(gs), int(gs_ts), int(gs_pitch), _p(gv), _p(y), _p(g), _p(b), _p(mean),
                                                   _p(invstd), _p(v0), _p(gy), _p(gg), _p(gb), _p(gv0), _p(ws), ws.numel(), T, B,
                                                   C, HW, float(tau), float(v_threshold), float(v_reset), float(alpha),
                                                   int(bool(detach_reset)), _stream(y)), "spk_bn_lif_train_bwd_strided")
        return (gy, gg if g is not None else None, gb if b is not None else None, gv0) + (None,) * 10


class MaskedCEFunction(torch.autograd.Function):
    """loss = sum_b coef_b * sum_p ce[b, p] with ce the masked cross-entropy of R/snn_model/vq_diffusion.py:85-88
    (ignore_index = -1).  logits [B,K,h,w] fp32, target [B,1,h,w] (or [B,h,w]) fp32 token ids, coef [B] fp32.
    Forward and gradient come from one spk_masked_ce launch; the weighted sum over [B, hw] is a tiny torch reduction."""

    @staticmethod
    def forward(ctx, logits, target, coef):
        lg = _dev(logits, "logits", torch.float32)
        B, K = int(lg.shape[0]), int(lg.shape[1])
        HW = lg[0, 0].numel()
        tg = _dev(target, "target", torch.float32)
        cf = _dev(coef, "coef", torch.float32)
        if tg.numel() != B * HW or cf.numel() != B:
            raise ValueError("target must have B*h*w entries and coef B entries")
        ce = torch.empty((B, HW), dtype=torch.float32, device=lg.device)
        dl = torch.empty_like(lg) if ctx.needs_input_grad[0] else None
        check(lib.spk_masked_ce(_p(lg), _p(tg), _p(cf), _p(ce), _p(dl), B, K, HW, _stream(lg)), "spk_masked_ce")
        ctx.has_dl = dl is not None
        if dl is not None:
            ctx.save_for_backward(dl)          # survives retain_graph / a second backward like any saved tensor
        return (ce.sum(1) * cf).sum()

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.has_dl:
            return None, None, None
        (dl,) = ctx.saved_tensors
        return dl * grad_out, None, None


def masked_ce(logits, target):
    """Per-position masked cross-entropy [B, hw] (no gradient): the forward half of spk_masked_ce."""
    lg = _dev(logits, "logits", torch.float32)
    B, K = int(lg.shape[0]), int(lg.shape[1])
    HW = lg[0, 0].numel()
    tg = _dev(target, "target", torch.float32)
    ce = torch.empty((B, HW), dtype=torch.float32, device=lg.device)
    check(lib.spk_masked_ce(_p(lg), _p(tg), None, _p(ce), None, B, K, HW, _stream(lg)), "spk_masked_ce")
    return ce


def bn_prepare(gamma, beta, mean, var, eps):
    mean = _dev(mean, "running_mean", torch.float32)
    var = _dev(var, "running_var", torch.float32)
    C = mean.numel()
    a = torch.empty(C, dtype=torch.float32, device=mean.device)
    b = torch.empty_like(a)
    g = None if gamma is None else _dev(gamma.detach(), "bn.weight", torch.float32)
    be = None if beta is None else _dev(beta.detach(), "bn.bias", torch.float32)
    check(lib.spk_bn_prepare(_p(g), _p(be), _p(mean), _p(var), float(eps), _p(a), _p(b), C, _stream(mean)),
          "spk_bn_prepare")
    return a, b


def bn_eval(x, a, b):
    """x [M, C, H, W] (or [M, C, HW]) fp32."""
    x = _dev(x, "x", torch.float32)
    M, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    y = torch.empty_like(x)
    check(lib.spk_bn_eval_fwd(_p(x), _p(a), _p(b), _p(y), M, C, HW, _stream(x)), "spk_bn_eval_fwd")
    return y


def conv2d(x, w, bias, stride, pad):
    x = _dev(x, "x", torch.float32)
    w = _dev(w.detach(), "weight", torch.float32)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    M, Cin, H, W = x.shape
    Cout, Cin_w, k, k2 = w.shape
    if Cin_w != Cin or k != k2:
        raise ValueError(f"weight {tuple(w.shape)} does not match input channels {Cin} / square kernels only")
    Ho, Wo = conv_out_size(H, k, stride, pad), conv_out_size(W, k, stride, pad)
    y = torch.empty((M, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.spk_conv2d_fwd(_p(x), _p(w), _p(b), _p(y), M, Cin, H, W, Cout, k, stride, pad, _stream(x)),
          "spk_conv2d_fwd")
    return y


def conv_transpose2d(x, w, bias, stride, pad, out_pad):
    x = _dev(x, "x", torch.float32)
    w = _dev(w.detach(), "weight", torch.float32)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    M, Cin, H, W = x.shape
    Cin_w, Cout, k, k2 = w.shape
    if Cin_w != Cin or k != k2:
        raise ValueError(f"weight {tuple(w.shape)} does not match input channels {Cin} / square kernels only")
    Ho, Wo = conv_out_size(H, k, stride, pad, True, out_pad), conv_out_size(W, k, stride, pad, True, out_pad)
    y = torch.empty((M, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    check(lib.spk_conv_transpose2d_fwd(_p(x), _p(w), _p(b), _p(y), M, Cin, H, W, Cout, k, stride, pad, out_pad,
                                       _stream(x)), "spk_conv_transpose2d_fwd")
    return y


class ExactConvTrainFunction(torch.autograd.Function):
    """y = conv2d / conv_transpose2d(x, weight) + bias in training: the forward is this library's direct kernel (fp64
    accumulation: the correctly rounded exact result, the same numbers as the inference path and -- up to the reference's
    own fp32 round-off -- the reference's), the backward is the framework's convolution backward.  Used for the small
    layers (VQ-VAE, the denoiser's first convolution); the denoiser's spike-input layers use SpikeConvTrainFunction."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, transposed, out_pad):
        if transposed:
            y = conv_transpose2d(x, weight, bias, stride, pad, out_pad)
        else:
            y = conv2d(x, weight, bias, stride, pad)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (int(stride), int(pad), bool(transposed), int(out_pad), bias is not None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight = ctx.saved_tensors
        stride, pad, transposed, out_pad, has_bias = ctx.cfg
        cout = int(weight.shape[1] if transposed else weight.shape[0])
        if (NATIVE_WGRAD and not transposed and not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and stride == 1 and
                pad == 1 and tuple(weight.shape[2:]) == (3, 3) and int(weight.shape[1]) <= 4 and x.dim() == 4 and
                lib.spk_conv3x3_wgrad_small_ws_bytes(int(x.shape[0]), int(x.shape[2]), int(x.shape[3]), cout,
                                                     int(weight.shape[1])) > 0):
            # few input channels, dense input, no input gradient wanted: the denoiser's first layer (spk_conv3x3_wgrad_small:
            # maps up to 8x8 -- larger ones answer the workspace query with -1 and take the framework's operator below)
            return (None,) + conv3x3_wgrad_small(grad_y, x, weight, has_bias and ctx.needs_input_grad[2]) + (None,) * 4
        needs = (bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]), bool(has_bias and ctx.needs_input_grad[2]))
        if conv_train_supported(x.shape, weight, stride, pad, transposed, out_pad, needs[0]):
            # the native data / weight gradients of csrc/conv_train.hip (the VQ-VAE's stride-2, 1x1 and transposed layers)
            return conv_train_backward(grad_y, x, weight, stride, pad, transposed, out_pad, needs) + (None,) * 4
        gi, gw, gb = torch.ops.aten.convolution_backward(
            grad_y.contiguous(), x.contiguous(), weight, [cout], [stride, stride], [pad, pad], [1, 1], transposed,
            [out_pad, out_pad], 1, list(needs))
        return gi, gw, gb, None, None, None, None


# ---- native training convolutions (csrc/conv_train.hip): the VQ-VAE's stride-2 / 1x1 / transposed layers ---------------------
NATIVE_TRAIN_CONV = True        # False: the framework's (library) convolution backward, as in rounds 3-4
NATIVE_TRAIN_FORWARD = True     # False: exact direct kernel / library forward by ops.EXACT_TRAIN_FORWARD_MACS (the parity runs)


def _conv_w_strides(weight, transposed):
    """(tap, cin, cout) element strides of a convolution weight as stored ([Cout,Cin,k,k] / [Cin,Cout,k,k], contiguous or
    channels-last), or None when the taps are not k*k equally spaced elements."""
    st, k = weight.stride(), int(weight.shape[2])
    if weight.dim() != 4 or weight.shape[2] != weight.shape[3] or (k > 1 and st[2] != k * st[3]):
        return None
    tap = int(st[3]) if k > 1 else 0
    return (tap, int(st[0]), int(st[1])) if transposed else (tap, int(st[1]), int(st[0]))


def _conv_train_geometry(xshape, weight, stride, pad, transposed, out_pad):
    N, Cin, Hi, Wi = (int(v) for v in xshape)
    k = int(weight.shape[2])
    Cout = int(weight.shape[1] if transposed else weight.shape[0])
    Ho, Wo = (conv_out_size(Hi, k, stride, pad, transposed, out_pad), conv_out_size(Wi, k, stride, pad, transposed, out_pad))
    return N, Cin, Hi, Wi, Cout, Ho, Wo, k


def conv_train_supported(xshape, weight, stride, pad, transposed, out_pad, need_gi=True, forward=False):
    """Do the native training kernels take this layer (backward: data gradient if wanted + weight gradient; forward=True: the
    forward as well)?"""
    if not NATIVE_TRAIN_CONV or len(xshape) != 4 or weight.dim() != 4 or not weight.is_cuda or weight.dtype != torch.float32:
        return False
    if _conv_w_strides(weight, transposed) is None or int(weight.shape[1 if not transposed else 0]) != int(xshape[1]):
        return False
    N, Cin, Hi, Wi, Cout, Ho, Wo, k = _conv_train_geometry(xshape, weight, stride, pad, transposed, out_pad)
    if transposed and (out_pad >= stride or (Hi - 1) * stride - 2 * pad + k + out_pad != Ho):
        return False
    f_fwd, f_bwd = (1, 0) if transposed else (0, 1)
    if forward and not lib.spk_conv_train_gather_supported(Cin, Cout, k, stride, f_fwd):
        return False
    if need_gi and not lib.spk_conv_train_gather_supported(Cout, Cin, k, stride, f_bwd):
        return False
    if transposed:          # u = gy (finer grid), v = the input
        return lib.spk_conv_train_wgrad_ws_bytes(N, Hi, Wi, Cout, Cin, k) > 0
    return lib.spk_conv_train_wgrad_ws_bytes(N, Ho, Wo, Cin, Cout, k) > 0


def conv_train_forward(x, weight, bias, stride, pad, transposed, out_pad):
    """y [N,Cout,Ho,Wo] (channels-last memory) = conv2d / conv_transpose2d(x, weight) + bias on the fp32 matrix cores
    (spk_conv_train_gather)."""
    xin = _cl4(x.detach(), "x")
    w = weight.detach()
    N, Cin, Hi, Wi, Cout, Ho, Wo, k = _conv_train_geometry(xin.shape, w, stride, pad, transposed, out_pad)
    tap, s_ci, s_co = _conv_w_strides(w, transposed)
    y = _empty_cl((N, Cout, Ho, Wo), xin.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    with timed("train.conv_fwd"):
        check(lib.spk_conv_train_gather(_p(xin), _p(w), _p(b), _p(y), N, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, pad,
                                        1 if transposed else 0, tap, s_ci, s_co, _stream(xin)), "spk_conv_train_gather")
    return y


def conv_train_backward(grad_y, x, weight, stride, pad, transposed, out_pad, needs):
    """(gi, gw, gb) of a convolution layer from gy (any layout; converted to channels-last if it is not), the saved input and
    the weight: spk_conv_train_gather in the transposed role + spk_conv_train_wgrad.  gw in the memory format of ``weight``."""
    gy = _cl4(grad_y, "grad_y")
    xin = _cl4(x.detach(), "x")
    w = weight.detach()
    N, Cin, Hi, Wi, Cout, Ho, Wo, k = _conv_train_geometry(xin.shape, w, stride, pad, transposed, out_pad)
    tap, s_ci, s_co = _conv_w_strides(w, transposed)
    st = _stream(gy)
    gi = gw = gb = None
    if needs[0]:
        gi = _empty_cl((N, Cin, Hi, Wi), gy.device)
        with timed("train.conv_bwd_data"):
            check(lib.spk_conv_train_gather(_p(gy), _p(w), None, _p(gi), N, Ho, Wo, Cout, Hi, Wi, Cin, k, stride, pad,
                                            0 if transposed else 1, tap, s_co, s_ci, st), "spk_conv_train_gather")
    if needs[1] or needs[2]:
        gw = torch.empty_like(w)                     # (the memory format of a dense weight is preserved; the kernel takes gw's own strides)
        g_tap, g_ci, g_co = _conv_w_strides(gw, transposed)
        gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if needs[2] else None
        if transposed:
            u, v, Hu, Wu, Cu, Hv, Wv, Cv, g_u, g_v, bf = gy, xin, Ho, Wo, Cout, Hi, Wi, Cin, g_co, g_ci, 2
        else:
            u, v, Hu, Wu, Cu, Hv, Wv, Cv, g_u, g_v, bf = xin, gy, Hi, Wi, Cin, Ho, Wo, Cout, g_ci, g_co, 1
        nb = int(lib.spk_conv_train_wgrad_ws_bytes(N, Hv, Wv, Cu, Cv, k))
        ws = torch.empty(nb, dtype=torch.uint8, device=gy.device)
        with timed("train.conv_bwd_weight"):
            check(lib.spk_conv_train_wgrad(_p(u), _p(v), _p(ws), nb, _p(gw), _p(gb), N, Hu, Wu, Cu, Hv, Wv, Cv, k, stride, pad,
                                           g_tap, g_u, g_v, bf if gb is not None else 0, st), "spk_conv_train_wgrad")
        if not needs[1]:
            gw = None
    return gi, gw, gb


class NativeConvTrainFunction(torch.autograd.Function):
    """y = conv2d / conv_transpose2d(x, weight) + bias in training, forward and backward on this library's fp32 matrix-core
    kernels (csrc/conv_train.hip): the stride-2, 1x1 and transposed layers of the spiking VQ-VAE
    (R/snn_model/vae_model.py:101-159; cuDNN in the reference).  Tensors are channels-last in memory (what the BatchNorm+LIF
    block tails produce and consume); another layout is converted."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, transposed, out_pad):
        y = conv_train_forward(x, weight, bias, stride, pad, transposed, out_pad)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (int(stride), int(pad), bool(transposed), int(out_pad), bias is not None)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight = ctx.saved_tensors
        stride, pad, transposed, out_pad, has_bias = ctx.cfg
        needs = (bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1]), bool(has_bias and ctx.needs_input_grad[2]))
        return conv_train_backward(grad_y, x, weight, stride, pad, transposed, out_pad, needs) + (None,) * 4


def conv3x3_wgrad_small(grad_y, x, weight, want_bias):
    """(gw, gb) of a 3x3 / s1 / p1 convolution with <= 4 input channels from gy [N,Cout,H,W] and the dense input x [N,Cin,H,W]
    (spk_conv3x3_wgrad_small); gw comes in the memory format of ``weight`` (channels-last while training)."""
    N, Cout, H, W = (int(v) for v in grad_y.shape)
    Cin = int(x.shape[1])
    gy = grad_y if grad_y.is_contiguous(memory_format=torch.channels_last) else grad_y.contiguous(memory_format=torch.channels_last)
    xin = _dev(x.detach(), "x", torch.float32)
    cl = weight.dim() == 4 and weight.is_contiguous(memory_format=torch.channels_last) and not weight.is_contiguous()
    gw = torch.empty_like(weight, memory_format=torch.channels_last if cl else torch.contiguous_format)
    gb = torch.empty(Cout, dtype=torch.float32, device=gy.device) if want_bias else None
    nb = int(lib.spk_conv3x3_wgrad_small_ws_bytes(N, H, W, Cout, Cin))
    if nb <= 0:
        raise NotImplementedError(f"spk_conv3x3_wgrad_small: unsupported shape N={N} H={H} W={W} Cout={Cout} Cin={Cin} "
                                  "(maps of at most 64 positions, at most 4 input channels)")
    ws = torch.empty(nb, dtype=torch.uint8, device=gy.device)
    with timed("train.conv_bwd_weight"):
        check(lib.spk_conv3x3_wgrad_small(_p(gy), _p(xin), _p(ws), nb, _p(gw), _p(gb), N, H, W, Cout, Cin, int(cl), _stream(gy)),
              "spk_conv3x3_wgrad_small")
    return gw, gb


# Training forward of a stand-alone convolution: exact direct kernel up to this many multiply-accumulates per call (its fp64
# accumulation is not a matrix-core kernel), the library operator beyond.  Round 4: 4e9 -> 1e8 -- at the reference's batch of 32
# the VQ-VAE's 64 -> 32 transposed convolution (1.85e9) took 2.4 ms of a 5.4 ms training iteration in the exact kernel, the
# 16 -> 64 one (2.3e8) 0.33 ms (profiles/r4_train_vqvae.md)
EXACT_TRAIN_FORWARD_MACS = 100_000_000


def memout(x_seq, coef):
    x_seq = _dev(x_seq, "x_seq", torch.float32)
    coef = _dev(coef, "coef", torch.float32)
    T = x_seq.shape[0]
    if coef.numel() != T:
        raise RuntimeError(f"The size of tensor a ({T}) must match the size of tensor b ({coef.numel()}) at "
                           "non-singleton dimension 0")      # what torch raises for x*coef in the reference
    out = torch.empty(x_seq.shape[1:], dtype=torch.float32, device=x_seq.device)
    check(lib.spk_memout_fwd(_p(x_seq), _p(coef), _p(out), T, out.numel(), _stream(x_seq)), "spk_memout_fwd")
    return out


class MemoutFunction(torch.autograd.Function):
    """Differentiable MembraneOutputLayer (training path): forward = spk_memout_fwd, backward dL/dx[t] = dL/dout * coef[t]
    (one broadcast multiply)."""

    @staticmethod
    def forward(ctx, x_seq, coef):
        ctx.save_for_backward(coef)
        return memout(x_seq, coef)

    @staticmethod
    def backward(ctx, grad_out):
        (coef,) = ctx.saved_tensors
        return grad_out.unsqueeze(0) * coef.view((-1,) + (1,) * grad_out.dim()), None


def _dense_tn(x, name):
    """[T, ...] fp32 device tensor whose memory is T dense planes (any element order inside a plane)."""
    if not x.is_cuda:
        raise RuntimeError(f"spkdiff: {name} is on '{x.device}'; there is no CPU path")
    if x.dtype != torch.float32:
        raise NotImplementedError(f"spkdiff: {name} must be float32, got {x.dtype}")
    n = x[0].numel()
    if x.shape[0] > 1 and x.stride(0) != n:
        return x.contiguous()
    if not torch.empty_like(x).stride() == x.stride():       # not a dense permutation: copy
        return x.contiguous()
    return x


def psp(x_seq, tau_s=2.0, backward=False):
    x = _dense_tn(x_seq, "inputs")
    out = torch.empty_like(x)
    check(lib.spk_psp(_p(x), _p(out), int(x.shape[0]), x[0].numel(), float(tau_s), int(backward), _stream(x)), "spk_psp")
    return out


class PSPFunction(torch.autograd.Function):
    """syns = PSP(inputs) (R/snn_model/snn_layers.py:12-26) with its adjoint as the backward, both spk_psp."""

    @staticmethod
    def forward(ctx, x_seq, tau_s):
        ctx.tau_s = tau_s
        return psp(x_seq, tau_s, False)

    @staticmethod
    def backward(ctx, grad_out):
        return psp(grad_out, ctx.tau_s, True), None


# ---------------------------------------------------------------------------------------------- layouts
def spikes_to_ptc(s, chunk=None):
    """fp32 [T,B,C,H,W] -> u8 [B,H,W,T,C] (plain PTC) or, with ``chunk``, CPTC [B,C/chunk,H,W,T,chunk]."""
    s = _dev(s, "spikes", torch.float32)
    T, B, C, H, W = s.shape
    if chunk is None:
        o = torch.empty((B, H, W, T, C), dtype=torch.uint8, device=s.device)
    else:
        o = torch.empty((B, C // chunk, H, W, T, chunk), dtype=torch.uint8, device=s.device)
    check(lib.spk_spikes_to_ptc(_p(s), _p(o), T, B, C, H * W, chunk or C, _stream(s)), "spk_spikes_to_ptc")
    return o


def ptc_to_spikes(p):
    """u8 [B,H,W,T,C] or CPTC [B,C/chunk,H,W,T,chunk] (or int8-tagged C4) -> fp32 [T,B,C,H,W]."""
    if p.dtype == C4_DTYPE:
        return s32_to_spikes(p) if p.shape[-1] == 16 else c4_to_spikes(p)
    p = _dev(p, "ptc", torch.uint8)
    if p.dim() == 6:
        B, nch, H, W, T, chunk = p.shape
        C = nch * chunk
    else:
        B, H, W, T, C = p.shape
        chunk = C
    o = torch.empty((T, B, C, H, W), dtype=torch.float32, device=p.device)
    check(lib.spk_ptc_to_spikes(_p(p), _p(o), T, B, C, H * W, chunk, _stream(p)), "spk_ptc_to_spikes")
    return o


def count_spikes(t: torch.Tensor):
    """Spike statistics of a tensor in any storage format of this library, counted on the device (spk_count_spikes):
    dict(total, t0, numel, numel_t0, binary).  fp32 [T, ...] (the reference interface; ``binary`` = every nonzero entry is
    exactly 1.0), u8 PTC [B,H,W,T,C] / CPTC [B,C/ch,H,W,T,ch], int8-tagged C4 / S32 [B,C/rec_ch,H,W,T,rec]."""
    if not t.is_cuda:
        raise RuntimeError(f"spkdiff: tensor on '{t.device}'; there is no CPU path")
    t = t if t.is_contiguous() else t.contiguous()
    out = torch.empty(3, dtype=torch.int64, device=t.device)
    if t.dtype == torch.float32:
        T = int(t.shape[0])
        n = t.numel()
        inner, kind, numel = n // T, 2, n
    elif t.dtype == torch.uint8:
        T, rec = int(t.shape[-2]), int(t.shape[-1])
        if rec % 4:
            raise NotImplementedError("count_spikes: PTC records must be a multiple of 4 bytes")
        n, inner, kind, numel = t.numel() // 4, rec // 4, 0, t.numel()
    elif t.dtype == C4_DTYPE:
        T, rec = int(t.shape[-2]), int(t.shape[-1])
        n, inner, kind, numel = t.numel() // 4, rec // 4, 1, t.numel() * 2
    else:
        raise NotImplementedError(t.dtype)
    check(lib.spk_count_spikes(_p(t), n, inner, T, kind, _p(out), _stream(t)), "spk_count_spikes")
    tot, t0, ones = (int(v) for v in out.tolist())
    return {"total": tot, "t0": t0, "numel": numel, "numel_t0": numel // T, "binary": kind != 2 or tot == ones}


# ---------------------------------------------------------------------------------------------- fused conv
def pack_conv_weight(w, transposed):
    w = _dev(w.detach(), "weight", torch.float32)
    if transposed:
        Cin, Cout, k, k2 = w.shape
    else:
        Cout, Cin, k, k2 = w.shape
    if k != k2:
        raise NotImplementedError("square kernels only")
    out = torch.empty((k * k, Cin, Cout), dtype=torch.float32, device=w.device)
    check(lib.spk_pack_conv_weight(_p(w), _p(out), Cout, Cin, k, int(transposed), _stream(w)), "spk_pack_conv_weight")
    return out


IN_PTC, IN_TINV, IN_SEQ = 0, 1, 2
# "C4" spike tensors (fp4 e2m1 nibbles, 64 channels per 32-byte record: [B, C/64, H, W, 16, 32]) carry dtype int8 so that
# they cannot be mistaken for the u8 CPTC layout of the same shape
CHUNK_C4 = -64
STEP_TAIL_MAX_K = 512        # csrc/step_tail.hip TK_MAX: classes the fused reverse-step tail takes (four 16-channel groups per wave)
VQ_TRAIN_MAX_D = 64          # csrc/vq_train.hip VT_MAX_D: the fused VQ training operators keep one code vector per thread
CHUNK_S32 = -32          # "S32": the same nibbles in 32-channel records [B, C/32, H, W, 16, 16] (fp6v2 kernel)
C4_DTYPE = torch.int8


def conv_fused(in0, w_packed, bias, *, in_kind, T, mode, k, stride, pad, transposed=False, out_pad=0, in1=None,
               bn_a=None, bn_b=None, v=None, want_ptc=False, want_f32=False, want_pre=False, want_u8=False,
               coef=None, apply_tanh=False, out_ptc=None, out_f32=None, chunk_out=None, want_counts=False):
    """Launch spk_conv_fused_fwd. Returns dict(ptc=, f32=, pre=, u8=).

    in_kind IN_PTC: in0 u8 [B,H,W,T,C0] (+ in1 [B,H,W,T,C1]);  IN_TINV: in0 fp32 [B,C0,H,W];
    IN_SEQ: in0 fp32 [T,B,C0,H,W] (any values)."""
    dev = in0.device
    chunk0 = chunk1 = 0
    if in_kind == IN_PTC:
        in0 = _dev(in0, "in0", torch.uint8)
        if in0.dim() == 6:                                   # CPTC [B, C/chunk, H, W, T, chunk]
            B, nch, H, W, T_in, chunk0 = in0.shape
            C0 = nch * chunk0
        else:
            B, H, W, T_in, C0 = in0.shape
        if T_in != T:
            raise ValueError("T mismatch")
    elif in_kind == IN_TINV:
        in0 = _dev(in0, "in0", torch.float32)
        B, C0, H, W = in0.shape
    else:
        in0 = _dev(in0, "in0", torch.float32)
        T_in, B, C0, H, W = in0.shape
        if T_in != T:
            raise ValueError("T mismatch")
    if T > MAX_T:
        raise NotImplementedError(f"fused kernels keep T <= {MAX_T} steps in registers, got T={T}")
    C1 = 0
    if in1 is not None:
        in1 = _dev(in1, "in1", torch.uint8)
        if in1.dim() == 6:
            chunk1 = in1.shape[-1]
            C1 = in1.shape[1] * chunk1
        else:
            C1 = in1.shape[-1]
    kk, Cin, Cout = w_packed.shape
    if Cin != C0 + C1 or kk != k * k:
        raise ValueError(f"packed weight {tuple(w_packed.shape)} does not match Cin={C0 + C1}, k={k}")
    Ho, Wo = conv_out_size(H, k, stride, pad, transposed, out_pad), conv_out_size(W, k, stride, pad, transposed, out_pad)
    res = {"ptc": None, "f32": None, "pre": None, "u8": None, "cnt": None}
    if mode == MODE_LIF:
        if want_counts:
            res["cnt"] = torch.empty((B, Cout // 32, Ho, Wo, 32), dtype=torch.uint8, device=dev)
        if want_ptc:
            if chunk_out == CHUNK_C4:                        # nibble-packed fp4 spikes, tagged by dtype int8
                res["ptc"] = torch.empty((B, Cout // 64, Ho, Wo, T, 32), dtype=C4_DTYPE, device=dev)
            elif chunk_out == CHUNK_S32:
                res["ptc"] = torch.empty((B, Cout // 32, Ho, Wo, T, 16), dtype=C4_DTYPE, device=dev)
            else:
                shape = (B, Ho, Wo, T, Cout) if not chunk_out else (B, Cout // chunk_out, Ho, Wo, T, chunk_out)
                res["ptc"] = out_ptc if out_ptc is not None else torch.empty(shape, dtype=torch.uint8, device=dev)
        if want_f32:
            res["f32"] = out_f32 if out_f32 is not None else torch.empty((T, B, Cout, Ho, Wo), dtype=torch.float32, device=dev)
        if want_pre:
            shape = (B, Cout, Ho, Wo) if in_kind == IN_TINV else (T, B, Cout, Ho, Wo)
            res["pre"] = torch.empty(shape, dtype=torch.float32, device=dev)
    elif mode == MODE_RAW:
        res["f32"] = out_f32 if out_f32 is not None else torch.empty((T, B, Cout, Ho, Wo), dtype=torch.float32, device=dev)
    else:
        res["f32"] = out_f32 if out_f32 is not None else torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=dev)
        if want_u8:
            res["u8"] = torch.empty((B, Cout, Ho, Wo), dtype=torch.uint8, device=dev)
    if v is not None:
        v = _dev(v, "v", torch.float32)
        if v.numel() != B * Cout * Ho * Wo:
            raise ValueError("membrane potential tensor has the wrong size")
    check(lib.spk_conv_fused_fwd(
        _p(in0), _p(in1), C0, C1, in_kind, _p(w_packed), _p(bias), _p(bn_a), _p(bn_b), _p(v), _p(res["ptc"]),
        _p(res["f32"]), _p(res["pre"]), _p(res["u8"]), _p(coef), int(apply_tanh), mode, T, B, H, W, Cout, k, stride,
        pad, int(transposed), out_pad, chunk0, chunk1, chunk_out or 0, _p(res["cnt"]), _n_dyn(), _stream(in0)),
        "spk_conv_fused_fwd")
    return res


# ---------------------------------------------------------------------------------------------- MFMA denoiser convs
def den_mfma_supported(Cout, Cin, k, stride, pad, T, H, W):
    ntiles = (H * W + 1) // 2
    nt = (ntiles + 3) // 4
    lds = 2 * ((H + 2) * (W + 2) * 512 + 18432)
    return (k == 3 and stride == 1 and pad == 1 and T == 16 and Cout % 32 == 0 and Cin % 32 == 0
            and nt <= 8 and lds <= 160 * 1024)


def den_pack_weight_i8(w, bias, pad_cout=False):
    """[Cout,Cin,3,3] fp32 -> (int8 digit planes, fp64 scale [Cout], fp64 bias [Cout]).  pad_cout: output channels are
    zero-padded to the next multiple of 16 first (the logits layer for any --codebook_size, R/main.py:58: the kernels that read
    these planes take 16-channel column groups; scale / bias then have ceil16(Cout) entries, the extra ones are never read back)."""
    w = _dev(w.detach(), "weight", torch.float32)
    Cout, Cin = w.shape[0], w.shape[1]
    if pad_cout and Cout % 16:
        kp = (Cout + 15) // 16 * 16
        w = torch.cat([w, w.new_zeros((kp - Cout,) + tuple(w.shape[1:]))], 0).contiguous()
        if bias is not None:
            bias = torch.cat([bias.detach(), bias.detach().new_zeros(kp - Cout)], 0)
        Cout = kp
    nbytes = lib.spk_den_packed_weight_bytes(Cout, Cin)
    if nbytes < 0:
        raise NotImplementedError("spkdiff: MFMA conv needs Cout % 16 == 0 and Cin % 32 == 0")
    wq = torch.empty(nbytes, dtype=torch.int8, device=w.device)
    scale = torch.empty(Cout, dtype=torch.float64, device=w.device)
    bias_d = torch.empty(Cout, dtype=torch.float64, device=w.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    check(lib.spk_den_pack_weight_i8(_p(w), _p(b), _p(wq), _p(scale), _p(bias_d), Cout, Cin, _stream(w)),
          "spk_den_pack_weight_i8")
    return wq, scale, bias_d


def den_conv3x3_mfma(in0, packed, Cout, *, mode, in1=None, bn_a=None, bn_b=None, v=None, out=None, want_counts=False):
    """in0/in1: CPTC u8 [B, C/32, H, W, 16, 32]. mode LIF -> CPTC spikes [B, Cout/32, H, W, 16, 32]
    (or (spikes, counts u8 [B, Cout/32, H, W, 32]) with want_counts); mode MEAN -> fp32 [B, Cout, H, W]."""
    in0 = _dev(in0, "in0", torch.uint8)
    B, nch0, H, W, T, chunk = in0.shape
    if chunk != 32:
        raise ValueError("MFMA conv reads 32-channel chunked spikes")
    nch1 = 0
    if in1 is not None:
        in1 = _dev(in1, "in1", torch.uint8)
        nch1 = in1.shape[1]
    wq, scale, bias_d = packed
    out_c = out_f = None
    if mode == MODE_LIF:
        out_c = out if out is not None else torch.empty((B, Cout // 32, H, W, T, 32), dtype=torch.uint8, device=in0.device)
    else:
        out_f = out if out is not None else torch.empty((B, Cout, H, W), dtype=torch.float32, device=in0.device)
    cnt = None
    if want_counts and mode == MODE_LIF:
        cnt = torch.empty((B, Cout // 32, H, W, 32), dtype=torch.uint8, device=in0.device)
    check(lib.spk_den_conv3x3_mfma(_p(in0), nch0, _p(in1), nch1, _p(wq), _p(scale), _p(bias_d), _p(bn_a), _p(bn_b),
                                   _p(v), _p(out_c), _p(cnt), _p(out_f), mode, T, B, H, W, Cout, _n_dyn(),
                                   _stream(in0)), "spk_den_conv3x3_mfma")
    if mode == MODE_LIF:
        return (out_c, cnt) if want_counts else out_c
    return out_f


def den_conv3x3_counts(cnt0, packed, Cout, T, cnt1=None):
    """Time-collapsed conv6: cnt0/cnt1 u8 spike counts [B, C/32, H, W, 32] -> logits fp32 [B, Cout, H, W].  ``packed`` may hold
    more (zero-padded) output channels than Cout (den_pack_weight_i8(pad_cout=True)): the kernel computes all of them and the
    first Cout are returned."""
    cnt0 = _dev(cnt0, "cnt0", torch.uint8)
    B, nch0, H, W, ck = cnt0.shape
    if ck != 32:
        raise ValueError("counts are 32-channel chunked")
    nch1 = 0
    if cnt1 is not None:
        cnt1 = _dev(cnt1, "cnt1", torch.uint8)
        nch1 = cnt1.shape[1]
    wq, scale, bias_d = packed
    kp = int(scale.numel())
    if kp < Cout or kp - Cout >= 16:
        raise ValueError("packed logits weights do not belong to a layer of Cout output channels")
    out = torch.empty((B, kp, H, W), dtype=torch.float32, device=cnt0.device)
    check(lib.spk_den_conv3x3_counts_mfma(_p(cnt0), nch0, _p(cnt1), nch1, _p(wq), _p(scale), _p(bias_d), _p(out), T, B,
                                          H, W, kp, _n_dyn(), _stream(cnt0)), "spk_den_conv3x3_counts_mfma")
    return out if kp == Cout else out[:, :Cout].contiguous()


# ------------------------------------------------------------------------------- fp6/fp4 block-scaled MFMA denoiser convs
def den_fp6_supported(Cout, Cin, k, stride, pad, T, H, W):
    ntiles = (H * W + 1) // 2
    bands = (ntiles + 3) // 4 > 7 and H % 2 == 0 and (H // 2) * W <= 32 and H >= 4     # 8x8: two row bands per image
    Hin = H // 2 + 1 if bands else H
    npa = (Hin * ((W + 1) // 2) + 3) // 4
    lds = 2 * (((Hin + 2) * (W + 1) + 1) * 512 + 41984)
    return (k == 3 and stride == 1 and pad == 1 and T == 16 and Cout % 64 == 0 and Cin % 64 == 0
            and (bands or (ntiles + 3) // 4 <= 7) and npa <= 7 and lds <= 160 * 1024)


def den_pack_weight_fp6(w, bias):
    """[Cout,Cin,3,3] fp32 -> (e2m3 digit planes u8, fp64 scale [Cout], fp64 bias [Cout])."""
    wd = w.detach()
    # a channels-last weight (the training path's parameter format) is packed as stored: no layout copy per layer and iteration
    cl = (wd.is_cuda and wd.dtype == torch.float32 and wd.dim() == 4 and tuple(wd.shape[2:]) == (3, 3) and not wd.is_contiguous()
          and wd.is_contiguous(memory_format=torch.channels_last))
    w = wd if cl else _dev(wd, "weight", torch.float32)
    Cout, Cin = w.shape[0], w.shape[1]
    nbytes = lib.spk_den_packed_weight_fp6_bytes(Cout, Cin)
    if nbytes < 0:
        raise NotImplementedError("spkdiff: fp6 MFMA conv needs Cout % 16 == 0 and Cin % 64 == 0")
    wq = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    scale = torch.empty(Cout, dtype=torch.float64, device=w.device)
    bias_d = torch.empty(Cout, dtype=torch.float64, device=w.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    check((lib.spk_den_pack_weight_fp6_cl if cl else lib.spk_den_pack_weight_fp6)(
        _p(w), _p(b), _p(wq), _p(scale), _p(bias_d), Cout, Cin, _stream(w)), "spk_den_pack_weight_fp6")
    return wq, scale, bias_d


def den_conv3x3_mfma_fp6(in0, packed, Cout, *, bn_a, bn_b, v=None, want_counts=False):
    """in0: C4 spikes [B, C/64, H, W, 16, 32] (int8-tagged). Returns C4 spikes [B, Cout/64, H, W, 16, 32]
    (or (spikes, counts u8 [B, Cout/32, H, W, 32]) with want_counts)."""
    in0 = _dev(in0, "in0", C4_DTYPE)
    B, nch, H, W, T, rec = in0.shape
    if rec != 32:
        raise ValueError("C4 spike records are 32 bytes (64 channels)")
    wq, scale, bias_d = packed
    out = torch.empty((B, Cout // 64, H, W, T, 32), dtype=C4_DTYPE, device=in0.device)
    cnt = torch.empty((B, Cout // 32, H, W, 32), dtype=torch.uint8, device=in0.device) if want_counts else None
    check(lib.spk_den_conv3x3_mfma_fp6(_p(in0), nch, _p(wq), _p(scale), _p(bias_d), _p(bn_a), _p(bn_b), _p(v), _p(out),
                                       _p(cnt), T, B, H, W, Cout, _n_dyn(), _stream(in0)), "spk_den_conv3x3_mfma_fp6")
    return (out, cnt) if want_counts else out


def spikes_cl_to_c4(s):
    """fp32 spikes [T,B,C,H,W] with channels-last memory -> C4 [B, C/64, H, W, T, 32]."""
    s = _cl5(s, "spikes")
    T, B, C, H, W = s.shape
    o = torch.empty((B, C // 64, H, W, T, 32), dtype=C4_DTYPE, device=s.device)
    check(lib.spk_spikes_nhwc_to_fp4(_p(s), _p(o), T, B, C, H * W, _stream(s)), "spk_spikes_nhwc_to_fp4")
    return o


def spikes_cl_to_c4_counts(s):
    """spikes_cl_to_c4 + the spike counts over T, fp32 [B,C,H,W] (channels-last memory), from the same pass."""
    s = _cl5(s, "spikes")
    T, B, C, H, W = s.shape
    o = torch.empty((B, C // 64, H, W, T, 32), dtype=C4_DTYPE, device=s.device)
    cnt = _empty_cl((B, C, H, W), s.device)
    check(lib.spk_spikes_nhwc_to_fp4_counts(_p(s), _p(o), _p(cnt), T, B, C, H * W, _stream(s)), "spk_spikes_nhwc_to_fp4_counts")
    return o, cnt


def den_conv3x3_fp6_raw(in0, packed, Cout):
    """in0: C4 spikes [B, C/64, H, W, 16, 32] -> exact pre-activations fp32 [T,B,Cout,H,W], channels-last memory."""
    in0 = _dev(in0, "in0", C4_DTYPE)
    B, nch, H, W, T, rec = in0.shape
    wq, scale, bias_d = packed
    y = _empty_cl((T, B, Cout, H, W), in0.device)
    check(lib.spk_den_conv3x3_fp6_raw(_p(in0), nch, _p(wq), _p(scale), _p(bias_d), _p(y), T, B, H, W, Cout, _stream(in0)),
          "spk_den_conv3x3_fp6_raw")
    return y


def conv3x3_wgrad_supported(Cout, Cin, H, W):
    return (H, W) in ((7, 7), (8, 8)) and Cout % 128 == 0 and Cin % 64 == 0


# False: the weight gradient of the spike-input convolutions comes from the framework's operator (as in rounds 1-2)
NATIVE_WGRAD = True


def conv3x3_wgrad(gy_cl, spikes_cl, Cout, Cin, want_bias=False):
    """gw [Cout,Cin,3,3] (channels-last memory) of a 3x3 / s1 / p1 convolution from gy [N,Cout,H,W] and BINARY spikes
    [N,Cin,H,W] (7x7 or 8x8 maps), both channels-last fp32 (spk_conv3x3_wgrad_bf16: bf16 matrix cores, exact three-term split of
    gy)."""
    N, H, W = int(gy_cl.shape[0]), int(gy_cl.shape[2]), int(gy_cl.shape[3])
    nb = int(lib.spk_conv3x3_wgrad_ws_bytes(N, int(Cout), int(Cin)))
    if nb <= 0:
        raise NotImplementedError("spk_conv3x3_wgrad_bf16: unsupported shape")
    ws = torch.empty(nb // 4, dtype=torch.float32, device=gy_cl.device)
    gw = torch.empty((Cout, 3, 3, Cin), dtype=torch.float32, device=gy_cl.device)
    gb = torch.empty(Cout, dtype=torch.float32, device=gy_cl.device) if want_bias else None
    with timed("train.conv_wrw"):
        check(lib.spk_conv3x3_wgrad_bf16(_p(gy_cl), _p(spikes_cl), _p(ws), nb, _p(gw), _p(gb), N, H, W, int(Cout), int(Cin),
                                         _stream(gy_cl)), "spk_conv3x3_wgrad_bf16")
    return (gw.permute(0, 3, 1, 2), gb) if want_bias else gw.permute(0, 3, 1, 2)


def conv3x3_dgrad_supported(Cout, Cin, H, W, N):
    # (small batches: the framework's operator is as fast)
    return ((H, W) in ((7, 7), (8, 8)) and Cout % 16 == 0 and Cin % 32 == 0 and Cout * Cin >= 8192 and N >= 64
            and N * H * W * Cout < 2 ** 31)              # (32-bit element offsets in the kernel's staging table)


# False: the data gradient of the spike-input convolutions comes from the framework's operator (as in rounds 1-2)
NATIVE_DGRAD = True


# "bf16x3": three bf16 terms per operand, six products (exact splits); "f16x2": two scaled fp16 terms, three products
DGRAD_FORM = "f16x2"


class WeightPrep:
    """What one optimizer step's worth of a spike-input convolution's weight is turned into ONCE per training iteration by
    ``train_weight_prep``: ``fp6`` = (digit planes, scale, bias) of the exact MFMA forward, ``dg`` = (workspace, bytes, N) of the
    two-term fp16 data gradient, or None where that layer's backward does not take it."""
    __slots__ = ("fp6", "dg", "key")

    def __init__(self, fp6, dg, key):
        self.fp6, self.dg, self.key = fp6, dg, key

    def matches(self, weight):
        return self.key == (weight.data_ptr(), weight._version, tuple(weight.shape))


def _ptr_array(ctype, values):
    return (ctype * len(values))(*values)


def train_weight_prep(layers):
    """layers: [(weight [Cout,Cin,3,3] channels-last fp32 parameter, bias or None, N_dgrad, (H, W))], at most eight.  TWO launches
    for all of them -- spk_den_pack_weight_fp6_cl_multi and spk_conv3x3_dgrad_f16x2_pack_multi -- instead of one fp6 pack per
    layer in the forward and a fill + maximum + pack per layer in the backward (20 launches, ~130 us of a 2.0 ms iteration at the
    reference's batch).  N_dgrad: the image count that layer's data gradient will be called with (T * B; B for the collapsed last
    layer; 0: that layer's backward does not take the native two-term data gradient).  Returns one WeightPrep per layer, or None when a layer does not fit (callers then pack per
    layer as before)."""
    import ctypes
    if not layers or len(layers) > 8:
        return None
    dev = layers[0][0].device
    for w, b, n_dg, hw in layers:
        if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
                and w.is_contiguous(memory_format=torch.channels_last) and w.shape[0] % 16 == 0 and w.shape[1] % 64 == 0):
            return None
    n = len(layers)
    wd = [w.detach() for w, _, _, _ in layers]
    couts, cins = [int(w.shape[0]) for w in wd], [int(w.shape[1]) for w in wd]
    wq = [torch.empty(int(lib.spk_den_packed_weight_fp6_bytes(co, ci)), dtype=torch.uint8, device=dev) for co, ci in zip(couts, cins)]
    scale = [torch.empty(co, dtype=torch.float64, device=dev) for co in couts]
    bias_d = [torch.empty(co, dtype=torch.float64, device=dev) for co in couts]
    bs = [None if b is None else _dev(b.detach(), "bias", torch.float32) for _, b, _, _ in layers]
    vp = ctypes.c_void_p
    ia = lambda v: _ptr_array(ctypes.c_int, [int(x) for x in v])
    pa = lambda ts: _ptr_array(vp, [None if t is None else t.data_ptr() for t in ts])
    check(lib.spk_den_pack_weight_fp6_cl_multi(pa(wd), pa(bs), pa(wq), pa(scale), pa(bias_d), ia(couts), ia(cins), n, _stream(wd[0])),
          "spk_den_pack_weight_fp6_cl_multi")
    dg = [None] * n
    sel = [i for i, (w, _, n_dg, hw) in enumerate(layers)
           if NATIVE_DGRAD and DGRAD_FORM == "f16x2" and n_dg > 0 and tuple(hw) in ((7, 7), (8, 8))
           and n_dg * hw[0] * hw[1] * couts[i] < 2 ** 31]
    if sel:
        nbs = [int(lib.spk_conv3x3_dgrad_ws_bytes(couts[i], cins[i])) for i in sel]
        wss = [torch.empty(nb, dtype=torch.uint8, device=dev) for nb in nbs]
        check(lib.spk_conv3x3_dgrad_f16x2_pack_multi(pa([wd[i] for i in sel]), pa(wss), _ptr_array(ctypes.c_longlong, nbs),
                                                     ia([layers[i][2] for i in sel]), ia([couts[i] for i in sel]),
                                                     ia([cins[i] for i in sel]), len(sel), _stream(wd[0])),
              "spk_conv3x3_dgrad_f16x2_pack_multi")
        for i, ws, nb in zip(sel, wss, nbs):
            dg[i] = (ws, nb, int(layers[i][2]))
    return [WeightPrep((wq[i], scale[i], bias_d[i]), dg[i], (layers[i][0].data_ptr(), layers[i][0]._version, tuple(wd[i].shape)))
            for i in range(n)]


def conv3x3_dgrad(gy_cl, weight, Cin, form=None, prep=None):
    """gi [N,Cin,H,W] (channels-last memory) of a 3x3 / s1 / p1 convolution from gy [N,Cout,H,W] (channels-last fp32; 7x7 or
    8x8 maps) and the weight [Cout,Cin,3,3]: spk_conv3x3_dgrad_f16x2 (two scaled fp16 terms per operand, three products) or
    spk_conv3x3_dgrad_bf16 (three exact bf16 terms, six products).  Both measure 2-4e-7 relative L2 against fp64 on the training
    shapes (an fp32 operator: 2-6e-7).  Per ELEMENT they differ: the bf16 form splits every value exactly, whatever its
    magnitude; the fp16 form scales each image's gy / each input channel's weights by one power of two and keeps fp32-like
    relative precision only within ~17 binades of that set's maximum (below: an absolute 2^-40 of the maximum) -- choose
    ``form='bf16x3'`` where a gradient map's magnitudes span more than that inside one image."""
    N, Cout, H, W = int(gy_cl.shape[0]), int(gy_cl.shape[1]), int(gy_cl.shape[2]), int(gy_cl.shape[3])
    nb = int(lib.spk_conv3x3_dgrad_ws_bytes(Cout, int(Cin)))
    if nb <= 0:
        raise NotImplementedError("spk_conv3x3_dgrad_bf16: unsupported shape")
    form = form or DGRAD_FORM
    gi = torch.empty((N, H, W, Cin), dtype=torch.float32, device=gy_cl.device)
    if prep is not None and prep.dg is not None and form == "f16x2" and prep.dg[2] == N and prep.matches(weight):
        # the weight half was packed with the other layers' at the start of the iteration (train_weight_prep)
        with timed("train.conv_bwd_data"):
            check(lib.spk_conv3x3_dgrad_f16x2_prepacked(_p(gy_cl), _p(prep.dg[0]), prep.dg[1], _p(gi), N, H, W, Cout, int(Cin),
                                                        _stream(gy_cl)), "spk_conv3x3_dgrad_f16x2_prepacked")
        return gi.permute(0, 3, 1, 2)
    w_cl = weight.detach().contiguous(memory_format=torch.channels_last)        # storage [Cout][3][3][Cin]
    ws = torch.empty(nb, dtype=torch.uint8, device=gy_cl.device)
    fn, name = ((lib.spk_conv3x3_dgrad_f16x2, "spk_conv3x3_dgrad_f16x2") if form == "f16x2"
                else (lib.spk_conv3x3_dgrad_bf16, "spk_conv3x3_dgrad_bf16"))
    with timed("train.conv_bwd_data"):
        check(fn(_p(gy_cl), _p(w_cl), _p(ws), nb, _p(gi), N, H, W, Cout, int(Cin), _stream(gy_cl)), name)
    return gi.permute(0, 3, 1, 2)


class SpikeConvTrainFunction(torch.autograd.Function):
    """y = conv3x3(spikes, weight) + bias for BINARY input spikes in training: the forward is the exact
    fp6 x fp4 MFMA convolution (weights re-packed into six radix-32 digit planes each call -- they change every optimizer
    step -- spikes packed to C4); the backward of the 7x7 layers is native where the shape fits -- weight (+ bias) gradient
    on the bf16 MFMA with an exact three-term split of gy (spk_conv3x3_wgrad_bf16), data gradient on the fp16 MFMA with both
    operands as two scaled terms (spk_conv3x3_dgrad_f16x2; DGRAD_FORM selects the three-term bf16 form) -- and the
    framework's operator otherwise.  spikes [T,B,Cin,H,W] in {0,1}; returns channels-last [T,B,Cout,H,W]."""

    @staticmethod
    def forward(ctx, spikes, weight, bias, prep=None, c4=None):
        # prep: this layer's WeightPrep of the iteration (ops.train_weight_prep), c4: the spikes as C4 records where the producing
        # BatchNorm+LIF launch wrote them -- either spares a launch here; both are optional and checked
        s = _cl5(spikes, "spikes")
        Cout = int(weight.shape[0])
        if prep is not None and not prep.matches(weight):
            prep = None
        T, B, C, H, W = (int(v) for v in s.shape)
        if c4 is not None and (tuple(c4.shape) != (B, C // 64, H, W, T, 32) or c4.dtype != C4_DTYPE or c4.device != s.device):
            c4 = None
        with timed("train.conv_fwd_fp6"):
            y = den_conv3x3_fp6_raw(c4 if c4 is not None else spikes_cl_to_c4(s),
                                    prep.fp6 if prep is not None else den_pack_weight_fp6(weight, bias), Cout)
        ctx.save_for_backward(s, weight)
        ctx.has_bias = bias is not None
        ctx.prep = prep
        return y

    @staticmethod
    def backward(ctx, grad_y):
        s, weight = ctx.saved_tensors
        gy = _cl5(grad_y, "grad_y").flatten(0, 1)
        s4 = s.flatten(0, 1)
        Cout, Cin = int(weight.shape[0]), int(weight.shape[1])
        need_gi, need_gw = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])
        need_gb = bool(ctx.has_bias and ctx.needs_input_grad[2])
        gw_native = None
        if (NATIVE_WGRAD and need_gw and tuple(weight.shape[2:]) == (3, 3) and
                conv3x3_wgrad_supported(Cout, Cin, int(s.shape[3]), int(s.shape[4]))):
            # the weight gradient multiplies gy by SPIKES: native on the bf16 matrix cores (the data gradient has no spike operand)
            gw_native = conv3x3_wgrad(gy, s4, Cout, Cin, want_bias=need_gb)      # (+ the bias gradient from the same pass)
            if need_gb:
                gw_native, gb_native = gw_native
            need_gw = False
        gi_native = None
        if (NATIVE_DGRAD and need_gi and tuple(weight.shape[2:]) == (3, 3) and
                conv3x3_dgrad_supported(Cout, Cin, int(s.shape[3]), int(s.shape[4]), int(gy.shape[0]))):
            gi_native = conv3x3_dgrad(gy, weight, Cin, prep=ctx.prep)
            need_gi = False
        gi = gw = gb = None
        if need_gi or need_gw or (need_gb and gw_native is None):
            gi, gw, gb = torch.ops.aten.convolution_backward(
                gy, s4, weight, [Cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                [need_gi, need_gw, need_gb and gw_native is None])
        if gw_native is not None:
            gw = gw_native
            if need_gb:
                gb = gb_native
        if gi_native is not None:
            gi = gi_native
        if gi is not None:
            gi = gi.reshape(s.shape) if gi_native is None else gi.unflatten(0, (s.shape[0], s.shape[1]))
        return gi, gw, gb, None, None


class CatChannelsFunction(torch.autograd.Function):
    """cat((a, b), dim=2) of two [T,B,C,H,W] spike trains (R/snn_model/vq_diffusion.py:205).  Forward: the framework's cat, on the
    4-D views when both operands are channels-last (so that the layout survives).  Backward: two channel SLICES of the incoming
    gradient, as views -- the gradient in front of the last layer is one [B,C,H,W] tensor broadcast over T (stride 0), which the
    reshape in the framework's own backward of flatten + cat + view had to expand and copy (32 MB at B = 32) before slicing."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = int(a.shape[2])
        T = int(a.shape[0])
        if a.permute(0, 1, 3, 4, 2).is_contiguous() and b.permute(0, 1, 3, 4, 2).is_contiguous():
            cat = torch.cat((a.flatten(0, 1), b.flatten(0, 1)), dim=1)
            return cat.view((T, a.shape[1]) + tuple(cat.shape[1:]))
        return torch.cat((a, b), dim=2)

    @staticmethod
    def backward(ctx, g):
        return g[:, :, :ctx.ca], g[:, :, ctx.ca:]


class SpikeConvMeanTrainFunction(torch.autograd.Function):
    """mean over T of conv3x3(spikes_t, weight) + bias -- the denoiser's last layer and its time mean
    (R/snn_model/vq_diffusion.py:185-187, 205-206) as ONE differentiable operator.  Forward: the exact fp6 x fp4 MFMA convolution
    of every step and `sum(0) / T`, the operations of the two-operator form.  Backward: every step receives the SAME output
    gradient g / T, so the weight gradient is sum_b (g_b / T) (x) COUNTS_b (counts = sum_t s_t, 0..T: exact in bf16) and the
    data gradient is one transposed convolution of g / T repeated over T -- 1/T of the matrix work of the per-step backward and
    no [T,B,Cout,H,W] gradient tensor.  spikes [T,B,Cin,H,W] in {0,1}; returns [B,Cout,H,W]."""

    @staticmethod
    def forward(ctx, spikes, weight, bias, prep=None):
        s = _cl5(spikes, "spikes")
        T, Cout = int(s.shape[0]), int(weight.shape[0])
        if prep is not None and not prep.matches(weight):
            prep = None
        with timed("train.conv_fwd_fp6"):
            c4, counts = spikes_cl_to_c4_counts(s)             # counts [B,Cin,H,W], channels-last like s: same pass as the packing
            y = den_conv3x3_fp6_raw(c4, prep.fp6 if prep is not None else den_pack_weight_fp6(weight, bias), Cout)
        ctx.save_for_backward(counts, weight)
        ctx.has_bias, ctx.T = bias is not None, T
        ctx.prep = prep
        return torch.sum(y, dim=0) / T

    @staticmethod
    def backward(ctx, grad_out):
        counts, weight = ctx.saved_tensors
        T, Cout, Cin = ctx.T, int(weight.shape[0]), int(weight.shape[1])
        B, H, W = int(counts.shape[0]), int(counts.shape[2]), int(counts.shape[3])
        g = (grad_out / T).contiguous(memory_format=torch.channels_last)     # the gradient of every step's output
        c = counts.contiguous(memory_format=torch.channels_last)
        need_gi, need_gw = bool(ctx.needs_input_grad[0]), bool(ctx.needs_input_grad[1])
        need_gb = bool(ctx.has_bias and ctx.needs_input_grad[2])
        gi = gw = gb = None
        native = tuple(weight.shape[2:]) == (3, 3) and (H, W) in ((7, 7), (8, 8))
        if need_gw and native and NATIVE_WGRAD and Cout % 128 == 0 and Cin % 64 == 0:
            gw = conv3x3_wgrad(g, c, Cout, Cin)
            need_gw = False
        if need_gi and native and NATIVE_DGRAD and Cout % 16 == 0 and Cin % 32 == 0:
            gi = conv3x3_dgrad(g, weight, Cin, prep=ctx.prep)
            need_gi = False
        if need_gi or need_gw:
            gi2, gw2, _ = torch.ops.aten.convolution_backward(g, c, weight, [Cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                              [need_gi, need_gw, False])
            gi = gi2 if need_gi else gi
            gw = gw2 if need_gw else gw
        if need_gb:
            gb = grad_out.sum(dim=(0, 2, 3))                   # sum over T of g / T
        if gi is not None:
            gi = gi.unsqueeze(0).expand((T,) + tuple(gi.shape))
        return gi, gw, gb, None


def spikes_to_c4(s):
    """fp32 [T,B,C,H,W] -> C4 [B, C/64, H, W, T, 32]."""
    s = _dev(s, "spikes", torch.float32)
    T, B, C, H, W = s.shape
    o = torch.empty((B, C // 64, H, W, T, 32), dtype=C4_DTYPE, device=s.device)
    check(lib.spk_spikes_to_fp4(_p(s), _p(o), T, B, C, H * W, _stream(s)), "spk_spikes_to_fp4")
    return o


def c4_to_spikes(q):
    """C4 [B, C/64, H, W, T, 32] -> fp32 [T,B,C,H,W]."""
    q = _dev(q, "c4", C4_DTYPE)
    B, nch, H, W, T, _ = q.shape
    o = torch.empty((T, B, nch * 64, H, W), dtype=torch.float32, device=q.device)
    check(lib.spk_fp4_to_spikes(_p(q), _p(o), T, B, nch * 64, H * W, _stream(q)), "spk_fp4_to_spikes")
    return o


# ------------------------------------------------------------------------------- fp6v2: the sampler's denoiser convolutions
def den_fp6v2_supported(Cout, Cin, k, stride, pad, T, H, W):
    return (k == 3 and stride == 1 and pad == 1 and T == 16 and (H, W) in ((7, 7), (8, 8)) and Cout % 32 == 0 and
            Cin % 32 == 0)


def den_pack_weight_fp6v2(w, bias):
    """[Cout,Cin,3,3] fp32 -> (digit tiles u8, fp64 scale [Cout], fp64 bias [Cout], fp32 L1 norms [Cout], the quantised
    weights as int32 [Cout, 9, Cin]: the exact recomputation of flagged neurons reads them)."""
    w = _dev(w.detach(), "weight", torch.float32)
    Cout, Cin = w.shape[0], w.shape[1]
    nbytes = lib.spk_den_packed_weight_fp6v2_bytes(Cout, Cin)
    if nbytes < 0:
        raise NotImplementedError("spkdiff: fp6v2 MFMA conv needs Cout % 32 == 0 and Cin % 32 == 0")
    wq = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    scale = torch.empty(Cout, dtype=torch.float64, device=w.device)
    bias_d = torch.empty(Cout, dtype=torch.float64, device=w.device)
    wl1 = torch.empty(Cout, dtype=torch.float32, device=w.device)
    qtab = torch.empty((Cout, 9, Cin), dtype=torch.int32, device=w.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    check(lib.spk_den_pack_weight_fp6v2(_p(w), _p(b), _p(wq), _p(scale), _p(bias_d), _p(wl1), _p(qtab), Cout, Cin, _stream(w)),
          "spk_den_pack_weight_fp6v2")
    return wq, scale, bias_d, wl1, qtab


# Flagged-neuron workspaces of the certified kernels (fp6v2 / vae_fp6: live counter, id list, overflow bitmap, hand-over
# ticket; zero-initialised, every call leaves them clean).  A workspace must never be shared by launches that can run
# concurrently: outside a ``flag_scope`` it is keyed by (kind, device, STREAM, words); inside one it belongs to the scope's
# owner -- a captured graph keeps its scope dict in its cache entry, so the buffers its launches address live as long as the
# graph and no eager call ever touches them.  Nothing here is ever dropped while a graph may address it.
_FLAG_DEFAULT = {}
# (thread-local: the sampler captures with capture_error_mode='thread_local' precisely so that OTHER threads may use the device
#  meanwhile -- their certified-kernel calls must keep their own per-stream workspaces, never the capturing thread's scope)
import threading as _threading
_FLAG_TLS = _threading.local()


class flag_scope:
    """``with ops.flag_scope(store):`` -- certified kernels launched inside take their flag workspaces from ``store`` (a dict
    owned by the caller: one per captured graph / call site)."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        self.prev = getattr(_FLAG_TLS, "store", None)
        _FLAG_TLS.store = self.store
        return self.store

    def __exit__(self, *exc):
        _FLAG_TLS.store = self.prev
        return False


def _flag_ws(kind, device, words):
    store = getattr(_FLAG_TLS, "store", None)
    if store is None:
        key = (kind, str(device), int(torch.cuda.current_stream(device).cuda_stream), int(words))
        buf = _FLAG_DEFAULT.get(key)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("spkdiff: first use of a certified-kernel shape during hipGraph capture outside an "
                                   "ops.flag_scope (its workspace would live in the graph's pool and be shared with eager calls)")
            buf = _FLAG_DEFAULT[key] = torch.zeros(int(words), dtype=torch.int32, device=device)
        return buf
    key = (kind, str(device), int(words))
    buf = store.get(key)
    if buf is None:
        # (inside a capture the memset is captured with it: replays re-zero a buffer the kernels already left clean)
        buf = store[key] = torch.zeros(int(words), dtype=torch.int32, device=device)
    return buf


def _flag_bitmap(device, words):
    return _flag_ws("den", device, words)


# How many flagged neurons the id list of a certified-kernel call takes before the rest go to the overflow bitmap (the `flag_cap`
# argument of spk_den_conv3x3_mfma_fp6v2* / spk_vae_fp6_fwd).  -1 = the whole list (2^20 entries): every product call.  The parity
# suite sets 64 and 0 so that whole samplers / decoders run on the overflow path (tests: test_flag_overflow_*).
FLAG_CAP = -1
# `form` argument of spk_den_conv3x3_mfma_fp6v2: 0 = automatic (small batches: two half-image items per image), 1 = whole-image items
# always (tests: the split form against it)
FP6V2_FORM = 0


def den_conv3x3_mfma_fp6v2(in0, packed, Cout, *, bn_a, bn_b, want_counts=False, need_radius=None):
    """in0: S32 spikes [B, C/32, 7, 7, 16, 16] (int8-tagged). Returns S32 spikes [B, Cout/32, 7, 7, 16, 16]
    (or (spikes, counts u8 [B, Cout/32, 7, 7, 32]) with want_counts).  Fresh LIF state, none written back.
    need_radius: inside ``active_set(..., need=NeedLists)`` the layer computes only the positions listed for that radius
    (1 = the layer the logits convolution reads, 2 = the one below, ...); every other position of the result is
    unspecified."""
    in0 = _dev(in0, "in0", C4_DTYPE)
    B, nch, H, W, T, rec = in0.shape
    if rec != 16:
        raise ValueError("S32 spike records are 16 bytes (32 channels)")
    wq, scale, bias_d, wl1, qtab = packed
    out = torch.empty((B, Cout // 32, H, W, T, 16), dtype=C4_DTYPE, device=in0.device)
    cnt = torch.empty((B, Cout // 32, H, W, 32), dtype=torch.uint8, device=in0.device) if want_counts else None
    flags = _flag_bitmap(in0.device, lib.spk_den_fp6v2_flag_words(B, Cout, H, W))
    if (need_radius is not None and NEED is not None and ACTIVE is not None and (H, W) == (7, 7) and
            need_radius <= NEED.radii and NEED.batch == B):
        rc = lib.spk_den_conv3x3_mfma_fp6v2_listed(_p(in0), nch, _p(wq), _p(scale), _p(bias_d), _p(wl1), _p(qtab), _p(bn_a),
                                                   _p(bn_b), _p(out), _p(cnt), _p(flags), T, B, H, W, Cout, _n_dyn(),
                                                   _p(NEED.buf), NEED.radii, int(need_radius), int(FLAG_CAP), _stream(in0))
        if rc != -2:
            check(rc, "spk_den_conv3x3_mfma_fp6v2_listed")
            return (out, cnt) if want_counts else out
        # SPK_ERR_UNSUPPORTED: this device / partition has too few CUs for the per-class division of the listed launch (e.g.
        # a 32-CU partition with Cout = 512).  The unlisted launch computes a superset of the listed positions: same tokens.
    check(lib.spk_den_conv3x3_mfma_fp6v2(_p(in0), nch, _p(wq), _p(scale), _p(bias_d), _p(wl1), _p(qtab), _p(bn_a), _p(bn_b),
                                         _p(out), _p(cnt), _p(flags), T, B, H, W, Cout, _n_dyn(), int(FLAG_CAP), int(FP6V2_FORM),
                                         _stream(in0)),
          "spk_den_conv3x3_mfma_fp6v2")
    if FP6V2_STATS is not None:
        _fp6v2_stats(in0, packed, Cout, bn_a, bn_b, out, cnt, flags)
    return (out, cnt) if want_counts else out


# bench.py's instrumented pass sets this to a list: every fp6v2 layer call then appends (Cout, Cin, flagged neurons,
# neurons, repair ms, last-position ms) -- the tail parts re-run and timed on their own (spk_den_conv3x3_mfma_fp6v2_part)
FP6V2_STATS = None


def _fp6v2_stats(in0, packed, Cout, bn_a, bn_b, out, cnt, flags):
    B, nch, H, W, T, _ = in0.shape
    wq, scale, bias_d, wl1, qtab = packed
    ms = {}
    for part in (2, 4):
        if part == 4 and (H * W) % 2 == 0:
            ms[part] = 0.0
            continue
        evs = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.spk_den_conv3x3_mfma_fp6v2_part(_p(in0), nch, _p(wq), _p(scale), _p(bias_d), _p(wl1), _p(qtab), _p(bn_a),
                                                      _p(bn_b), _p(out), _p(cnt), _p(flags), T, B, H, W, Cout, _n_dyn(), part,
                                                      int(FLAG_CAP), _stream(in0)), "spk_den_conv3x3_mfma_fp6v2_part")
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms[part] = sorted(a.elapsed_time(b) for a, b in evs)[1]
    FP6V2_STATS.append({"Cout": int(Cout), "Cin": int(nch * 32), "flagged": int(flags[1].item()),
                        "neurons": int(B * Cout * H * W), "repair_ms": ms[2], "last_position_ms": ms[4]})


def spikes_to_s32(s):
    """fp32 [T,B,C,H,W] -> S32 [B, C/32, H, W, T, 16]."""
    s = _dev(s, "spikes", torch.float32)
    T, B, C, H, W = s.shape
    o = torch.empty((B, C // 32, H, W, T, 16), dtype=C4_DTYPE, device=s.device)
    check(lib.spk_spikes_to_s32(_p(s), _p(o), T, B, C, H * W, _stream(s)), "spk_spikes_to_s32")
    return o


def s32_to_spikes(q):
    """S32 [B, C/32, H, W, T, 16] -> fp32 [T,B,C,H,W]."""
    q = _dev(q, "s32", C4_DTYPE)
    B, nch, H, W, T, _ = q.shape
    o = torch.empty((T, B, nch * 32, H, W), dtype=torch.float32, device=q.device)
    check(lib.spk_s32_to_spikes(_p(q), _p(o), T, B, nch * 32, H * W, _stream(q)), "spk_s32_to_spikes")
    return o


# ---------------------------------------------------------------------------------------------- MFMA VQ-VAE layers
def conv_mfma_supported(Cin, Cout, T, mode):
    return T == 16 and Cin % 16 == 0 and (mode == MODE_MEMOUT or Cout % 16 == 0)


def pack_conv_weight_i8(w, bias, transposed):
    """Conv2d [Cout,Cin,k,k] / ConvTranspose2d [Cin,Cout,k,k] fp32 -> (int8 digit planes, scale, bias) (padded to 16)."""
    w = _dev(w.detach(), "weight", torch.float32)
    if transposed:
        Cin, Cout, k, k2 = w.shape
    else:
        Cout, Cin, k, k2 = w.shape
    if k != k2:
        raise NotImplementedError("square kernels only")
    nbytes = lib.spk_conv_packed_weight_i8_bytes(Cout, Cin, k)
    cpad = (Cout + 15) // 16 * 16
    wq = torch.empty(nbytes, dtype=torch.int8, device=w.device)
    scale = torch.empty(cpad, dtype=torch.float64, device=w.device)
    bias_d = torch.empty(cpad, dtype=torch.float64, device=w.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    check(lib.spk_pack_conv_weight_i8(_p(w), _p(b), _p(wq), _p(scale), _p(bias_d), Cout, Cin, k, int(transposed),
                                      _stream(w)), "spk_pack_conv_weight_i8")
    return wq, scale, bias_d


def conv_mfma_fused(in_ptc, packed, Cout, *, mode, k, stride, pad, transposed=False, out_pad=0, bn_a=None, bn_b=None,
                    v=None, coef=None, apply_tanh=False, want_u8=False, collapse_coef=None, out_s32=False):
    """in_ptc u8 [B,H,W,16,Cin] -> LIF: spikes u8 [B,Ho,Wo,16,Cout]; MEMOUT: dict(f32=[B,Cout,Ho,Wo], u8=...).
    LIF with collapse_coef [16]: returns sum_t collapse_coef[t] * spikes[t] as fp32 [B,Ho,Wo,Cout] instead of the spikes
    (input of readout_collapsed)."""
    in_ptc = _dev(in_ptc, "in_ptc", torch.uint8)
    B, H, W, T, Cin = in_ptc.shape
    Ho, Wo = conv_out_size(H, k, stride, pad, transposed, out_pad), conv_out_size(W, k, stride, pad, transposed, out_pad)
    wq, scale, bias_d = packed
    out_p = out_f = out_u = None
    if mode == MODE_LIF and out_s32:
        # nibble-packed "S32" spikes [B, Cout/32, Ho, Wo, 16, 16]: the input layout of the fp6 kernels
        o = torch.empty((B, Cout // 32, Ho, Wo, T, 16), dtype=C4_DTYPE, device=in_ptc.device)
        check(lib.spk_conv_mfma_fused_lif_s32(_p(in_ptc), _p(wq), _p(scale), _p(bias_d), _p(bn_a), _p(bn_b), _p(v), _p(o), T, B,
                                              H, W, Cin, Cout, k, stride, pad, int(transposed), out_pad, _stream(in_ptc)),
              "spk_conv_mfma_fused_lif_s32")
        return o
    if mode == MODE_LIF and collapse_coef is not None:
        coef = _dev(collapse_coef, "collapse_coef", torch.float32)
        if coef.numel() != T:
            raise ValueError("collapse_coef must hold one coefficient per time step")
        out_f = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=in_ptc.device)
    elif mode == MODE_LIF:
        coef = None
        out_p = torch.empty((B, Ho, Wo, T, Cout), dtype=torch.uint8, device=in_ptc.device)
    else:
        out_f = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=in_ptc.device)
        if want_u8:
            out_u = torch.empty((B, Cout, Ho, Wo), dtype=torch.uint8, device=in_ptc.device)
    check(lib.spk_conv_mfma_fused_fwd(_p(in_ptc), _p(wq), _p(scale), _p(bias_d), _p(bn_a), _p(bn_b), _p(v), _p(out_p),
                                      _p(coef), _p(out_f), _p(out_u), int(apply_tanh), mode, T, B, H, W, Cin, Cout, k,
                                      stride, pad, int(transposed), out_pad, _stream(in_ptc)), "spk_conv_mfma_fused_fwd")
    if mode == MODE_LIF:
        return out_f if collapse_coef is not None else out_p
    return {"f32": out_f, "u8": out_u}


VAE_OUT_COLLAPSED, VAE_OUT_S32, VAE_OUT_PTC = 0, 1, 2


def vae_fp6_kind(Cin, Cout, k, stride, pad, out_pad, transposed, T, H, W):
    """Which output form of the fp6 VQ-VAE kernel (csrc/vae_fp6.hip) exists for this 3x3 stride-2 layer on an H x W input:
    VAE_OUT_COLLAPSED (decoder convT2), VAE_OUT_S32 (decoder convT1), VAE_OUT_PTC (encoder conv2), or None."""
    if k != 3 or stride != 2 or pad != 1 or T != 16 or Cout % 32:
        return None
    if transposed and out_pad == 1 and Cin == 64 and (H, W) in ((14, 14), (16, 16)):
        return VAE_OUT_COLLAPSED
    if transposed and out_pad == 1 and Cin == 16 and (H, W) in ((7, 7), (8, 8)):
        return VAE_OUT_S32
    if not transposed and Cin == 32 and (H, W) in ((14, 14), (16, 16)):
        return VAE_OUT_PTC
    return None


def convT_fp6_supported(Cin, Cout, k, stride, pad, out_pad, transposed, T, H, W):
    return vae_fp6_kind(Cin, Cout, k, stride, pad, out_pad, transposed, T, H, W) == VAE_OUT_COLLAPSED


def vae_fp6_pack(w, bias, transposed):
    """Conv2d [Cout,Cin,3,3] / ConvTranspose2d [Cin,Cout,3,3] fp32 (+bias) -> (digit tiles, scale f64, bias f64, qtab int32)."""
    w = _dev(w.detach(), "weight", torch.float32)
    Cin, Cout = (int(w.shape[0]), int(w.shape[1])) if transposed else (int(w.shape[1]), int(w.shape[0]))
    n = lib.spk_vae_fp6_packed_bytes(Cout, Cin)
    if n <= 0:
        raise NotImplementedError("spk_vae_fp6_pack: unsupported shape")
    wq = torch.empty(n, dtype=torch.uint8, device=w.device)
    scale = torch.empty(Cout, dtype=torch.float64, device=w.device)
    bias_d = torch.empty(Cout, dtype=torch.float64, device=w.device)
    qtab = torch.empty((Cout, 9, Cin), dtype=torch.int32, device=w.device)
    b = None if bias is None else _dev(bias.detach(), "bias", torch.float32)
    check(lib.spk_vae_fp6_pack(_p(w), _p(b), _p(wq), _p(scale), _p(bias_d), _p(qtab), Cout, Cin, int(transposed), _stream(w)),
          "spk_vae_fp6_pack")
    return wq, scale, bias_d, qtab, Cin


def convT_fp6_pack(w, bias):
    return vae_fp6_pack(w, bias, True)


def ptc_to_s32(ptc):
    """u8 PTC spikes [B,H,W,16,C] -> S32 [B, ceil(C/32), H, W, 16, 16] (zero nibbles beyond C)."""
    ptc = _dev(ptc, "ptc", torch.uint8)
    B, H, W, T, C = ptc.shape
    o = torch.empty((B, (C + 31) // 32, H, W, T, 16), dtype=C4_DTYPE, device=ptc.device)
    check(lib.spk_ptc_to_s32(_p(ptc), _p(o), T, B, H * W, C, _stream(ptc)), "spk_ptc_to_s32")
    return o


def vae_fp6_fwd(in_s32, packed, Cout, *, bn_a, bn_b, transposed, out_kind, coef=None):
    """One spike-input 3x3 stride-2 VQ-VAE layer + BN + LIF from the reset state on the fp6 MFMA (see vae_fp6_kind).
    in_s32: S32 spikes [B, nch, H, W, 16, 16].  Returns fp32 [B,Ho,Wo,Cout] (collapsed), S32 [B,Cout/32,Ho,Wo,16,16] or u8 PTC
    [B,Ho,Wo,16,Cout]."""
    in_s32 = _dev(in_s32, "in_s32", C4_DTYPE)
    B, nch, H, W, T, rec = in_s32.shape
    wq, scale, bias_d, qtab, Cin = packed
    if rec != 16 or nch != (Cin + 31) // 32:
        raise ValueError("S32 spikes with ceil(Cin / 32) chunks expected")
    Ho, Wo = (2 * H, 2 * W) if transposed else (H // 2, W // 2)
    if out_kind == VAE_OUT_COLLAPSED:
        coef = _dev(coef, "coef", torch.float32)
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32, device=in_s32.device)
    elif out_kind == VAE_OUT_S32:
        out = torch.empty((B, Cout // 32, Ho, Wo, T, 16), dtype=C4_DTYPE, device=in_s32.device)
    else:
        out = torch.empty((B, Ho, Wo, T, Cout), dtype=torch.uint8, device=in_s32.device)
    flags = _flag_ws("vae", in_s32.device, lib.spk_vae_fp6_flag_words(B, Cout, Ho, Wo))
    check(lib.spk_vae_fp6_fwd(_p(in_s32), _p(wq), _p(scale), _p(bias_d), _p(qtab), _p(bn_a), _p(bn_b), _p(coef), _p(out),
                              int(out_kind), _p(flags), T, B, H, W, Cin, Cout, int(transposed), int(FLAG_CAP), _stream(in_s32)),
          "spk_vae_fp6_fwd")
    return out


def convT_fp6_collapsed(in_s32, packed, Cout, *, bn_a, bn_b, coef):
    """Decoder convT2: S32 spikes [B, 2, H, W, 16, 16] -> fp32 [B, 2H, 2W, Cout] = sum_t coef[t] * spikes[t]."""
    return vae_fp6_fwd(in_s32, packed, Cout, bn_a=bn_a, bn_b=bn_b, transposed=True, out_kind=VAE_OUT_COLLAPSED, coef=coef)


_COEF_SUMS = {}


def readout_collapsed_supported(Cin, Cout, k):
    return Cin % 8 == 0 and k % 2 == 1 and ((4 + k - 1) * 64 * (Cin + 4) + Cout * k * k * Cin) * 4 <= 64 * 1024


def readout_collapsed(x_bhwc, weight, bias, coef, *, apply_tanh=False, want_u8=False, k=3, pad=1, transposed=True):
    """Linear read-out layer on time-collapsed spikes: x_bhwc fp32 [B,H,W,Cin] = sum_t coef[t] * spikes[t] (conv_mfma_fused
    with collapse_coef) -> dict(f32=[B,Cout,H,W] = conv(x) + bias * sum(coef) (tanh), u8=...).  Stride 1, 'same' padding."""
    x = _dev(x_bhwc, "x", torch.float32)
    B, H, W, Cin = x.shape
    w = _dev(weight, "weight", torch.float32)
    Cout = w.shape[1] if transposed else w.shape[0]
    if W > 64 or not readout_collapsed_supported(Cin, Cout, k):
        raise NotImplementedError("spk_readout_collapsed_fwd: unsupported geometry")
    if bias is not None:
        bias = _dev(bias, "bias", torch.float32)
    out_f = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    out_u = torch.empty((B, Cout, H, W), dtype=torch.uint8, device=x.device) if want_u8 else None
    ckey = (coef.data_ptr(), coef._version, coef.numel())
    csum = _COEF_SUMS.get(ckey)
    if csum is None:                    # (a device -> host copy: once per coefficient tensor, not per call)
        csum = 0.0
        for c in coef.detach().reshape(-1).float().cpu().tolist():      # fp32 left-to-right, as torch.sum over T would add
            csum = float(torch.tensor(csum, dtype=torch.float32) + torch.tensor(c, dtype=torch.float32))
        if len(_COEF_SUMS) > 16:
            _COEF_SUMS.clear()
        _COEF_SUMS[ckey] = csum
    check(lib.spk_readout_collapsed_fwd(_p(x), _p(w), _p(bias), float(csum), _p(out_f), _p(out_u), int(apply_tanh), B, H, W,
                                        Cin, Cout, int(k), int(pad), int(transposed), _stream(x)), "spk_readout_collapsed_fwd")
    return {"f32": out_f, "u8": out_u}


# ---------------------------------------------------------------------------------------------- VQ
def vq_readout_argmin(z_ptc, coef, alpha, codebook, want_zq=True, want_xm=False):
    z_ptc = _dev(z_ptc, "z_ptc", torch.uint8)
    B, H, W, T, D = z_ptc.shape
    codebook = _dev(codebook.detach(), "codebook", torch.float32)
    K = codebook.shape[0]
    coef = _dev(coef, "coef", torch.float32)
    if coef.numel() != T:
        raise RuntimeError(f"The size of tensor a ({T}) must match the size of tensor b ({coef.numel()}) at "
                           "non-singleton dimension 0")
    alpha = _dev(alpha.detach().reshape(1), "alpha", torch.float32)
    idx = torch.empty(B * H * W, dtype=torch.int64, device=z_ptc.device)
    zq = torch.empty((B, D, H, W), dtype=torch.float32, device=z_ptc.device) if want_zq else None
    xm = torch.empty((B * H * W, D), dtype=torch.float32, device=z_ptc.device) if want_xm else None
    check(lib.spk_vq_readout_argmin(_p(z_ptc), _p(coef), _p(alpha), _p(codebook), _p(idx), _p(zq), _p(xm), T, B, D,
                                    H * W, K, _stream(z_ptc)), "spk_vq_readout_argmin")
    return idx, zq, xm


def vq_argmin(flat_x, codebook):
    flat_x = _dev(flat_x, "flat_x", torch.float32)
    codebook = _dev(codebook.detach(), "codebook", torch.float32)
    N, D = flat_x.shape
    idx = torch.empty(N, dtype=torch.int64, device=flat_x.device)
    check(lib.spk_vq_argmin(_p(flat_x), _p(codebook), _p(idx), N, D, codebook.shape[0], _stream(flat_x)),
          "spk_vq_argmin")
    return idx


_VQ_TRAIN_WS = {}


def _vq_train_ws(device):
    ws = _VQ_TRAIN_WS.get(device)
    if ws is None:
        ws = _VQ_TRAIN_WS[device] = torch.zeros(int(lib.spk_vq_train_ws_bytes()), dtype=torch.uint8, device=device)
    return ws


class VQTrainFunction(torch.autograd.Function):
    """Training branch of VectorQuantizer.forward up to the spike generator (R/snn_model/vae_model.py:61-78): read-out, code
    search, VQ + commitment loss and the straight-through value as three launches (spk_vq_train_readout, spk_vq_argmin,
    spk_vq_train_quant); the backward -- gradients of the encoder spikes, of alpha and of the codebook -- is one
    (spk_vq_train_bwd: the codebook rows are summed by one workgroup per code in a fixed order).
    apply(x_seq [T,B,D,h,w], coef [T], alpha [] , codebook [K,D], beta) -> (quantized [B,D,h,w], loss_1 [])."""

    @staticmethod
    def forward(ctx, x_seq, coef, alpha, codebook, beta):
        x = _dev(x_seq.detach(), "x_seq", torch.float32)
        T, B, D, h, w = x.shape
        HW, N = h * w, B * h * w
        cf = _dev(coef.detach().flatten(), "coef", torch.float32)
        al = _dev(alpha.detach().reshape(1), "alpha", torch.float32)
        E = _dev(codebook.detach(), "codebook", torch.float32)
        xm = torch.empty((N, D), dtype=torch.float32, device=x.device)
        dxa = torch.empty_like(xm)
        check(lib.spk_vq_train_readout(_p(x), _p(cf), _p(al), _p(xm), _p(dxa), T, B, D, HW, _stream(x)), "spk_vq_train_readout")
        idx = vq_argmin(xm, E)
        out = torch.empty((B, D, h, w), dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        check(lib.spk_vq_train_quant(_p(xm), _p(idx), _p(E), _p(out), _p(loss), float(beta), _p(_vq_train_ws(x.device)), N, D, HW,
                                     _stream(x)), "spk_vq_train_quant")
        ctx.save_for_backward(xm, idx, E, dxa, cf, al)
        ctx.cfg = (T, B, D, h, w, float(beta), tuple(alpha.shape))
        ctx.mark_non_differentiable(idx)
        ctx.indices = idx
        return out, loss

    @staticmethod
    def backward(ctx, g_out, g_loss):
        xm, idx, E, dxa, cf, al = ctx.saved_tensors
        T, B, D, h, w, beta, ashape = ctx.cfg
        N, K = B * h * w, E.shape[0]
        go = torch.zeros((B, D, h, w), dtype=torch.float32, device=xm.device) if g_out is None else g_out.contiguous()
        gl = None if g_loss is None else g_loss.reshape(1).contiguous()
        gx = torch.empty((T, B, D, h, w), dtype=torch.float32, device=xm.device)
        ga = torch.empty(1, dtype=torch.float32, device=xm.device)
        gE = torch.empty_like(E)
        with timed("train.vq_bwd"):
            check(lib.spk_vq_train_bwd(_p(go), _p(gl), _p(xm), _p(idx), _p(E), _p(dxa), _p(cf), _p(al), beta, _p(gx), _p(ga), _p(gE),
                                       _p(_vq_train_ws(xm.device)), T, N, D, h * w, K, _stream(xm)), "spk_vq_train_bwd")
        return gx, None, ga.reshape(ashape), gE, None


class PSPLossFunction(torch.autograd.Function):
    """loss_2 of VectorQuantizer.forward (R/snn_model/vae_model.py:79-84): mean((psp(q) - sg(psp(x)))^2) + beta * mean((sg(psp(q)) -
    psp(x))^2) over spike tensors [T, ...] -- one launch forward (spk_psp_loss_fwd: both filters in registers), one backward
    (spk_psp_loss_bwd: both adjoint filters) instead of four filter launches and ~16 element-wise / reduce launches.
    apply(q_seq, x_seq, beta, tau_s) -> loss []."""

    @staticmethod
    def forward(ctx, q_seq, x_seq, beta, tau_s):
        q = _dev(q_seq.detach(), "q_seq", torch.float32)
        x = _dev(x_seq.detach(), "x_seq", torch.float32)
        if q.shape != x.shape:
            raise ValueError(f"PSP loss: shapes {tuple(q.shape)} and {tuple(x.shape)} differ")
        T, N = q.shape[0], q[0].numel()
        if T > 16:
            raise NotImplementedError("spk_psp_loss: at most 16 time steps")
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        check(lib.spk_psp_loss_fwd(_p(q), _p(x), _p(loss), float(beta), float(tau_s), _p(_vq_train_ws(q.device)), T, N, _stream(q)),
              "spk_psp_loss_fwd")
        ctx.save_for_backward(q, x)
        ctx.cfg = (float(beta), float(tau_s))
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        q, x = ctx.saved_tensors
        beta, tau_s = ctx.cfg
        T, N = q.shape[0], q[0].numel()
        gq, gx = torch.empty_like(q), torch.empty_like(x)
        with timed("train.psp_loss_bwd"):
            check(lib.spk_psp_loss_bwd(_p(q), _p(x), _p(g_loss.reshape(1).contiguous()), _p(gq), _p(gx), beta, tau_s, T, N, _stream(q)),
                  "spk_psp_loss_bwd")
        return gq, gx, None, None


class ReconLossFunction(torch.autograd.Function):
    """mse_loss(tanh(memout(y)), image) of SNN_VQVAE.forward in training (R/snn_model/vae_model.py:189-196): read-out, tanh and the
    mean square as one launch, the gradient of the decoder's output as another (spk_recon_loss_fwd / _bwd).
    apply(y_seq [T,B,C,H,W], coef [T], image [B,C,H,W]) -> loss []."""

    @staticmethod
    def forward(ctx, y_seq, coef, image):
        y = _dev(y_seq.detach(), "y_seq", torch.float32)
        img = _dev(image.detach(), "image", torch.float32)
        cf = _dev(coef.detach().flatten(), "coef", torch.float32)
        T, N = y.shape[0], y[0].numel()
        if img.numel() != N or cf.numel() != T:
            raise ValueError(f"recon loss: y {tuple(y.shape)}, image {tuple(img.shape)}, coef {tuple(cf.shape)}")
        xr = torch.empty_like(img)
        loss = torch.empty((), dtype=torch.float32, device=y.device)
        check(lib.spk_recon_loss_fwd(_p(y), _p(cf), _p(img), _p(xr), _p(loss), _p(_vq_train_ws(y.device)), T, N, _stream(y)),
              "spk_recon_loss_fwd")
        ctx.save_for_backward(xr, img, cf)
        ctx.shape = tuple(y.shape)
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        xr, img, cf = ctx.saved_tensors
        gy = torch.empty(ctx.shape, dtype=torch.float32, device=xr.device)
        with timed("train.recon_loss_bwd"):
            check(lib.spk_recon_loss_bwd(_p(xr), _p(img), _p(cf), _p(g_loss.reshape(1).contiguous()), _p(gy), ctx.shape[0], xr.numel(),
                                         _stream(xr)), "spk_recon_loss_bwd")
        return gy, None, None


def embedding(tokens, codebook, nchw_hw=None):
    """nn.Embedding lookup; nchw_hw=(h,w) writes [B,D,h,w] for tokens [B,h,w]."""
    tokens = _dev(tokens, "tokens", torch.int64)
    codebook = _dev(codebook.detach(), "codebook", torch.float32)
    K, D = codebook.shape
    N = tokens.numel()
    if nchw_hw is None:
        out = torch.empty(tuple(tokens.shape) + (D,), dtype=torch.float32, device=tokens.device)
        HW, nchw = 1, 0
    else:
        h, w = nchw_hw
        HW, nchw = h * w, 1
        out = torch.empty((N // HW, D, h, w), dtype=torch.float32, device=tokens.device)
    check(lib.spk_embedding_fwd(_p(tokens), _p(codebook), _p(out), N, D, K, HW, nchw, _stream(tokens)),
          "spk_embedding_fwd")
    return out


def spikegen_tokens_s32(tokens, codebook, w_packed, bias, bn_a, bn_b, T=16, table_key=None, table_slot=None):
    """tokens int64 [B,h,w] -> S32 spikes [B,1,h,w,16,16] of the spike generator (embedding + 1x1 conv + BN + LIF from the reset state on
    the repeated code vector), by a per-token pattern table (spk_spikegen_tokens_s32).  w_packed: [1][D][Cout], Cout 16 or 32.
    table_slot: a dict OWNED BY THE CALLER's module (one per spike generator: no table is shared between models) that keeps the table
    buffer per (device, K, Cout) with the key and the stream of its last eager build.  table_key: anything that changes whenever
    codebook, weights or BN terms do (parameter versions + the owner's invalidation epoch); the slot's table is reused while the key
    AND the stream are equal.  A call under stream capture always builds, into a buffer of its own (the graph's pool), and never touches
    the slot: a replay bakes the derived inputs of capture time and must not leave its table behind an eager call's key."""
    tokens = _dev(tokens, "tokens", torch.int64)
    codebook = _dev(codebook.detach(), "codebook", torch.float32)
    K, D = codebook.shape
    kk, Cin, Cout = w_packed.shape
    if kk != 1 or Cin != D:
        raise ValueError("a 1x1 generator over the code vector expected")
    nbytes = lib.spk_spikegen_table_bytes(K, Cout)
    if nbytes < 0 or T != 16:
        raise NotImplementedError("spikegen_tokens_s32: Cout 16 or 32, T = 16")
    stream = _stream(tokens)
    capturing = torch.cuda.is_current_stream_capturing()
    ent = None
    if table_slot is not None and not capturing:    # (a captured call builds into a buffer of its own, from the graph's pool: replays
        skey = (tokens.device, K, Cout)             #  -- which bake the derived inputs of capture time -- never write the eager slot)
        ent = table_slot.get(skey)
        if ent is None:
            ent = table_slot[skey] = [torch.empty(nbytes // 2, dtype=torch.int16, device=tokens.device), None, None]
    if ent is None:
        ws, build = torch.empty(nbytes // 2, dtype=torch.int16, device=tokens.device), True
    else:
        ws = ent[0]
        build = table_key is None or ent[1] != table_key or ent[2] != stream
        ent[1], ent[2] = (None, None) if table_key is None else (table_key, stream)
    B, h, w = tokens.shape
    out = torch.empty((B, 1, h, w, T, 16), dtype=C4_DTYPE, device=tokens.device)
    check(lib.spk_spikegen_tokens_s32(_p(tokens), _p(codebook), _p(w_packed), _p(bias), _p(bn_a), _p(bn_b), _p(ws), int(build), _p(out),
                                      T, tokens.numel(), K, D, Cout, stream), "spk_spikegen_tokens_s32")
    return out


# ---------------------------------------------------------------------------------------------- sampler
def den_build_input(x, t, out=None):
    """x: float [B,1,h,w] or int64 tokens; t: int64 [B] tensor or python int -> fp32 [B,2,h,w]."""
    xf = xi = None
    if x.dtype == torch.int64:
        xi = _dev(x, "x_t", torch.int64)
    else:
        xf = _dev(x, "x", torch.float32)
    B = x.shape[0]
    HW = x[0].numel()
    h, w = x.shape[-2], x.shape[-1]
    if out is None:
        out = torch.empty((B, 2, h, w), dtype=torch.float32, device=x.device)
    tv, ts = None, 0
    if torch.is_tensor(t):
        tv = _dev(t, "t", torch.int64)
        if tv.numel() != B:
            raise ValueError("t must have one entry per sample")
    else:
        ts = int(t)
    act, nact = (None, None) if ACTIVE is None else ACTIVE
    check(lib.spk_den_build_input(_p(xf), _p(xi), _p(tv), ts, _p(out), B, HW, _p(act), _p(nact), _stream(x)),
          "spk_den_build_input")
    return out


def psample_step(logits, x_t, unmasked, t, temp=1.0, u=None, q=None, seed=0, offset=0, x0_hat=None, philox_state=None,
                 next_input=None):
    """In-place update of x_t (int64) and unmasked (bool/u8) from logits [B,K,h,w].  next_input (dense form only): fp32
    [B,2,h,w] that receives the denoiser input of the next reverse step, cat(x_t, t - 1)."""
    logits = _dev(logits, "logits", torch.float32)
    B, K = logits.shape[0], logits.shape[1]
    HW = logits[0, 0].numel()
    if x_t.dtype != torch.int64 or not x_t.is_cuda or not x_t.is_contiguous():
        raise ValueError("x_t must be a contiguous int64 device tensor (updated in place)")
    if unmasked.dtype not in (torch.bool, torch.uint8) or not unmasked.is_contiguous():
        raise ValueError("unmasked must be a contiguous bool/uint8 device tensor (updated in place)")
    if x_t.numel() != B * HW or unmasked.numel() != B * HW:
        raise ValueError("x_t / unmasked size mismatch")
    if u is not None:
        u = _dev(u, "u", torch.float32)
    if q is not None:
        q = _dev(q, "q", torch.float32)
        if q.numel() != B * HW * K:
            raise ValueError("q must have B*HW*K entries")
    if philox_state is not None and (philox_state.dtype != torch.int64 or philox_state.numel() != 2):
        raise ValueError("philox_state must be an int64 device tensor {seed, base offset}")
    act, nact = (None, None) if ACTIVE is None else ACTIVE
    if next_input is not None:
        next_input = _dev(next_input, "next_input", torch.float32)
        if next_input.numel() != B * 2 * HW or not next_input.is_contiguous():
            raise ValueError("next_input must be a contiguous fp32 [B,2,h,w] tensor")
    check(lib.spk_psample_step(_p(logits), _p(x_t), _p(unmasked), int(t), float(temp), _p(u), _p(q), int(seed),
                               int(offset), _p(philox_state), _p(x0_hat), B, HW, K, _p(act), _p(nact), _p(next_input),
                               _stream(logits)), "spk_psample_step")
    return x_t, unmasked


def den_step_tail(cnt5, cnt1, packed6, x_t, unmasked, t, temp, *, T, K, u=None, q=None, seed=0, offset=0, philox_state=None,
                  conv1=None, want_logits=False):
    """The tail of one dense reverse step as one launch (spk_den_step_tail): conv6 on the spike counts + time mean, the token
    update of ``psample_step`` (x_t / unmasked in place, same noise arguments) and -- with ``conv1 = (w_packed [9,2,64], bias,
    bn_a, bn_b)`` -- the first denoiser layer of the next step.  Returns (x1 S32 [B,2,h,w,16,16], cnt1 u8 [B,2,h,w,32]) or None,
    and the logits fp32 [B,K,h,w] when asked for.  Inside an ``active_set`` scope (the untouched-image elimination) cnt5 / cnt1 / logits
    are per SLOT of the active list and x_t / unmasked / the noise per image; ``conv1`` must be None there (the next step's first layer
    belongs to the next step's active set)."""
    cnt5 = _dev(cnt5, "cnt5", torch.uint8)
    cnt1 = _dev(cnt1, "cnt1", torch.uint8)
    B, nch5, H, W, _ = cnt5.shape
    wq, scale, bias_d = packed6
    if x_t.dtype != torch.int64 or not x_t.is_contiguous() or x_t.numel() != B * H * W:
        raise ValueError("x_t must be a contiguous int64 device tensor [B,1,h,w]")
    if unmasked.dtype not in (torch.bool, torch.uint8) or not unmasked.is_contiguous() or unmasked.numel() != B * H * W:
        raise ValueError("unmasked must be a contiguous bool/uint8 device tensor [B,1,h,w]")
    if u is not None:
        u = _dev(u, "u", torch.float32)
    if q is not None:
        q = _dev(q, "q", torch.float32)
        if q.numel() != B * H * W * K:
            raise ValueError("q must have B*HW*K entries")
    logits = torch.empty((B, K, H, W), dtype=torch.float32, device=cnt5.device) if want_logits else None
    act, nact = (None, None) if ACTIVE is None else ACTIVE
    if act is not None and conv1 is not None:
        raise ValueError("den_step_tail: the fused first layer is the dense form's (inside an active_set scope pass conv1=None)")
    x1 = c1o = w1 = b1 = a1 = bb1 = None
    if conv1 is not None and int(T) != 16:
        raise NotImplementedError("spk_den_step_tail: the fused first layer is the T = 16, LIFNode(tau=2, v_threshold=1, "
                                  "v_reset=0) form; run conv1 as its own launch for other step counts")
    if conv1 is not None:
        w1, b1, a1, bb1 = conv1
        x1 = torch.empty((B, 2, H, W, T, 16), dtype=C4_DTYPE, device=cnt5.device)
        c1o = torch.empty((B, 2, H, W, 32), dtype=torch.uint8, device=cnt5.device)
    check(lib.spk_den_step_tail(_p(cnt5), int(nch5), _p(cnt1), int(cnt1.shape[1]), _p(wq), _p(scale), _p(bias_d), _p(logits),
                                _p(x_t), _p(unmasked), int(t), float(temp), _p(u), _p(q), int(seed), int(offset),
                                _p(philox_state), _p(w1), _p(b1), _p(a1), _p(bb1), _p(x1), _p(c1o), int(T), B, H, W, int(K),
                                _p(act), _p(nact), _stream(cnt5)), "spk_den_step_tail")
    return (None if x1 is None else (x1, c1o)), logits


def q_sample(x_0, t, u, num_timesteps, mask_id):
    """(x_t, x_0_ignore, mask) of R/snn_model/vq_diffusion.py:61-75 from x_0 fp32 [B,1,h,w], t int64 [B] and the uniforms u
    (spk_q_sample: one launch)."""
    x0 = _dev(x_0, "x_0", torch.float32)
    uu = _dev(u, "u", torch.float32)
    tt = t.contiguous()
    if tt.dtype != torch.int64 or not tt.is_cuda:
        raise ValueError("t must be an int64 device tensor")
    B = int(x0.shape[0])
    HW = x0[0].numel()
    x_t, ign = torch.empty_like(x0), torch.empty_like(x0)
    mask = torch.empty(x0.shape, dtype=torch.bool, device=x0.device)
    check(lib.spk_q_sample(_p(x0), _p(tt), _p(uu), _p(x_t), _p(ign), _p(mask), B, HW, int(num_timesteps), float(mask_id),
                           _stream(x0)), "spk_q_sample")
    return x_t, ign, mask


def philox_noise(seed, offset, B, HW, K, device, philox_state=None, want_q=True):
    """The (u [B*HW], q [B*HW, K]) a Philox-mode reverse step with these (seed, offset, philox_state) arguments draws
    (spk_philox_noise): parity aid -- the oracle run on the dumped noise must reproduce the Philox-mode tokens."""
    u = torch.empty(B * HW, dtype=torch.float32, device=device)
    q = torch.empty((B * HW, K), dtype=torch.float32, device=device) if want_q else None
    check(lib.spk_philox_noise(int(seed), int(offset), _p(philox_state), _p(u), _p(q), int(B), int(HW), int(K), _stream(u)),
          "spk_philox_noise")
    return u, q


class TensorChecksum:
    """Content checksum of a fixed set of device tensors in one launch (spk_checksum_multi): ``value()`` synchronises."""

    @staticmethod
    def select(tensors):
        """The tensors a checksum covers: device tensors whose numel * itemsize bytes from data_ptr are exactly their
        elements in SOME order (default or channels-last memory: the sum does not depend on the order)."""
        out = []
        for t in tensors:
            if t is None or not t.is_cuda or t.numel() == 0:
                continue
            dense = t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
            if dense:
                out.append(t)
        return out

    @staticmethod
    def key_of(tensors):
        return tuple((t.data_ptr(), t.numel() * t.element_size()) for t in TensorChecksum.select(tensors))

    def __init__(self, tensors):
        ts = self.select(tensors)
        if not ts:
            raise ValueError("TensorChecksum: no device tensors")
        self.key = tuple((t.data_ptr(), t.numel() * t.element_size()) for t in ts)
        dev = ts[0].device
        tab = []
        for t in ts:
            nb = t.numel() * t.element_size()
            if nb % 4 or t.data_ptr() % 4:
                raise NotImplementedError("TensorChecksum: tensors must be whole 4-byte words")
            tab += [t.data_ptr(), nb // 4]
        self.table = torch.tensor(tab, dtype=torch.int64).to(dev)
        self.out = torch.zeros(1, dtype=torch.int64, device=dev)
        self.n = len(ts)
        self._keep = ts

    def value(self):
        check(lib.spk_checksum_multi(_p(self.table), self.n, _p(self.out), _stream(self.out)), "spk_checksum_multi")
        return int(self.out.item())


class NeedLists:
    """Device buffer of spk_select_needed for ``batch`` image slots and ``radii`` layer depths (zero-initialised once)."""
    __slots__ = ("buf", "batch", "radii")

    def __init__(self, batch, radii, device):
        n = lib.spk_select_needed_bytes(int(batch), int(radii))
        if n <= 0:
            raise ValueError("spk_select_needed_bytes: bad (batch, radii)")
        self.buf = torch.zeros(n, dtype=torch.uint8, device=device)
        self.batch, self.radii = int(batch), int(radii)

    def records(self, radius):
        """uint8 [B, 64] view of the records of one radius (tests)."""
        R, B = self.radii, self.batch
        lists = 64 + R * 64 + R * 6 * B * 4
        off = (lists + 63) // 64 * 64 + (radius - 1) * B * 64
        return self.buf[off:off + B * 64].view(B, 64)


def select_needed(unmasked, t, active, need, u=None, seed=0, offset=0, philox_state=None, K=128):
    """Positions each active image needs from the layers below the logits at reverse step t (spk_select_needed); ``active``
    = the pair returned by select_active for the same step, ``need`` a NeedLists for the same batch.  7x7 latents.  ``K``:
    the class count of the step's psample_step call (the stride of the Philox counter layout; unused with injected u)."""
    B = unmasked.shape[0]
    H, W = int(unmasked.shape[-2]), int(unmasked.shape[-1])
    if need.batch != B:
        raise ValueError("NeedLists was sized for another batch")
    if u is not None:
        u = _dev(u, "u", torch.float32)
    check(lib.spk_select_needed(_p(unmasked), int(t), _p(u), int(seed), int(offset), _p(philox_state), _p(active[0]),
                                _p(active[1]), _p(need.buf), B, H, W, need.radii, int(K), _stream(unmasked)), "spk_select_needed")
    return need


def select_active(unmasked, t, u=None, seed=0, offset=0, philox_state=None, out=None, K=128):
    """Images that reverse step t touches (at least one position with u < 1/t still masked): returns (active int32 [B]
    ascending image list, n_active int32 [2]: count and a work word) on the device; same u / Philox arguments as psample_step
    (``K``: that call's class count, the stride of the Philox counter layout; unused with injected u)."""
    if unmasked.dtype not in (torch.bool, torch.uint8) or not unmasked.is_cuda or not unmasked.is_contiguous():
        raise ValueError("unmasked must be a contiguous bool/uint8 device tensor")
    B = unmasked.shape[0]
    HW = unmasked[0].numel()
    if u is not None:
        u = _dev(u, "u", torch.float32)
    if out is None:
        out = (torch.empty(B, dtype=torch.int32, device=unmasked.device),
               torch.zeros(2, dtype=torch.int32, device=unmasked.device))     # [count, work word (zero between calls)]
    elif out[1].numel() < 2:
        raise ValueError("n_active needs two int32 words: [count, work word (zero before the first call)]")
    check(lib.spk_select_active(_p(unmasked), int(t), _p(u), int(seed), int(offset), _p(philox_state), _p(out[0]),
                                _p(out[1]), B, HW, int(K), _stream(unmasked)), "spk_select_active")
    return out
