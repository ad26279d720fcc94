"""ctypes binding of ``libspkdiff.so`` (the C-ABI declared in ``include/spkdiff.h``).

The library is the ONLY engine of this package: if it is missing (not built) the import fails loudly --
there is no PyTorch/CPU fallback anywhere in the product path.  ``torch`` is imported first so that the
HIP runtime the library binds to is the one PyTorch-ROCm already loaded (same ``libamdhip64.so`` soname),
which is what makes torch's device pointers and streams valid arguments.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_ulonglong, c_void_p

import torch  # noqa: F401  (must precede the dlopen below: shares torch's HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPKDIFF_LIB") or os.path.join(_HERE, "libspkdiff.so")   # SPKDIFF_LIB: A/B builds


class SpkdiffError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"spkdiff: native library not found at {LIB_PATH}. Build it with "
            f"`make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no fallback path: the HIP kernels are the implementation.")
    return ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


lib = _load()

P = c_void_p
_SIGS = {
    "spk_version": (c_int, []),
    "spk_error_string": (c_char_p, [c_int]),
    "spk_lif_fwd": (c_int, [P, P, P, c_int, c_longlong, c_float, c_float, c_float, c_int, P]),
    "spk_lif_fwd_ex": (c_int, [P, P, P, P, c_int, c_longlong, c_float, c_float, c_float, c_int, c_int, P]),
    "spk_bn_prepare": (c_int, [P, P, P, P, c_float, P, P, c_int, P]),
    "spk_bn_eval_fwd": (c_int, [P, P, P, P, c_longlong, c_int, c_int, P]),
    "spk_conv2d_fwd": (c_int, [P, P, P, P, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_conv_transpose2d_fwd": (c_int, [P, P, P, P, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                         c_int, P]),
    "spk_memout_fwd": (c_int, [P, P, P, c_int, c_longlong, P]),
    "spk_lif_train_fwd": (c_int, [P, P, P, P, P, c_int, c_longlong, c_float, c_float, c_float, P]),
    "spk_lif_train_bwd": (c_int, [P, P, P, P, P, c_int, c_longlong, c_float, c_float, c_float, c_float, c_int, P]),
    "spk_bn_lif_train_ws_bytes": (c_longlong, [c_int, c_int, c_int]),
    "spk_bn_lif_train_fwd": (c_int, [P, P, P, P, P, c_float, c_float, P, P, P, P, P, P, c_longlong, c_int, c_int, c_int,
                                     c_int, c_float, c_float, c_float, P]),
    "spk_bn_lif_train_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_longlong, c_int, c_int, c_int, c_int,
                                     c_float, c_float, c_float, c_float, c_int, P]),
    "spk_bn_lif_train_bwd_strided": (c_int, [P, c_longlong, c_longlong, P, P, P, P, P, P, P, P, P, P, P, P, c_longlong, c_int,
                                             c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_int, P]),
    "spk_bn_lif_train_fwd_c4": (c_int, [P, P, P, P, P, c_float, c_float, P, P, P, P, P, P, P, c_longlong, c_int, c_int, c_int,
                                        c_int, c_float, c_float, c_float, P]),
    "spk_den_pack_weight_fp6_cl_multi": (c_int, [P, P, P, P, P, P, P, c_int, P]),
    "spk_conv3x3_dgrad_f16x2_pack_multi": (c_int, [P, P, P, P, P, P, c_int, P]),
    "spk_conv3x3_dgrad_f16x2_prepacked": (c_int, [P, P, c_longlong, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_den_conv3x3_fp6_raw": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_spikes_nhwc_to_fp4": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_spikes_nhwc_to_fp4_counts": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "spk_conv3x3_wgrad_ws_bytes": (c_longlong, [c_int, c_int, c_int]),
    "spk_conv3x3_wgrad_bf16": (c_int, [P, P, P, c_longlong, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_conv3x3_dgrad_ws_bytes": (c_longlong, [c_int, c_int]),
    "spk_conv3x3_dgrad_bf16": (c_int, [P, P, P, c_longlong, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_conv3x3_dgrad_f16x2": (c_int, [P, P, P, c_longlong, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_psp": (c_int, [P, P, c_int, c_longlong, c_float, c_int, P]),
    "spk_masked_ce": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "spk_spikes_to_ptc": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_ptc_to_spikes": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_conv_out_size": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "spk_lif_const_input_table": (c_int, [P, P]),
    "spk_pack_conv_weight": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_conv_fused_fwd": (c_int, [P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_int, P, P, P]),
    "spk_den_packed_weight_bytes": (c_longlong, [c_int, c_int]),
    "spk_den_pack_weight_i8": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "spk_den_conv3x3_mfma": (c_int, [P, c_int, P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int,
                                     c_int, P, P]),
    "spk_den_packed_weight_fp6_bytes": (c_longlong, [c_int, c_int]),
    "spk_den_pack_weight_fp6": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "spk_den_pack_weight_fp6_cl": (c_int, [P, P, P, P, P, c_int, c_int, P]),
    "spk_den_conv3x3_mfma_fp6": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "spk_den_packed_weight_fp6v2_bytes": (c_longlong, [c_int, c_int]),
    "spk_den_pack_weight_fp6v2": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P]),
    "spk_den_fp6v2_flag_words": (c_longlong, [c_int, c_int, c_int, c_int]),
    "spk_den_conv3x3_mfma_fp6v2": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_int, P]),
    "spk_den_conv3x3_mfma_fp6v2_part": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int,
                                                c_int, P]),
    "spk_spikes_to_s32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_s32_to_spikes": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_spikes_to_fp4": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_fp4_to_spikes": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_den_conv3x3_counts_mfma": (c_int, [P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "spk_conv_packed_weight_i8_bytes": (c_longlong, [c_int, c_int, c_int]),
    "spk_pack_conv_weight_i8": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "spk_conv_mfma_fused_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_vq_readout_argmin": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_vq_argmin": (c_int, [P, P, P, c_longlong, c_int, c_int, P]),
    "spk_embedding_fwd": (c_int, [P, P, P, c_longlong, c_int, c_int, c_int, c_int, P]),
    "spk_vq_train_ws_bytes": (c_longlong, []),
    "spk_vq_train_readout": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "spk_vq_train_quant": (c_int, [P, P, P, P, P, c_float, P, c_longlong, c_int, c_int, P]),
    "spk_psp_loss_fwd": (c_int, [P, P, P, c_float, c_float, P, c_int, c_longlong, P]),
    "spk_psp_loss_bwd": (c_int, [P, P, P, P, P, c_float, c_float, c_int, c_longlong, P]),
    "spk_recon_loss_fwd": (c_int, [P, P, P, P, P, P, c_int, c_longlong, P]),
    "spk_recon_loss_bwd": (c_int, [P, P, P, P, P, c_int, c_longlong, P]),
    "spk_vq_train_bwd": (c_int, [P, P, P, P, P, P, P, P, c_float, P, P, P, P, c_int, c_longlong, c_int, c_int, c_int, P]),
    "spk_select_active": (c_int, [P, c_int, P, c_ulonglong, c_ulonglong, P, P, P, c_int, c_int, c_int, P]),
    "spk_readout_collapsed_fwd": (c_int, [P, P, P, c_float, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                          P]),
    "spk_conv_mfma_fused_lif_s32": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_int, c_int, P]),
    "spk_vae_fp6_packed_bytes": (c_longlong, [c_int, c_int]),
    "spk_vae_fp6_pack": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "spk_vae_fp6_flag_words": (c_longlong, [c_int, c_int, c_int, c_int]),
    "spk_ptc_to_s32": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spk_spikegen_table_bytes": (c_longlong, [c_int, c_int]),
    "spk_spikegen_tokens_s32": (c_int, [P, P, P, P, P, P, P, c_int, P, c_int, c_longlong, c_int, c_int, c_int, P]),
    "spk_vae_fp6_fwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_select_needed_bytes": (c_longlong, [c_int, c_int]),
    "spk_select_needed": (c_int, [P, c_int, P, c_ulonglong, c_ulonglong, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_den_conv3x3_mfma_fp6v2_listed": (c_int, [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P,
                                                  P, c_int, c_int, c_int, P]),
    "spk_den_build_input": (c_int, [P, P, P, c_longlong, P, c_int, c_int, P, P, P]),
    "spk_psample_step": (c_int, [P, P, P, c_int, c_float, P, P, c_ulonglong, c_ulonglong, P, P, c_int, c_int, c_int,
                                 P, P, P, P]),
    "spk_den_step_tail": (c_int, [P, c_int, P, c_int, P, P, P, P, P, P, c_int, c_float, P, P, c_ulonglong, c_ulonglong, P, P, P,
                                  P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "spk_philox_noise": (c_int, [c_ulonglong, c_ulonglong, P, P, P, c_int, c_int, c_int, P]),
    "spk_checksum_multi": (c_int, [P, c_int, P, P]),
    "spk_clock_probe": (c_int, [P, c_int, c_int, P]),
    "spk_count_spikes": (c_int, [P, c_longlong, c_longlong, c_int, c_int, P, P]),
    "spk_conv3x3_wgrad_small_ws_bytes": (c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    "spk_conv3x3_wgrad_small": (c_int, [P, P, P, c_longlong, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spk_conv_train_gather_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "spk_conv_train_gather": (c_int, [P, P, P, P] + [c_int] * 11 + [c_longlong] * 3 + [P]),
    "spk_conv_train_wgrad_ws_bytes": (c_longlong, [c_int] * 6),
    "spk_conv_train_wgrad": (c_int, [P, P, P, c_longlong, P, P] + [c_int] * 10 + [c_longlong] * 3 + [c_int, P]),
    "spk_q_sample": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_float, P]),
}
# `make variants` builds only (include/spkdiff_variants.h): the shipped library exports neither and keeps no process-wide state
_VARIANT_SIGS = {
    "spk_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "spk_get_option": (c_int, [ctypes.c_char_p, P]),
}

EXPORTS = tuple(_SIGS)

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here = header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args
HAS_OPTIONS = hasattr(lib, "spk_set_option")          # True only for a variants build loaded through SPKDIFF_LIB
if HAS_OPTIONS:
    for _name, (_res, _args) in _VARIANT_SIGS.items():
        _fn = getattr(lib, _name)
        _fn.restype = _res
        _fn.argtypes = _args


def check(rc: int, what: str = ""):
    """Translate a C-ABI return code into the exception the reference surface would raise."""
    if rc == 0:
        return
    msg = lib.spk_error_string(rc).decode()
    if rc == -1:
        raise ValueError(f"{what}: {msg}")
    if rc == -2:
        raise NotImplementedError(f"{what}: {msg}")
    raise SpkdiffError(f"{what}: HIP error {rc}: {msg}")


def version() -> int:
    return lib.spk_version()


# The signatures declared above are those of include/spkdiff.h at this version.  A stale libspkdiff.so or an SPKDIFF_LIB A/B
# variant built from another header would take arguments at the wrong positions (silently wrong results): refuse it here.
EXPECTED_VERSION = 105
if version() != EXPECTED_VERSION:
    raise ImportError(f"spkdiff: {LIB_PATH} reports C-ABI version {version()}, this binding declares version "
                      f"{EXPECTED_VERSION} (include/spkdiff.h SPK_VERSION). Rebuild the library: make -C "
                      f"{os.path.join(os.path.dirname(_HERE), 'csrc')}")


# Measurement options (include/spkdiff_variants.h): settable only in a `make variants` library (SPKDIFF_LIB=.../variants/
# libspkdiff_variants.so); the A/B tools under tools/ select a launch form with SPKDIFF_<NAME>=<int>, forwarded here ONCE at import.
# The shipped library has the defaults compiled in: set_option raises NotImplementedError there.
OPTIONS = ("v2_waves", "v2_lag", "v2_duo", "v2_defer", "v2_lps", "fp6_waves", "fp6_xcd_walk", "conv6_shared", "conv6_shared_dyn", "mfma_debug")
OPTION_DEFAULTS = {"conv6_shared": 1, "conv6_shared_dyn": 1, "mfma_debug": 0, "fp6_xcd_walk": 1, "fp6_waves": 4, "v2_waves": 8,
                   "v2_lag": 0, "v2_duo": 0, "v2_defer": 0, "v2_lps": 1}


def set_option(name: str, value: int):
    if not HAS_OPTIONS:
        raise NotImplementedError(f"spkdiff: option {name!r} is a compile-time constant of the shipped library; build "
                                  "`make -C spiking-diffusion_amd/csrc variants` and load it through SPKDIFF_LIB")
    check(lib.spk_set_option(name.encode(), int(value)), f"spk_set_option({name!r})")


def get_option(name: str) -> int:
    if not HAS_OPTIONS:
        if name not in OPTION_DEFAULTS:
            raise NotImplementedError(f"spkdiff: unknown option {name!r}")
        return OPTION_DEFAULTS[name]
    v = c_int(0)
    check(lib.spk_get_option(name.encode(), ctypes.byref(v)), f"spk_get_option({name!r})")
    return int(v.value)


for _o in OPTIONS:
    _e = os.environ.get("SPKDIFF_" + _o.upper()) or (os.environ.get("SPK_MFMA_DEBUG") if _o == "mfma_debug" else None)
    if _e not in (None, ""):
        set_option(_o, int(_e))
