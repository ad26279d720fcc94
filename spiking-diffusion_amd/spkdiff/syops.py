"""Synaptic-operation (energy) report of one denoiser call in the reference's "syops" convention
(R/syops/ops.py:14-24 ``spike_rate``, :121-158 ``conv_syops_counter_hook``, reported by R/main.py:332): for every
convolution ``overall = (k*k*Cin*Cout + Cout) * T*B*H*W``; a layer whose input is binary spikes performs
``overall * input_spike_rate`` accumulates (ACs), any other layer ``overall`` multiply-accumulates (MACs).

The spike rates come from the packed spike maps the fused kernels already produce (a recorded call), counted on the
device by ``spk_count_spikes`` without unpacking them; SURVEY.md §8(f) item 4.  This is the fused fast path for the
denoiser; the whole-model report with the reference's entry point and per-module conventions is the ``syops`` package.
"""
from __future__ import annotations

import torch

from . import ops


def _rate(spikes_any_layout) -> float:
    st = ops.count_spikes(spikes_any_layout)
    return st["total"] / st["numel"]


@torch.no_grad()
def denoiser_syops(den, x_t: torch.Tensor, t):
    """One ``DummyModel`` call on tokens ``x_t`` [B,1,h,w] at diffusion step ``t`` (an int, or a [B] tensor of steps).

    Returns ``(logits, report)``; ``report`` is a list of dicts, one per conv layer, with keys ``layer``, ``overall``
    (synaptic operations), ``acs``, ``macs``, ``in_rate`` (None for a non-spike input), ``out_rate`` (None for conv6),
    plus a final ``{"layer": "total", ...}`` row."""
    rec = []
    from spikingjelly.activation_based import functional
    functional.reset_net(den)
    tt = t if isinstance(t, torch.Tensor) else int(t)
    logits = den._run(ops.den_build_input(x_t, tt), stateful=False, record=rec)
    B, _, H, W = x_t.shape
    T = den.n_steps
    rates = [_rate(s) for s in rec]                       # outputs of conv1..conv5
    convs = [den.conv1[0], den.conv2[0], den.conv3[0], den.conv4[0], den.conv5[0], den.conv6[0]]
    cin6 = convs[5].in_channels
    c5, c1 = convs[4].out_channels, convs[0].out_channels
    # conv6 reads cat(x5, x1): its input spike rate is the channel-weighted mean of both
    in_rates = [None, rates[0], rates[1], rates[2], rates[3], (rates[4] * c5 + rates[0] * c1) / cin6]
    out_rates = rates + [None]
    report, tot = [], {"layer": "total", "overall": 0, "acs": 0.0, "macs": 0}
    for i, conv in enumerate(convs):
        k = conv.kernel_size[0] * conv.kernel_size[1]
        per_pos = k * conv.in_channels * conv.out_channels + (conv.out_channels if conv.bias is not None else 0)
        overall = per_pos * T * B * H * W
        acs = overall * in_rates[i] if in_rates[i] is not None else 0.0
        macs = overall if in_rates[i] is None else 0
        report.append({"layer": f"conv{i + 1}", "overall": overall, "acs": acs, "macs": macs,
                       "in_rate": in_rates[i], "out_rate": out_rates[i]})
        tot["overall"] += overall; tot["acs"] += acs; tot["macs"] += macs
    report.append(tot)
    return logits, report
