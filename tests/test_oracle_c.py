"""CPU tests: the plain-C restatement (oracle/lif_ref.c) against the golden fixtures captured from the reference
and against the torch oracle -- a third, code-independent implementation of the exact (threshold / index) arithmetic."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import snn_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clib():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_lif_matches_golden_f1(clib, golden_dir):
    d = np.load(os.path.join(golden_dir, "f1_lif.npz"))
    x = np.ascontiguousarray(d["x_seq"])
    T, N = x.shape
    v = np.zeros(N, np.float32)
    s = np.zeros((T, N), np.uint8)
    clib.lif_ref(_p(x), _p(v), _p(s), T, ctypes.c_longlong(N), ctypes.c_float(2.0), ctypes.c_float(1.0), ctypes.c_float(0.0))
    want = np.unpackbits(d["spikes"])[: T * N].reshape(T, N)
    assert np.array_equal(s, want) and np.array_equal(v, d["v"])
    x2 = np.ascontiguousarray(x[::-1])
    clib.lif_ref(_p(x2), _p(v), _p(s), T, ctypes.c_longlong(N), ctypes.c_float(2.0), ctypes.c_float(1.0), ctypes.c_float(0.0))
    assert np.array_equal(s, np.unpackbits(d["spikes_carry"])[: T * N].reshape(T, N)) and np.array_equal(v, d["v_carry"])


def test_c_bn_matches_golden_f7(clib, golden_dir):
    d = np.load(os.path.join(golden_dir, "f7_bn.npz"))
    x = np.ascontiguousarray(d["x"])
    T, B, C, H, W = x.shape
    y = np.empty_like(x)
    clib.bn_fma_ref(_p(x), _p(d["weight"]), _p(d["bias"]), _p(d["running_mean"]), _p(d["running_var"]),
                    ctypes.c_float(1e-5), _p(y), ctypes.c_longlong(T * B), C, H * W)
    assert np.array_equal(y, d["y"])


def test_c_vq_argmin_matches_torch_oracle(clib):
    g = torch.Generator().manual_seed(1)
    x = torch.rand(300, 16, generator=g) * 2
    cb = torch.randn(128, 16, generator=g)
    idx = np.zeros(300, np.int64)
    clib.vq_argmin_ref(_p(x.numpy()), _p(cb.numpy()), _p(idx), ctypes.c_longlong(300), 16, 128)
    assert np.array_equal(idx, ref.vq_distances(x.double(), cb.double()).argmin(1).numpy())


def test_c_psample_matches_golden_f6(clib, golden_dir):
    d = np.load(os.path.join(golden_dir, "f6_psample.npz"))
    B = int(d["B"])
    x = np.full(B * 49, 128, np.int64)
    un = np.zeros(B * 49, np.uint8)
    for i, t in enumerate(d["ts"]):
        logits = np.ascontiguousarray(np.transpose(d["logits"][i], (0, 3, 1, 2)))      # [B,K,h,w]
        clib.psample_ref(_p(logits), _p(x), _p(un), int(t), ctypes.c_float(1.0), _p(np.ascontiguousarray(d["u"][i])),
                         _p(np.ascontiguousarray(d["q"][i])), B, 49, 128)
        assert np.array_equal(x, d["x_after"][i].reshape(-1)), f"t={t}"
        assert np.array_equal(un.astype(bool), d["unmasked_after"][i].reshape(-1))
