"""Oracle (oracle/adacof_cpu.c) against the fixtures produced by the reference's own
specialised kernel text (tests/golden/make_golden.py::gen_adacof)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import adacof_cpu

CASES = sorted(glob.glob(os.path.join(GOLDEN, "adacof_sampling_*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[16:-4] for p in CASES])
def test_sampling_bit_exact_against_reference_kernel(path):
    g = np.load(path)
    out = adacof_cpu.adacof_forward(g["input"], g["weight"], g["offset_i"], g["offset_j"], int(g["dilation"]))
    assert np.array_equal(out, g["output"])  # fp32, same evaluation order -> bit exact


def test_fixture_set_is_present():
    assert len(CASES) >= 4


def test_blend_mask_matches_numpy_statement():
    # fusion_adacofnet.py:198-213 written with numpy broadcasting, as the reference does with torch
    rng = np.random.default_rng(3)
    n, k, h, w = 2, 25, 6, 10
    lg = rng.standard_normal((2, n, k, h, w)).astype(np.float32)
    W = (np.exp(lg) / np.exp(lg).sum(2, keepdims=True)).astype(np.float32)
    A = (rng.standard_normal((2, n, k, h, w)) * 3).astype(np.float32)
    B = (rng.standard_normal((2, n, k, h, w)) * 3).astype(np.float32)
    t1 = rng.random((n, 3, h, w), dtype=np.float32)
    t2 = rng.random((n, 3, h, w), dtype=np.float32)
    occ = rng.random((n, 1, h, w), dtype=np.float32)
    frame, mask = adacof_cpu.blend_mask(t1, t2, occ, W[0], A[0], B[0], W[1], A[1], B[1])
    np.testing.assert_allclose(frame, occ * t1 + (1 - occ) * t2, rtol=0, atol=1e-7)
    var = []
    for s in range(2):
        dp = np.stack([A[s], B[s]], 0)                      # (2,N,K,H,W)
        mean = (W[s] * dp).sum(-3)                          # (2,N,H,W)
        v = (W[s] * (mean[:, :, None] - dp) ** 2).sum(-3)   # (2,N,H,W)
        var.append(v.sum(0))
    ref = np.clip(np.maximum(var[0], var[1]), 0, 20) / 20
    np.testing.assert_allclose(mask[:, 0], ref, rtol=1e-5, atol=1e-6)
