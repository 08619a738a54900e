"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Seeded synthetic inputs shared by the fixture generator
(tests/golden/make_golden.py) and the tests, so fixtures need to store outputs only.
numpy PCG64 streams are platform independent."""
import math

import numpy as np
import torch

from .nets_cpu import DecompValues


def level_sizes(h, w, height):
    """Band-level sizes finest-first and the low-residual size: ceil(d / 2^(k/2))."""
    lv = [(math.ceil(h / 2 ** (k / 2) - 1e-9), math.ceil(w / 2 ** (k / 2) - 1e-9)) for k in range(height - 1)]
    return lv[:-1], lv[-1]


def synthetic_vals(seed, n_img, h, w, height, nbands=4):
    """A DecompValues in the per-image layout of Pyramid.filter (reference src/train/pyramid.py:48-78)."""
    rng = np.random.default_rng(seed)
    bands, low = level_sizes(h, w, height)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    high = f(n_img, 1, h, w)
    phase = [torch.from_numpy(rng.uniform(-np.pi, np.pi, (n_img * nbands, 1, a, b)).astype(np.float32))
             for a, b in bands]
    amp = [f(n_img * nbands, 1, a, b).abs() * (1 + k) for k, (a, b) in enumerate(bands)]
    lowt = f(n_img, 1, *low).abs() + 0.1
    return DecompValues(high, phase, amp, lowt)


def frames(seed, n, c, h, w):
    rng = np.random.default_rng(seed)
    return [torch.from_numpy(rng.random((n, c, h, w), dtype=np.float32)) for _ in range(2)]


def translating_pair(seed, h, w, shift=(3.5, -2.25), n_waves=6):
    """Synthetic frame triplet: sum of random band-limited sinusoid textures plus a
    smooth gradient; frame k is the pattern translated by k*shift/2, so the middle frame is analytic.
    Returns (frame0, frame1_true, frame2) as (3,H,W) float32 in [0,1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    waves = [(rng.uniform(0.02, 0.25) * np.pi, rng.uniform(0, np.pi), rng.uniform(0, 2 * np.pi),
              rng.uniform(0.3, 1.0, 3)) for _ in range(n_waves)]

    def render(dy, dx):
        img = np.zeros((3, h, w))
        for freq, ang, ph, col in waves:
            arg = freq * (np.cos(ang) * (xx - dx) + np.sin(ang) * (yy - dy)) + ph
            img += col[:, None, None] * np.sin(arg)[None]
        img = img / (2.0 * n_waves) * 1.2 + 0.5
        img += 0.15 * ((xx - dx) / w - 0.5)[None] + 0.1 * ((yy - dy) / h - 0.5)[None]
        return np.clip(img, 0.0, 1.0).astype(np.float32)

    return render(0, 0), render(shift[0] / 2, shift[1] / 2), render(shift[0], shift[1])
