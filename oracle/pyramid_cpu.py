"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).          *** PARITY UNPINNED ***

Complex steerable pyramid in the frequency domain (torch.fft on CPU), the arithmetic behind
`Pyramid.filter / inv_filter` (reference call sites src/train/pyramid.py:28-33,37,44).

The reference does NOT contain this arithmetic: it imports `steerable.SCFpyr_PyTorch` from a
third-party package (upstream tomrunia/PyTorchSteerablePyramid) that is un-vendored, un-pinned
(absent from requirements.txt / environment.yml) and absent from this image, in a fork that accepts
`scale_factor=sqrt(2)` with height 12 on 256x256 crops (src/train/pyramid.py:23-33,
src/train/utils.py:168-171) -- which stock upstream rejects.  The reference holds no test or golden
vector at this boundary.  This file therefore restates the PUBLISHED upstream algorithm
(SCFpyr_PyTorch.build / reconstruct, math_utils.prepare_grid / rcosFn / pointOp) and generalises the
per-level subsampling from 1/2 to 1/scale_factor:

    level k (k = 0 finest) works on the centred (h_k x w_k) window of the fft-shifted spectrum,
    h_k = ceil(H / s^k), w_k = ceil(W / s^k), DC at index (h_k // 2, w_k // 2),
    radial transition shifted by log2(s) per level (upstream: `Xrcos - log2(scale_factor)`).

After level k's low-pass the spectrum is supported on radius < s^-(k+1) (in Nyquist units), so the
1/s window is alias free and reconstruction telescopes exactly as upstream.  With s = 2 and even
sizes the windows coincide with upstream's `ceil((d-0.5)/2)` rule.  Pinned by properties only
(tests/test_oracle_pyramid.py): perfect reconstruction, level-shape table, analytic bands, energy.
"""
import math
from math import factorial

import numpy as np
import torch

from .layout_cpu import coeff_to_values, values_to_coeff  # noqa: F401  (adapters live with the layouts)


# ---- upstream math_utils ---------------------------------------------------------------------
def prepare_grid(m, n):
    x = np.linspace(-(m // 2) / (m / 2), (m // 2) / (m / 2) - (1 - m % 2) * 2 / m, num=m)
    y = np.linspace(-(n // 2) / (n / 2), (n // 2) / (n / 2) - (1 - n % 2) * 2 / n, num=n)
    xv, yv = np.meshgrid(y, x)
    angle = np.arctan2(yv, xv)
    rad = np.sqrt(xv ** 2 + yv ** 2)
    rad[m // 2][n // 2] = rad[m // 2][n // 2 - 1]
    return np.log2(rad), angle


def rcos_fn(width=1, position=-0.5):
    n = 256
    x = np.pi * np.arange(-n - 1, 2) / 2 / n
    y = np.cos(x) ** 2
    y[0] = y[1]
    y[n + 2] = y[n + 1]
    return position + 2 * width / np.pi * (x + np.pi / 4), y


def point_op(im, y, x):
    return np.interp(im.flatten(), x, y).reshape(im.shape)


# ---- plan: every mask as float32 tables, in fft-shifted window coordinates --------------------
class PyramidSpec:
    """All level geometry + masks for one (H, W, height, nbands, scale_factor)."""

    def __init__(self, H, W, height, nbands=4, scale_factor=math.sqrt(2), size_rule="ceil", dc_rule="half"):
        """size_rule / dc_rule select spec VARIANTS for tools/pyramid_spec_pin.py only (the functional pin of the
        self-defined spec against the reference's trained phase_net.pt); the product implements ("ceil", "half")."""
        self.H, self.W, self.height, self.nbands, self.s = H, W, height, nbands, float(scale_factor)
        self.nlev = height - 2
        if self.nlev < 1:
            raise ValueError("height must be >= 3")
        rnd = {"ceil": lambda v: math.ceil(v - 1e-9), "floor": lambda v: math.floor(v + 1e-9),
               "round": lambda v: math.floor(v + 0.5)}[size_rule]
        size = lambda d, k: int(rnd(d / self.s ** k))
        self.sizes = [(size(H, k), size(W, k)) for k in range(self.nlev + 1)]   # [nlev] = low residual
        if min(self.sizes[-1]) < 2:
            raise ValueError(f"pyramid height {height} too large for {H}x{W}")
        dc = (lambda d: d // 2) if dc_rule == "half" else (lambda d: (d - 1) // 2)
        self.starts = [(H // 2 - (dc(h) if h < H else h // 2), W // 2 - (dc(w) if w < W else w // 2)) for h, w in self.sizes]
        log_rad, angle = prepare_grid(H, W)
        xr, yr = rcos_fn(1, -0.5)
        yr = np.sqrt(yr)
        yir = np.sqrt(np.abs(1 - yr ** 2))
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()
        self.lo0 = f32(point_op(log_rad, yir, xr))
        self.hi0 = f32(point_op(log_rad, yr, xr))
        lut = 1024
        xcosn = np.pi * np.arange(-(2 * lut + 1), lut + 2) / lut
        alpha = (xcosn + np.pi) % (2 * np.pi) - np.pi
        order = nbands - 1
        const = 2 ** (2 * order) * factorial(order) ** 2 / (nbands * factorial(2 * order))
        ycos_a = 2 * np.sqrt(const) * np.cos(xcosn) ** order * (np.abs(alpha) < np.pi / 2)   # analysis (one sided)
        ycos_s = np.sqrt(const) * np.cos(xcosn) ** order                                      # synthesis
        self.himask, self.lomask, self.ang_a, self.ang_s = [], [], [], []
        for k in range(self.nlev):
            xr = xr - np.log2(self.s)
            (h, w), (sy, sx) = self.sizes[k], self.starts[k]
            lr = log_rad[sy:sy + h, sx:sx + w]
            an = angle[sy:sy + h, sx:sx + w]
            self.himask.append(f32(point_op(lr, yr, xr)))
            self.ang_a.append(torch.stack([f32(point_op(an, ycos_a, xcosn + np.pi * b / nbands)) for b in range(nbands)]))
            self.ang_s.append(torch.stack([f32(point_op(an, ycos_s, xcosn + np.pi * b / nbands)) for b in range(nbands)]))
            (h2, w2), (sy2, sx2) = self.sizes[k + 1], self.starts[k + 1]
            self.lomask.append(f32(point_op(log_rad[sy2:sy2 + h2, sx2:sx2 + w2], yir, xr)))

    def crop(self, k):
        """Slice of level k's window that is level k+1's window."""
        (sy, sx), (sy2, sx2), (h2, w2) = self.starts[k], self.starts[k + 1], self.sizes[k + 1]
        return slice(sy2 - sy, sy2 - sy + h2), slice(sx2 - sx, sx2 - sx + w2)


def _fft2s(x):
    return torch.fft.fftshift(torch.fft.fft2(x), dim=(-2, -1))


def _ifft2s(x):
    return torch.fft.ifft2(torch.fft.ifftshift(x, dim=(-2, -1)))


def build(spec, im):
    """upstream SCFpyr_PyTorch.build: im (N,H,W) -> [hi (N,H,W), [nbands x (N,h,w,2)] x nlev, lo (N,hL,wL)]."""
    assert im.shape[-2:] == (spec.H, spec.W)
    dft = _fft2s(im.to(torch.float32))
    lodft = dft * spec.lo0
    coeff = []
    rot = 1j ** (spec.nbands - 1) * (-1) ** (spec.nbands - 1)      # (-i)^(nbands-1), exact for nbands=4: +i
    for k in range(spec.nlev):
        bands = []
        for b in range(spec.nbands):
            banddft = lodft * spec.ang_a[k][b] * spec.himask[k]
            banddft = banddft * rot if spec.nbands != 4 else torch.complex(-banddft.imag, banddft.real)
            bands.append(torch.view_as_real(_ifft2s(banddft)))
        coeff.append(bands)
        ys, xs = spec.crop(k)
        lodft = lodft[:, ys, xs] * spec.lomask[k]
    lo = _ifft2s(lodft).real
    hi = _ifft2s(dft * spec.hi0).real
    return [hi] + coeff + [lo]


def reconstruct(spec, coeff):
    """upstream SCFpyr_PyTorch.reconstruct -> (N,H,W)."""
    n = coeff[0].shape[0]
    rot_exact = spec.nbands == 4                                      # (+i)^3 = -i
    res = _fft2s(coeff[-1].to(torch.float32))
    for k in range(spec.nlev - 1, -1, -1):
        h, w = spec.sizes[k]
        ys, xs = spec.crop(k)
        cur = torch.zeros((n, h, w), dtype=torch.complex64)
        cur[:, ys, xs] = res * spec.lomask[k]
        for b in range(spec.nbands):
            z = _fft2s(torch.view_as_complex(coeff[1 + k][b].contiguous())) * spec.ang_s[k][b] * spec.himask[k]
            z = torch.complex(z.imag, -z.real) if rot_exact else z * (1j ** (spec.nbands - 1))
            cur = cur + z
        res = cur
    out = res * spec.lo0 + _fft2s(coeff[0].to(torch.float32)) * spec.hi0
    return _ifft2s(out).real


class Pyramid:
    """Same surface as reference src/train/pyramid.py:20-46 (CPU, oracle)."""

    def __init__(self, height, nbands=4, scale_factor=math.sqrt(2), device="cpu", **variant):
        self.height, self.nbands, self.scale_factor, self.device = height, nbands, scale_factor, device
        self.variant = variant
        self._specs = {}

    def spec(self, h, w):
        if (h, w) not in self._specs:
            self._specs[(h, w)] = PyramidSpec(h, w, self.height, self.nbands, self.scale_factor, **self.variant)
        return self._specs[(h, w)]

    def filter(self, img):
        return coeff_to_values(build(self.spec(*img.shape[-2:]), img))

    def inv_filter(self, vals):
        return reconstruct(self.spec(*vals.high_level.shape[-2:]), values_to_coeff(vals, self.nbands))
