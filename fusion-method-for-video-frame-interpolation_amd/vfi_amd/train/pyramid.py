"""Pyramid -- mirror of reference src/train/pyramid.py (`Pyramid(height, nbands, scale_factor, device)`,
`.filter(img[N,H,W]) -> DecompValues`, `.inv_filter(vals) -> [N,H,W]`, attrs .height/.nbands/.device/.pyr).

One library call per direction: `vfi_pyr_analyze` / `vfi_pyr_synthesize` (csrc/vfi_pyramid.hip) run the FFTs and
the fused mask / crop / shift / polar kernels; `coeff_to_values` / `values_to_coeff` (pyramid.py:48-112)
are fused in, so neither the coefficient lists nor the deepcopy (pyramid.py:49) exist.  Level geometry,
mask tables and FFT plans live in a plan cached per (H, W) instead of being rebuilt per frame
(reference src/fusion_net/interpolate_twoframe.py:124-129).

Extensions used by this package's own per-frame driver (not in the reference surface):
  * `filter(..., concat_frames=F)` writes the bands directly in PhaseNet's block-input layout
    (what separate_vals + get_concat_layers_inf + normalize_vals' phase/pi produce, src/train/utils.py:47-127);
  * entries of `vals.phase/amplitude` (or high_level / low_level) that are not tensors (the scalar 0 the
    reference itself uses for missing levels, src/phase_net/phase_net.py:91-93) are treated as zeros and
    their transforms are skipped.
"""
import ctypes
import math

import torch

from .. import _lib
from .._lib import VfiLibraryError
from ..steerable.SCFpyr_PyTorch import SCFpyr_PyTorch
from ..values import DecompValues

__all__ = ["DecompValues", "Pyramid"]


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if torch.is_tensor(t) else None
    return arr


class Pyramid:
    """Steerable Pyramid Decomposition (reference src/train/pyramid.py:20-46)."""

    def __init__(self, height, nbands, scale_factor, device):
        self.height = height
        self.nbands = nbands
        self.scale_factor = scale_factor
        self.device = torch.device(device)
        self.pyr = SCFpyr_PyTorch(height=height, nbands=nbands, scale_factor=scale_factor, device=self.device)

    # -- analysis -------------------------------------------------------------------------------------
    def filter(self, img, concat_frames=None, phase_scale=1.0, level_mask=None, want_high=True, want_low=True, amp_max_eps=None):
        """Psi filter.  img (N,H,W) -> DecompValues in the per-image layout: high (N,1,H,W),
        phase/amplitude[k] (N*nbands,1,h_k,w_k) finest first with index img*nbands+band, low (N,1,hL,wL).

        concat_frames=F (N = F*C images ordered frame-major): PhaseNet layout instead -- phase/amplitude[k]
        are (C, F*nbands, h, w) views (channels [f0 b0..b3, f1 b0..b3]) of block-input buffers, lists ordered
        COARSEST first, high (C,F,H,W), low (C,F,hL,wL); see PhaseNet.normalize_vals."""
        if img.dim() != 3:
            raise VfiLibraryError("Pyramid.filter expects (N,H,W)")
        img = img.contiguous()
        n, h, w = img.shape
        plan = self.pyr.plan(h, w, n)
        nlev, nb = self.height - 2, self.nbands
        sizes = plan.sizes
        mask = (1 << nlev) - 1 if level_mask is None else int(level_mask)
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=img.device)
        if concat_frames is None:
            phase = [new(n * nb, 1, *sizes[k]) if (mask >> k) & 1 else 0 for k in range(nlev)]
            amp = [new(n * nb, 1, *sizes[k]) if (mask >> k) & 1 else 0 for k in range(nlev)]
            high = new(n, 1, h, w) if want_high else 0
            low = new(n, 1, *sizes[nlev]) if want_low else 0
            plan.analyze(img, high, phase, amp, None, low, phase_scale, mask, 0)
            return DecompValues(high, phase, amp, low)
        f = int(concat_frames)
        c = n // f
        if c * f != n:
            raise VfiLibraryError("concat_frames must divide the number of images")
        # block-input buffers [feature 64 | prediction P | phase f*nb | amp f*nb], P = 1 at the coarsest level
        bufs, phase, amp, table = [], [], [], []
        for k in range(nlev):
            p_prev = 1 if k == nlev - 1 else 8
            ctot = 64 + p_prev + 2 * f * nb
            buf = new(c, ctot, *sizes[k])
            bufs.append(buf)
            phase.append(buf[:, 64 + p_prev:64 + p_prev + f * nb])
            amp.append(buf[:, 64 + p_prev + f * nb:])
            # image d = frame*c + colour -> batch colour, channel frame*nb (+ band); planes from the view's base
            table += [(d % c) * ctot + (d // c) * nb for d in range(n)]
        high = new(c, f, h, w) if want_high else 0
        low = new(c, f, *sizes[nlev]) if want_low else 0
        hi_tmp = new(n, h, w) if want_high else 0
        lo_tmp = new(n, *sizes[nlev]) if want_low else 0
        amp_max = None
        if amp_max_eps is not None:        # (levels finest first, colours): max amplitude + eps, for PhaseNet.normalize_vals
            amp_max = new(nlev, c)
        plan.analyze(img, hi_tmp, phase, amp, table, lo_tmp, phase_scale, mask, 0, amp_max=amp_max, groups=c,
                     eps=amp_max_eps if amp_max_eps is not None else 0.0)
        if want_high:
            high.copy_(hi_tmp.view(f, c, h, w).transpose(0, 1))
        if want_low:
            low.copy_(lo_tmp.view(f, c, *sizes[nlev]).transpose(0, 1))
        out = DecompValues(high, phase[::-1], amp[::-1], low)
        if amp_max is not None:
            return out, bufs[::-1], amp_max.flip(0).contiguous()      # coarsest first, like the lists
        return out, bufs[::-1]

    def band_filter(self, img, level_mask, keep_high=False, keep_low=False):
        """== inv_filter(keep(filter(img))) where `keep` zeroes every band level not in level_mask and the
        high / low residuals unless kept (get_last_value_levels / get_first_value_levels applied to UNMODIFIED
        values): a single radial frequency-domain gain (see vfi_pyr_plan_prepare_filter).  img (N,H,W)."""
        img = img.contiguous()
        n, h, w = img.shape
        return self.pyr.plan(h, w, n).band_filter(img, level_mask, keep_high, keep_low)

    def band_filter_pair(self, img_a, spec_a, img_b, spec_b):
        """band_filter(img_a, **spec_a) + band_filter(img_b, **spec_b) (the sum of two unmodified level subsets of two
        image sets, e.g. the reference's "baseline" mix) with one inverse transform.  spec = dict(level_mask=...,
        keep_high=..., keep_low=...)."""
        img_a, img_b = img_a.contiguous(), img_b.contiguous()
        if img_a.shape != img_b.shape:
            raise VfiLibraryError("band_filter_pair: shape mismatch")
        n, h, w = img_a.shape
        tup = lambda d: (d["level_mask"], d.get("keep_high", False), d.get("keep_low", False))
        return self.pyr.plan(h, w, 2 * n).band_filter_pair(img_a, tup(spec_a), img_b, tup(spec_b))

    # -- synthesis -------------------------------------------------------------------------------------
    def inv_filter(self, vals):
        """Psi^{-1} filter: per-image DecompValues -> (N,H,W)."""
        tensors = [t for t in list(vals.phase) + [vals.high_level, vals.low_level] if torch.is_tensor(t)]
        if not tensors:
            raise VfiLibraryError("inv_filter: all-zero values carry no shape")
        nlev, nb = self.height - 2, self.nbands
        if len(vals.phase) != nlev:
            raise VfiLibraryError(f"inv_filter: expected {nlev} band levels, got {len(vals.phase)}")
        if torch.is_tensor(vals.high_level):
            n, _, h, w = vals.high_level.shape
        else:
            k0 = next(k for k, p in enumerate(vals.phase) if torch.is_tensor(p))
            n = vals.phase[k0].shape[0] // nb
            h, w = self._full_size
        plan = self.pyr.plan(h, w, n)
        mask = 0
        phase, amp = [], []
        for k in range(nlev):
            p, a = vals.phase[k], vals.amplitude[k]
            if torch.is_tensor(p) and torch.is_tensor(a):
                if tuple(p.shape) != (n * nb, 1, *plan.sizes[k]):
                    raise VfiLibraryError(f"inv_filter: level {k} has shape {tuple(p.shape)}, expected "
                                          f"{(n * nb, 1, *plan.sizes[k])}")
                mask |= 1 << k
                phase.append(p.contiguous()); amp.append(a.contiguous())
            else:
                phase.append(None); amp.append(None)
        high = vals.high_level.contiguous() if torch.is_tensor(vals.high_level) else None
        low = vals.low_level.contiguous() if torch.is_tensor(vals.low_level) else None
        img = torch.empty((n, h, w), dtype=torch.float32, device=tensors[0].device)
        plan.synthesize(high, phase, amp, None, low, mask, 0, img)
        return img

    _full_size = None

    def set_full_size(self, h, w):
        """Only needed to invert values whose high_level was dropped (no tensor carries H, W)."""
        self._full_size = (h, w)
