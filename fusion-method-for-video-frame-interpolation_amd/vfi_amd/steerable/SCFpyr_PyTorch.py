"""`SCFpyr_PyTorch(height, nbands, scale_factor, device)` with `.build(x[N,1,H,W]) -> coeff` and
`.reconstruct(coeff) -> [N,H,W]` -- the surface of the third-party class the reference constructs at
src/train/pyramid.py:28-33 and calls at :37,44.  Coefficient layout as the reference expects it:
coeff = [hi (N,H,W), [nbands x (N,h,w,2)] per level finest first, lo (N,hL,wL)] (pyramid.py:56-61).

Backed by one plan per (H, W) in libvfi_hip.so (level geometry, mask tables, the tables of the hand-written FFT
engines -- no FFT library is linked --, workspace).
"""
import ctypes

import torch

from .. import _lib
from .._lib import VfiLibraryError

BAND_MAJOR, COMPLEX_COEFF = 1, 2


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if torch.is_tensor(t) else None
    return arr


class Plan:
    """RAII handle of a vfi_pyr_plan."""

    def __init__(self, h, w, height, nbands, scale_factor, max_images, device):
        if torch.device(device).type != "cuda":
            raise VfiLibraryError("the pyramid needs a HIP device (vfi_amd has no CPU path)")
        self.h, self.w, self.height, self.nbands, self.max_images = h, w, height, nbands, max_images
        self.device = torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.call("vfi_pyr_plan_create", h, w, height, nbands, float(scale_factor), max_images, ctypes.byref(self._h))
        self.sizes = []
        for k in range(height - 1):
            a, b = ctypes.c_int(), ctypes.c_int()
            _lib.call("vfi_pyr_plan_level_size", self._h, k, ctypes.byref(a), ctypes.byref(b))
            self.sizes.append((a.value, b.value))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().vfi_pyr_plan_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    def _bytes(self, n, mask, high, low):
        """Algorithmic HBM bytes of one transform: image + kept band planes (phase, amp) + residuals."""
        px = self.h * self.w + (self.h * self.w if high else 0) + (self.sizes[-1][0] * self.sizes[-1][1] if low else 0)
        px += sum(2 * self.nbands * a * b for k, (a, b) in enumerate(self.sizes[:-1]) if (mask >> k) & 1)
        return 4.0 * n * px

    def analyze(self, img, high, phase, amp, table, low, phase_scale, mask, flags, amp_max=None, groups=1, eps=0.0):
        """amp_max: optional (levels, groups) float tensor that receives max amplitude + eps per level and image group
        (image d belongs to group d % groups): PhaseNet.normalize_vals' maxima, reduced by the kernel that writes them."""
        n = img.shape[0]
        tab = (ctypes.c_int * len(table))(*table) if table is not None else None
        head = (self._h, _lib.dptr(img, "img"), n, high.data_ptr() if torch.is_tensor(high) else None, _ptr_array(phase),
                _ptr_array(amp) if amp is not None else None, tab, low.data_ptr() if torch.is_tensor(low) else None,
                float(phase_scale), mask, flags)
        work = ("byte", self._bytes(n, mask, torch.is_tensor(high), torch.is_tensor(low)), "pyr_analyze")
        if amp_max is None:
            _lib.call("vfi_pyr_analyze", *head, _lib.stream_ptr(), work=work)
        else:
            _lib.call("vfi_pyr_analyze_max", *head, _lib.dptr(amp_max, "amp_max"), int(groups), float(eps), _lib.stream_ptr(), work=work)

    def band_filter(self, img, level_mask, keep_high, keep_low):
        """real(ifft2(fft2(img) * G)): analysis + synthesis of an unmodified level subset as ONE radial filter."""
        key = (int(level_mask), bool(keep_high), bool(keep_low))
        if not hasattr(self, "_filters"):
            self._filters = {}
        if key not in self._filters:
            fid = ctypes.c_int()
            _lib.call("vfi_pyr_plan_prepare_filter", self._h, key[0], int(key[1]), int(key[2]), ctypes.byref(fid))
            self._filters[key] = fid.value
        out = torch.empty_like(img)
        n = img.shape[0]
        _lib.call("vfi_pyr_apply_filter", self._h, self._filters[key], _lib.dptr(img, "img"), n, out.data_ptr(),
                  _lib.stream_ptr(), work=("byte", 8.0 * n * self.h * self.w, "pyr_band_filter"))
        return out

    def _filter_id(self, level_mask, keep_high, keep_low):
        key = (int(level_mask), bool(keep_high), bool(keep_low))
        if not hasattr(self, "_filters"):
            self._filters = {}
        if key not in self._filters:
            fid = ctypes.c_int()
            _lib.call("vfi_pyr_plan_prepare_filter", self._h, key[0], int(key[1]), int(key[2]), ctypes.byref(fid))
            self._filters[key] = fid.value
        return self._filters[key]

    def band_filter_pair(self, img_a, spec_a, img_b, spec_b):
        """band_filter(img_a, *spec_a) + band_filter(img_b, *spec_b) with one inverse transform; spec = (level_mask,
        keep_high, keep_low)."""
        fa, fb = self._filter_id(*spec_a), self._filter_id(*spec_b)
        out = torch.empty_like(img_a)
        n = img_a.shape[0]
        _lib.call("vfi_pyr_apply_filter_pair", self._h, fa, _lib.dptr(img_a, "img_a"), fb, _lib.dptr(img_b, "img_b"), n,
                  out.data_ptr(), _lib.stream_ptr(), work=("byte", 12.0 * n * self.h * self.w, "pyr_band_filter_pair"))
        return out

    def synthesize(self, high, phase, amp, table, low, mask, flags, img):
        n = img.shape[0]
        tab = (ctypes.c_int * len(table))(*table) if table is not None else None
        _lib.call("vfi_pyr_synthesize", self._h, high.data_ptr() if torch.is_tensor(high) else None,
                  _ptr_array(phase), _ptr_array(amp) if amp is not None else None, tab,
                  low.data_ptr() if torch.is_tensor(low) else None, mask, flags, img.data_ptr(), n, _lib.stream_ptr(),
                  work=("byte", self._bytes(n, mask, torch.is_tensor(high), torch.is_tensor(low)), "pyr_synthesize"))


class SCFpyr_PyTorch(object):
    def __init__(self, height=5, nbands=4, scale_factor=2, device=None):
        self.height = height
        self.nbands = nbands
        self.scale_factor = scale_factor
        self.device = torch.device("cpu") if device is None else torch.device(device)
        self._plans = {}

    def plan(self, h, w, n):
        key = (h, w)
        p = self._plans.get(key)
        if p is None or p.max_images < n:
            p = Plan(h, w, self.height, self.nbands, self.scale_factor, max(n, 6), self.device)
            self._plans[key] = p
        return p

    def build(self, im_batch):
        if im_batch.dim() != 4 or im_batch.shape[1] != 1:
            raise VfiLibraryError("build expects (N,1,H,W)")
        img = im_batch.squeeze(1).contiguous()
        n, h, w = img.shape
        plan = self.plan(h, w, n)
        nlev, nb = self.height - 2, self.nbands
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=img.device)
        bands = [new(nb, n, *plan.sizes[k], 2) for k in range(nlev)]
        hi, lo = new(n, h, w), new(n, *plan.sizes[nlev])
        plan.analyze(img, hi, bands, None, None, lo, 1.0, (1 << nlev) - 1, BAND_MAJOR | COMPLEX_COEFF)
        return [hi] + [[b[i] for i in range(nb)] for b in bands] + [lo]

    def reconstruct(self, coeff):
        nb = self.nbands
        if nb != len(coeff[1]):
            raise Exception("Unmatched number of orientations")
        hi, lo = coeff[0].contiguous(), coeff[-1].contiguous()
        n, h, w = hi.shape
        plan = self.plan(h, w, n)
        bands = [torch.stack([b.contiguous() for b in level], 0) for level in coeff[1:-1]]   # (nb, N, h, w, 2)
        img = torch.empty((n, h, w), dtype=torch.float32, device=hi.device)
        plan.synthesize(hi, bands, None, None, lo, (1 << len(bands)) - 1, BAND_MAJOR | COMPLEX_COEFF, img)
        return img
