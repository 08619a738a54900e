"""Quality scoring on the GPU -- counterpart of reference src/evaluation/evaluate_image.py:7-30.

`evaluate_image(args, prediction_im, target_im)` returns the reference's 7-vector
[ssim, lpips, psnr, ssd, l1, mse, variance] for (3,H,W) images in [0,1], centre-cropped to `args.dim`:
  * psnr  -- piq.psnr(data_range=1): 10 log10(1 / (mean((x-y)^2) + 1e-8));
  * ssd, l1, mse, variance -- exactly the reference's expressions (NB: its `l1` and `mse` are the SIGNED sum / mean
    of the difference, evaluate_image.py:25-26);
  * ssim  -- piq.ssim defaults (11-tap Gaussian sigma 1.5, valid region, k1=0.01, k2=0.03, average-pool downsample by
    max(1, round(min(H,W)/256))): restated from piq's published formula (piq is absent here: PARITY UNPINNED);
  * lpips -- needs piq's pretrained VGG16 + linear-head weights, which exist neither in the reference tree nor in
    this image (no network): the column is NaN, and vfi_amd.evaluation.evaluate aggregates NaN-aware.
All reductions are deterministic (fixed-order double accumulation in libvfi_hip.so)."""
import math

import numpy as np
import torch

from .. import _lib, ops
from ..fusion_net.interpolate_twoframe import crop_center


def _sums(a, b):
    out = torch.empty(2, dtype=torch.float64, device=a.device)
    ws = torch.empty(2048, dtype=torch.float64, device=a.device)
    _lib.call("vfi_diff_sums", _lib.dptr(a.contiguous(), "a"), _lib.dptr(b.contiguous(), "b"), a.numel(), out.data_ptr(),
              ws.data_ptr(), _lib.stream_ptr())
    return out.cpu().numpy()


def psnr(x, y, data_range=1.0):
    s1, s2 = _sums(x, y)
    return 10.0 * math.log10(data_range ** 2 / (s2 / x.numel() + 1e-8))


def ssim(x, y, kernel_size=11, kernel_sigma=1.5, k1=0.01, k2=0.03, downsample=True):
    """x, y: (C,H,W) in [0,1] on the device."""
    c, h, w = x.shape
    f = max(1, round(min(h, w) / 256))
    if f > 1 and downsample:                                   # piq: F.avg_pool2d(kernel_size=f)
        def pool(t):
            t = t.contiguous()
            o = torch.empty((c, h // f, w // f), dtype=torch.float32, device=t.device)
            _lib.call("vfi_avg_pool", _lib.dptr(t, "x"), o.data_ptr(), c, h, w, f, _lib.stream_ptr())
            return o
        x, y = pool(x), pool(y)
        c, h, w = x.shape
    x, y = x.contiguous(), y.contiguous()
    n = x.numel()
    prod = lambda a, b: (lambda o: (_lib.call("vfi_mul", a.data_ptr(), b.data_ptr(), o.data_ptr(), n, _lib.stream_ptr()), o)[1])(torch.empty_like(a))
    r = kernel_size // 2
    g = lambda t: ops.gaussian_filter(t, kernel_sigma, truncate=(r + 0.25) / kernel_sigma)   # radius r taps, same weights
    mx, my, exx, eyy, exy = g(x), g(y), g(prod(x, x)), g(prod(y, y)), g(prod(x, y))
    out = torch.empty(2, dtype=torch.float64, device=x.device)
    ws = torch.empty(2048, dtype=torch.float64, device=x.device)
    _lib.call("vfi_ssim_sum", mx.data_ptr(), my.data_ptr(), exx.data_ptr(), eyy.data_ptr(), exy.data_ptr(), c, h, w, r,
              float(k1 ** 2), float(k2 ** 2), out.data_ptr(), ws.data_ptr(), _lib.stream_ptr())
    return float(out[0].item()) / (c * (h - 2 * r) * (w - 2 * r))


def evaluate_image(args, prediction_im, target_im):
    crop = lambda t: crop_center(t.permute(1, 2, 0), args.dim, args.dim).permute(2, 0, 1).contiguous()
    p, t = crop(prediction_im.float()), crop(target_im.float())
    s1, s2 = _sums(p, t)
    n = p.numel()
    mean = s1 / n
    var = (s2 - n * mean * mean) / (n - 1)                      # torch.var: unbiased
    return np.array([ssim(p, t), float("nan"), 10.0 * math.log10(1.0 / (s2 / n + 1e-8)), math.sqrt(s2), s1, mean, var])
