"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).          *** PARITY UNPINNED ***

rgb <-> Lab as the reference uses it: reference src/train/transform.py:17-25 (rgb2lab_single) and
:40-49 (lab2rgb_single), which call skimage.color.rgb2lab / lab2rgb (illuminant D65, observer 2) and
scale L/100, (a+128)/255, (b+128)/255.  skimage is absent from this image and un-pinned in the
reference, so the conversion is restated from skimage's published formulas (float64 numpy, cast to
float32 at the end); pinned only by self-consistency and known colour values (tests/test_oracle_misc.py).
"""
import numpy as np
import torch

XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                         [0.212671, 0.715160, 0.072169],
                         [0.019334, 0.119193, 0.950227]])
RGB_FROM_XYZ = np.linalg.inv(XYZ_FROM_RGB)
WHITE_D65_2 = np.array([0.95047, 1.0, 1.08883])


def rgb2lab_single(img):
    """(3,H,W) rgb in [0,1] -> (3,H,W) scaled Lab (transform.py:17-25)."""
    arr = img.detach().cpu().permute(1, 2, 0).numpy().astype(np.float64)
    mask = arr > 0.04045
    arr = np.where(mask, np.power((arr + 0.055) / 1.055, 2.4), arr / 12.92)
    xyz = arr @ XYZ_FROM_RGB.T / WHITE_D65_2
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = 116.0 * f[..., 1] - 16.0
    a = 500.0 * (f[..., 0] - f[..., 1])
    b = 200.0 * (f[..., 1] - f[..., 2])
    lab = np.stack([L / 100.0, (a + 128.0) / 255.0, (b + 128.0) / 255.0], -1)
    return torch.from_numpy(lab.astype(np.float32)).permute(2, 0, 1).contiguous()


def lab2rgb_single(img):
    """(3,H,W) scaled Lab -> (3,H,W) rgb clipped to [0,1] (transform.py:40-49)."""
    lab = img.detach().cpu().permute(1, 2, 0).numpy().astype(np.float64)
    L, a, b = lab[..., 0] * 100.0, lab[..., 1] * 255.0 - 128.0, lab[..., 2] * 255.0 - 128.0
    fy = (L + 16.0) / 116.0
    fx = a / 500.0 + fy
    fz = np.maximum(fy - b / 200.0, 0.0)
    f = np.stack([fx, fy, fz], -1)
    xyz = np.where(f > 0.2068966, f ** 3, (f - 16.0 / 116.0) / 7.787) * WHITE_D65_2
    rgb = xyz @ RGB_FROM_XYZ.T
    mask = rgb > 0.0031308
    rgb = np.where(mask, 1.055 * np.power(np.maximum(rgb, 1e-30), 1 / 2.4) - 0.055, rgb * 12.92)
    return torch.from_numpy(np.clip(rgb, 0, 1).astype(np.float32)).permute(2, 0, 1).contiguous()
