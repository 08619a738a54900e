"""Layout helpers -- mirror of reference src/train/utils.py (same names and argument meaning).

Reshapes are torch views; every copy / arithmetic step is a libvfi_hip.so kernel.  The level-masking
helpers return the scalar 0 for zeroed parts (the convention the reference itself uses for missing
levels, src/phase_net/phase_net.py:91-93) instead of allocating zero tensors; `Pyramid.inv_filter`
skips those levels' transforms.  This package's own per-frame driver bypasses most of this file through
`Pyramid.filter(concat_frames=...)`; it is kept so reference-style call sequences run unchanged.
"""
import math

import numpy as np
import torch

from .. import ops
from ..values import DecompValues
from .transform import *  # noqa: F401,F403  (the reference re-exports these: utils.py:8)


def _cat_channels(parts):
    n, _, h, w = parts[0].shape
    out = ops.new((n, sum(p.shape[1] for p in parts), h, w), parts[0])
    c0 = 0
    for p in parts:
        ops.affine_slice(p.contiguous(), out[:, c0:c0 + p.shape[1]])
        c0 += p.shape[1]
    return out


def get_concat_layers_inf(pyr, vals_list):
    """utils.py:47-80: -> batch = colour, channels [f0 b0..b3, f1 b0..b3], lists COARSEST first."""
    nb = pyr.nbands
    split = lambda t: t.reshape(t.shape[0] // nb, nb, t.shape[2], t.shape[3])
    nlev = pyr.height - 2
    phase = [_cat_channels([split(v.phase[k]) for v in vals_list]) for k in range(nlev)]
    amp = [_cat_channels([split(v.amplitude[k]) for v in vals_list]) for k in range(nlev)]
    return DecompValues(_cat_channels([v.high_level for v in vals_list]), phase[::-1], amp[::-1],
                        _cat_channels([v.low_level for v in vals_list]))


def get_concat_layers(pyr, vals1, vals2):
    """utils.py:19-44 (two-value form of the above)."""
    return get_concat_layers_inf(pyr, [vals1, vals2])


def separate_vals(vals, num_input):
    """utils.py:83-127: split the batched pyramid into per-frame values (views)."""
    def cut(t, i):
        return t.reshape(num_input, -1, t.shape[2], t.shape[3])[i].unsqueeze(1) if torch.is_tensor(t) else t
    return [DecompValues(cut(vals.high_level, i), [cut(p, i) for p in vals.phase],
                         [cut(a, i) for a in vals.amplitude], cut(vals.low_level, i)) for i in range(num_input)]


def calc_pyr_height(img):
    """utils.py:168-171."""
    size = img.shape[1:]
    return int(np.ceil((np.log2(min(size)) - 3) * 2) + 2)


def pad_img(img):
    """utils.py:155-165: zero-pad an (H,W,C) numpy image to the next sqrt(2)-power square."""
    size = np.array(img.shape[:2])
    pow2 = (2 ** (np.ceil(np.log2(size) * 2) / 2)).astype(int)
    pad = (max(pow2) - size).astype(int)
    return np.pad(img, [(0, pad[0]), (0, pad[1]), (0, 0)], mode="constant")


def get_last_value_levels(vals, use_levels=1):
    """utils.py:242-280: keep the `use_levels` FINEST band levels + high; zero the rest (as scalar 0)."""
    keep = lambda lst: [t if i < use_levels else 0 for i, t in enumerate(lst)]
    return DecompValues(vals.high_level, keep(vals.phase), keep(vals.amplitude), 0)


def get_first_value_levels(vals, use_levels=1):
    """utils.py:282-320: keep the `use_levels` COARSEST band levels + low; zero the rest (as scalar 0)."""
    n = len(vals.phase)
    keep = lambda lst: [t if i >= n - use_levels else 0 for i, t in enumerate(lst)]
    return DecompValues(0, keep(vals.phase), keep(vals.amplitude), vals.low_level)


def subtract_values(vals1, vals2):
    """utils.py:322-346: field-wise |a - b|."""
    d = lambda a, b: ops.absdiff(a, b) if torch.is_tensor(a) and torch.is_tensor(b) else 0
    return DecompValues(d(vals1.high_level, vals2.high_level), [d(a, b) for a, b in zip(vals1.phase, vals2.phase)],
                        [d(a, b) for a, b in zip(vals1.amplitude, vals2.amplitude)], d(vals1.low_level, vals2.low_level))


def combine_values(vals_list):
    """utils.py:208-240: concatenate values along the batch axis."""
    cat = lambda ts: torch.cat(ts, 0)
    return DecompValues(cat([v.high_level for v in vals_list]),
                        [cat([v.phase[i] for v in vals_list]) for i in range(len(vals_list[0].phase))],
                        [cat([v.amplitude[i] for v in vals_list]) for i in range(len(vals_list[0].phase))],
                        cat([v.low_level for v in vals_list]))


def exchange_vals(val_base, val_changer, start, end):
    """utils.py:145-152."""
    for level in range(start, end):
        val_base.phase[level] = val_changer.phase[level]
        val_base.amplitude[level] = val_changer.amplitude[level]
    return val_base


def preprocess(img, device, normalized=True):
    """utils.py:174-206: (B,C,H,W) rgb -> (B*C,H,W) Lab on `device`."""
    img = torch.as_tensor(img)
    if img.dim() != 4:
        print("Image shape has to be (B, C, H, W)!")
        return img
    h, w = img.shape[2:4]
    if not normalized:
        img = img / 255
    return rgb2lab(img.float().to(device)).reshape((-1, h, w))  # noqa: F405
