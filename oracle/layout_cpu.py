"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Layout shuffles between the per-image pyramid (`Pyramid.filter`) and PhaseNet's batched form,
level masking for the uncertainty maps, and the coefficient <-> (phase, amplitude) adapters.

  coeff_to_values / values_to_coeff <- reference src/train/pyramid.py:48-78 / :85-112
  separate_vals                     <- reference src/train/utils.py:83-127
  get_concat_layers_inf             <- reference src/train/utils.py:47-80
  get_last_value_levels             <- reference src/train/utils.py:242-280
  get_first_value_levels            <- reference src/train/utils.py:282-320
  subtract_values                   <- reference src/train/utils.py:322-346
  calc_pyr_height                   <- reference src/train/utils.py:168-171

Pinned by tests/golden/layout_helpers.npz (reference functions run on an integer-coded pyramid).
"""
import math

import torch

from .nets_cpu import DecompValues


def calc_pyr_height(h, w):
    return int(math.ceil((math.log2(min(h, w)) - 3) * 2) + 2)


def coeff_to_values(coeff):
    """coeff = [hi (N,H,W), [nbands x (N,h,w,2)] x L, lo (N,hL,wL)]  ->  per-image DecompValues:
    phase/amplitude[level] is (N*nbands, 1, h, w) with index img*nbands + band, finest first."""
    n = coeff[0].shape[0]
    phase, amp = [], []
    for bands in coeff[1:-1]:
        z = torch.stack([torch.view_as_complex(b.contiguous()) for b in bands], 1)   # (N, nb, h, w)
        z = z.reshape(n * len(bands), 1, *z.shape[2:])
        phase.append(torch.atan2(z.imag, z.real))      # == imag(log z), pyramid.py:64
        amp.append(torch.abs(z))                       # pyramid.py:67
    return DecompValues(coeff[0].unsqueeze(1), phase, amp, coeff[-1].unsqueeze(1))


def values_to_coeff(vals, nbands=4):
    n = vals.high_level.shape[0]
    coeff = [vals.high_level.squeeze(1)]
    for p, a in zip(vals.phase, vals.amplitude):
        p = p.reshape(n, nbands, *p.shape[2:])
        a = a.reshape(n, nbands, *a.shape[2:])
        coeff.append([torch.stack((torch.cos(p[:, b]) * a[:, b], torch.sin(p[:, b]) * a[:, b]), -1)
                      for b in range(nbands)])          # pyramid.py:103-107
    coeff.append(vals.low_level.squeeze(1))
    return coeff


def separate_vals(vals, num_input):
    def cut(t, i):
        return t.reshape(num_input, -1, *t.shape[2:])[i].unsqueeze(1)
    return [DecompValues(cut(vals.high_level, i), [cut(p, i) for p in vals.phase],
                         [cut(a, i) for a in vals.amplitude], cut(vals.low_level, i))
            for i in range(num_input)]


def get_concat_layers_inf(vals_list, nbands=4):
    """-> batch = colour channel, channels = [frame0 band0..3, frame1 band0..3], COARSEST first."""
    def per_level(field):
        levels = []
        for k in range(len(getattr(vals_list[0], field))):
            levels.append(torch.cat([getattr(v, field)[k].reshape(-1, nbands, *getattr(v, field)[k].shape[2:])
                                     for v in vals_list], 1))
        return levels[::-1]
    return DecompValues(torch.cat([v.high_level for v in vals_list], 1), per_level("phase"),
                        per_level("amplitude"), torch.cat([v.low_level for v in vals_list], 1))


def get_last_value_levels(vals, use_levels=1):
    """Keep the `use_levels` FINEST band levels and the high residual; zero everything else."""
    keep = lambda lst: [t.clone() if i < use_levels else torch.zeros_like(t) for i, t in enumerate(lst)]
    return DecompValues(vals.high_level.clone(), keep(vals.phase), keep(vals.amplitude),
                        torch.zeros_like(vals.low_level))


def get_first_value_levels(vals, use_levels=1):
    """Keep the `use_levels` COARSEST band levels and the low residual; zero everything else."""
    n = len(vals.phase)
    keep = lambda lst: [t.clone() if i >= n - use_levels else torch.zeros_like(t) for i, t in enumerate(lst)]
    return DecompValues(torch.zeros_like(vals.high_level), keep(vals.phase), keep(vals.amplitude),
                        vals.low_level.clone())


def subtract_values(v1, v2):
    return DecompValues((v1.high_level - v2.high_level).abs(),
                        [(a - b).abs() for a, b in zip(v1.phase, v2.phase)],
                        [(a - b).abs() for a, b in zip(v1.amplitude, v2.amplitude)],
                        (v1.low_level - v2.low_level).abs())
