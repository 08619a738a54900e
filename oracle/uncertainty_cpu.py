"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The two pyramid-derived uncertainty maps of the fused path, reference
src/fusion_net/interpolate_twoframe.py:198-225, using scipy exactly as the reference does
(scipy.ndimage.gaussian_filter(h, 5), scipy.ndimage.median_filter(f, size=50))."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter, median_filter

from . import layout_cpu


def phase_uncertainty_tail(h_freq, h_freq_ph):
    """interpolate_twoframe.py:210-214.  inputs (3,H,W) reconstructions -> (1,H,W)."""
    d = (h_freq.mean(0, keepdim=True) - h_freq_ph.mean(0, keepdim=True)).abs()
    d = (d * 100).clamp(min=0, max=1.0)
    return torch.stack([torch.as_tensor(gaussian_filter(h.numpy(), 5)) for h in d])


def ada_uncertainty_tail(freq_img):
    """interpolate_twoframe.py:219-224.  input (3,H,W) reconstruction -> (1,H,W)."""
    f = freq_img.mean(0, keepdim=True) * 30
    med = torch.stack([torch.as_tensor(median_filter(x.numpy(), size=50)) for x in f])
    return ((f - med).abs() * 5).clamp(0, 1)


def uncertainty_maps(pyr, ada_pred, rgb_pred):
    """interpolate_twoframe.py:198-225 with an oracle Pyramid.  ada_pred, rgb_pred (3,H,W) ->
    (phase_uncertainty, ada_uncertainty), each (1,H,W)."""
    vals = pyr.filter(torch.cat((ada_pred, rgb_pred), 0).float())
    vals_ada, vals_ph = layout_cpu.separate_vals(vals, 2)
    h_freq = pyr.inv_filter(layout_cpu.get_last_value_levels(vals_ada, 1))
    h_freq_ph = pyr.inv_filter(layout_cpu.get_last_value_levels(vals_ph, 1))
    phase_unc = phase_uncertainty_tail(h_freq, h_freq_ph)
    diff = layout_cpu.get_first_value_levels(layout_cpu.subtract_values(vals_ph, vals_ada), 6)
    ada_unc = ada_uncertainty_tail(pyr.inv_filter(diff))
    return phase_unc, ada_unc
