"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of ONE fused interpolation, reference src/fusion_net/interpolate_twoframe.py:148-330
(`interp` minus file I/O), built from the oracle pieces: color_cpu (Lab), nets_cpu (AdaCoF / PhaseNet /
FusionNet), pyramid_cpu (steerable pyramid, spec unpinned), layout_cpu, uncertainty_cpu (scipy).
Also the `cpu_baseline` leg of bench.py times this function.
"""
import itertools
import time

import numpy as np
import torch

from . import color_cpu, layout_cpu, nets_cpu, pyramid_cpu, uncertainty_cpu
from .nets_cpu import DecompValues


def seeded_weights(seed=0):
    return {"adacof": nets_cpu.adacofnet_random_state_dict(seed + 3),
            "phasenet": nets_cpu.phasenet_random_state_dict(seed + 1),
            "fusionnet": nets_cpu.fusionnet_random_state_dict(seed + 2)}


@torch.no_grad()
def interp(rgb1, rgb2, weights, output_baseline=False, timings=None, hooks=None, reuse=None, stop_after=None,
           keep_stages=False):
    """rgb1, rgb2: (3,H,W) float32 in [0,1] -> dict of tensors (same keys as the product's FusionInterpolator).

    Test aids (tests/test_pipeline_gpu.py, the explained-set check of `ada_uncertainty`): `hooks` maps a stage name
    ("vals_input": PhaseNet's input pyramid, :172-175; "vals_second": the analysis of (ada_pred, rgb_pred), :198-203) to a
    function value -> value applied before the stage is consumed; `reuse` = the dict an earlier call returned, whose
    AdaCoF #1 outputs and first analysis are taken over instead of recomputed; `stop_after="ada_uncertainty"` returns
    once both maps exist; `keep_stages` adds the hooked values as the stages consumed them ("_vals_input", "_vals_second")
    and PhaseNet's input before its hook ("_vals_input_raw", what `reuse` needs) to the returned dict."""
    tic = time.perf_counter()

    def lap(name):
        nonlocal tic
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - tic
            tic = now

    h, w = rgb1.shape[1:]
    height = layout_cpu.calc_pyr_height(h, w)                                      # :124-129
    pyr = pyramid_cpu.Pyramid(height, 4, np.sqrt(2))
    adacof = lambda a, b: nets_cpu.adacofnet_forward(weights["adacof"], a, b)
    lab1, lab2 = color_cpu.rgb2lab_single(rgb1), color_cpu.rgb2lab_single(rgb2)   # :148-149
    f1, f2 = rgb1.unsqueeze(0), rgb2.unsqueeze(0)
    lap("lab")
    hooks = hooks or {}
    if reuse is not None:
        ada_pred, flow_var_map, vals_input = reuse["ada_pred"][0], reuse["flow_var_map"].squeeze(1), reuse["_vals_input_raw"]
    else:
        _, _, ada_pred, flow_var_map = adacof(f1, f2)                              # :156
        ada_pred = ada_pred[0]
        flow_var_map = flow_var_map.squeeze(1)
        lap("adacof")
        vals_batch = pyr.filter(torch.cat((lab1, lab2), 0).float())                # :172
        vals_input = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(vals_batch, 2))
        lap("pyramid")
    vals_input_raw = vals_input
    if "vals_input" in hooks:
        vals_input = hooks["vals_input"](vals_input)
    normed, state = nets_cpu.phasenet_normalize(vals_input)                        # :175
    vals_pred = nets_cpu.phasenet_forward(weights["phasenet"], normed, state, height)   # :185
    lap("phasenet")
    lab_pred = pyr.inv_filter(vals_pred)                                           # :188
    lap("pyramid")
    rgb_pred = color_cpu.lab2rgb_single(lab_pred)                                  # :192
    phase_pred = rgb_pred.clone()
    lap("lab")
    vals = pyr.filter(torch.cat((ada_pred, rgb_pred), 0).float())                  # :198-203
    if "vals_second" in hooks:
        vals = hooks["vals_second"](vals)
    vals_ada, vals_ph = layout_cpu.separate_vals(vals, 2)
    h_freq = pyr.inv_filter(layout_cpu.get_last_value_levels(vals_ada, 1))
    h_freq_ph = pyr.inv_filter(layout_cpu.get_last_value_levels(vals_ph, 1))
    diff = layout_cpu.get_first_value_levels(layout_cpu.subtract_values(vals_ph, vals_ada), 6)
    freq = pyr.inv_filter(diff)
    lap("pyramid")
    phase_uncertainty = uncertainty_cpu.phase_uncertainty_tail(h_freq, h_freq_ph)  # :207-214
    lap("gaussian")
    ada_uncertainty = uncertainty_cpu.ada_uncertainty_tail(freq)                   # :217-225
    lap("median")
    pp = phase_pred.unsqueeze(0)
    if keep_stages:
        extra = {"_vals_input_raw": vals_input_raw, "_vals_input": vals_input, "_vals_second": vals}
    else:
        extra = {}
    if stop_after == "ada_uncertainty":
        return dict(extra, phase_pred=pp, ada_pred=ada_pred.unsqueeze(0), flow_var_map=flow_var_map.unsqueeze(1),
                    phase_uncertainty=phase_uncertainty, ada_uncertainty=ada_uncertainty)
    _, _, b1, _ = adacof(f1, pp)                                                   # :229-238
    _, _, b2, _ = adacof(pp, f2)
    _, _, base, _ = adacof(b1, b2)
    lap("adacof")
    out = {"phase_pred": pp, "ada_pred": ada_pred.unsqueeze(0), "base": base, "flow_var_map": flow_var_map.unsqueeze(1),
           "phase_uncertainty": phase_uncertainty, "ada_uncertainty": ada_uncertainty}
    if output_baseline:                                                            # :288-322
        va = pyr.filter(color_cpu.rgb2lab_single(ada_pred).float())
        vp = pyr.filter(color_cpu.rgb2lab_single(phase_pred).float())
        split = len(va.phase) // 2
        mix = DecompValues(va.high_level, list(itertools.chain(vp.phase[:split], va.phase[split:])),
                           list(itertools.chain(vp.amplitude[:split], va.amplitude[split:])), vp.low_level)
        out["baseline"] = color_cpu.lab2rgb_single(pyr.inv_filter(mix)).unsqueeze(0)
        lap("pyramid")
    other = torch.cat([lab1, lab2], 0).unsqueeze(0).float()                        # :324-327
    maps = torch.stack([ada_uncertainty, phase_uncertainty, flow_var_map], 1).float()
    out["final"] = nets_cpu.fusionnet_forward(weights["fusionnet"], base, out["ada_pred"], pp, other, maps, 0)   # :330
    lap("fusionnet")
    out.update(extra)
    return out
