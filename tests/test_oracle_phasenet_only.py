"""BASELINE.json configs[0]: PhaseNet-only interpolation of one 256x256 triplet on the CPU path (plumbing, no GPU).

The fixture tests/golden/phasenet_only_256.npz was produced by the REFERENCE's own Pyramid adapter,
get_concat_layers and PhaseNet classes following src/phase_net/interpolate_twoframe.py:62-104 (one colour channel at a
time), with the absent third-party transform served by oracle/pyramid_cpu.py (tests/golden/make_golden.py,
gen_phasenet_only).  Here the oracle's restatements of those three pieces (layout_cpu, nets_cpu) run the same flow in
the batched form the fused path uses and must reproduce it."""
import math
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import color_cpu, layout_cpu, nets_cpu, pyramid_cpu, synth


def test_phasenet_only_256_matches_reference_flow():
    g = np.load(os.path.join(GOLDEN, "phasenet_only_256.npz"))
    h, w, height = int(g["h"]), int(g["w"]), int(g["height"])
    assert (h, w, height) == (256, 256, 12) and layout_cpu.calc_pyr_height(h, w) == height
    f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(int(g["pair_seed"]), h, w))
    lab = torch.cat((color_cpu.rgb2lab_single(f0), color_cpu.rgb2lab_single(f2)), 0).float()
    pyr = pyramid_cpu.Pyramid(height, 4, np.sqrt(2))
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(pyr.filter(lab), 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    with torch.no_grad():
        pred = nets_cpu.phasenet_forward(nets_cpu.phasenet_random_state_dict(int(g["weight_seed"])), normed, state, height)
    lab_pred = pyr.inv_filter(pred)
    ref = torch.from_numpy(g["lab_pred"])
    assert lab_pred.shape == ref.shape == (3, h, w)
    err = (lab_pred - ref).abs().max().item()
    assert err <= 2e-5, err
    # the flow ends in an rgb frame in [0,1] (interpolate_twoframe.py:108-112)
    rgb = color_cpu.lab2rgb_single(lab_pred)
    assert rgb.min().item() >= 0.0 and rgb.max().item() <= 1.0 and math.isfinite(float(rgb.sum()))
