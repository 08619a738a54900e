#!/usr/bin/env python3
"""Functional pin of the SELF-DEFINED steerable-pyramid spec against the reference's TRAINED PhaseNet.

The pyramid arithmetic is not in the reference (third-party `steerable`, fork unknown: DESIGN.md section 2), so no
fixture can pin it.  What the reference does ship is `src/phase_net/phase_net.pt`, fit to that unknown pyramid's
outputs (training: 256x256 crops, height 12, nbands 4, scale sqrt(2): src/train/train.py:73-83).  If the oracle's spec
(level geometry, band order, phase sign, rotation) matches what the network was trained on, PhaseNet-only
interpolation (reference flow src/phase_net/interpolate_twoframe.py:38-112: Lab -> filter -> normalize_vals ->
PhaseNet -> inv_filter -> rgb) of a translating pattern must beat plain frame averaging, and a spec that differs in
a way the network can see must do worse.  This script measures exactly that, on seeded synthetic triplets whose
middle frame is analytic, for

  * the plausible level-size rules of a sqrt(2) fork:      ceil (ours) / floor / round,
  * controls that keep perfect reconstruction but change what the network sees:
        band order reversed, phase sign flipped, bands rotated by +90 / -90 / 180 degrees (other conventions for
        the (-i)^(nbands-1) factor), and -- not reconstruction preserving -- DC of the cropped windows at (h-1)//2.

Runs in the BUILD CONTAINER only (needs /root/reference/src/phase_net/phase_net.pt, loaded as data with
map_location='cpu'; no reference code is imported).  Writes tests/golden/pyramid_spec_pin.json; the table is copied
into DESIGN.md.  tests/test_oracle_pyramid_pin.py re-checks a reduced case whenever the checkpoint is present.
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
from oracle import color_cpu, layout_cpu, nets_cpu, pyramid_cpu, synth  # noqa: E402
from oracle.nets_cpu import DecompValues  # noqa: E402

CKPT = "/root/reference/src/phase_net/phase_net.pt"


def psnr(a, b):
    return 10 * math.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-30))


def wrap(p):
    return torch.remainder(p + math.pi, 2 * math.pi) - math.pi


class View:
    """A consistent re-labelling of the band values between the pyramid and the network (controls)."""

    def __init__(self, kind):
        self.kind = kind

    def _bands(self, t, fn):
        n4 = t.shape[0]
        return fn(t.reshape(n4 // 4, 4, *t.shape[1:])).reshape(t.shape)

    def fwd(self, vals):
        k = self.kind
        if k == "plain":
            return vals
        if k == "band_order_reversed":
            f = lambda lst: [self._bands(t, lambda x: x.flip(1)) for t in lst]
            return DecompValues(vals.high_level, f(vals.phase), f(vals.amplitude), vals.low_level)
        if k == "phase_sign_flipped":
            return DecompValues(vals.high_level, [-p for p in vals.phase], vals.amplitude, vals.low_level)
        if k.startswith("rotated_"):
            a = math.radians(float(k.split("_")[1]))
            return DecompValues(vals.high_level, [wrap(p + a) for p in vals.phase], vals.amplitude, vals.low_level)
        raise ValueError(k)

    def inv(self, vals):
        k = self.kind
        if k == "plain":
            return vals
        if k == "band_order_reversed":
            f = lambda lst: [self._bands(t, lambda x: x.flip(1)) for t in lst]
            return DecompValues(vals.high_level, f(vals.phase), f(vals.amplitude), vals.low_level)
        if k == "phase_sign_flipped":
            return DecompValues(vals.high_level, [-p for p in vals.phase], vals.amplitude, vals.low_level)
        if k.startswith("rotated_"):
            a = math.radians(float(k.split("_")[1]))
            return DecompValues(vals.high_level, [p - a for p in vals.phase], vals.amplitude, vals.low_level)
        raise ValueError(k)


@torch.no_grad()
def phasenet_only(sd, rgb0, rgb2, variant=None, view="plain"):
    """reference src/phase_net/interpolate_twoframe.py:38-112 on the CPU oracle (no padding needed at sqrt2-power
    squares); the three Lab channels go through as one batch, as the fused path does (interpolate_twoframe.py:168-188)."""
    h, w = rgb0.shape[1:]
    height = layout_cpu.calc_pyr_height(h, w)
    pyr = pyramid_cpu.Pyramid(height, 4, np.sqrt(2), **(variant or {}))
    v = View(view)
    lab = torch.cat((color_cpu.rgb2lab_single(rgb0), color_cpu.rgb2lab_single(rgb2)), 0).float()
    vals = v.fwd(pyr.filter(lab))
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(vals, 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    pred = nets_cpu.phasenet_forward(sd, normed, state, height)
    lab_pred = pyr.inv_filter(v.inv(pred))
    return color_cpu.lab2rgb_single(lab_pred)


CASES = [("ours: ceil sizes, DC at h//2", {}, "plain"),
         ("floor sizes", {"size_rule": "floor"}, "plain"),
         ("round sizes", {"size_rule": "round"}, "plain"),
         ("control: DC at (h-1)//2", {"dc_rule": "low"}, "plain"),
         ("control: band order reversed", {}, "band_order_reversed"),
         ("control: phase sign flipped", {}, "phase_sign_flipped"),
         ("control: bands rotated +90 deg", {}, "rotated_90"),
         ("control: bands rotated -90 deg", {}, "rotated_-90"),
         ("control: bands rotated 180 deg", {}, "rotated_180")]
SHIFTS = [(1.0, 0.5), (2.0, -1.5), (3.5, -2.25), (5.0, 3.0)]


def main(size=256, seeds=(0, 1, 2), out=os.path.join(ROOT, "tests", "golden", "pyramid_spec_pin.json")):
    sd = torch.load(CKPT, map_location="cpu")
    torch.set_num_threads(8)
    rows = []
    for name, variant, view in CASES:
        per_shift = []
        for shift in SHIFTS:
            d_pn, d_avg = [], []
            for seed in seeds:
                f0, f1, f2 = (torch.from_numpy(x) for x in synth.translating_pair(seed, size, size, shift=shift))
                pred = phasenet_only(sd, f0, f2, variant, view)
                d_pn.append(psnr(pred, f1))
                d_avg.append(psnr((f0 + f2) / 2, f1))
            per_shift.append({"shift": shift, "psnr_phasenet": float(np.mean(d_pn)), "psnr_average": float(np.mean(d_avg))})
        mean_pn = float(np.mean([r["psnr_phasenet"] for r in per_shift]))
        mean_avg = float(np.mean([r["psnr_average"] for r in per_shift]))
        rows.append({"variant": name, "mean_psnr_phasenet": mean_pn, "mean_psnr_average": mean_avg, "per_shift": per_shift})
        print(f"{name:36s} PhaseNet-only {mean_pn:6.2f} dB   frame average {mean_avg:6.2f} dB   " +
              "  ".join(f"{r['shift']}: {r['psnr_phasenet']:.2f}" for r in per_shift), flush=True)
    with open(out, "w") as f:
        json.dump({"size": size, "seeds": list(seeds), "checkpoint": "reference src/phase_net/phase_net.pt", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
