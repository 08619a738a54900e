"""Oracle restatements of PhaseNet / KernelEstimation / AdaCoFNet glue / FusionNet / layout helpers
against fixtures produced by the reference's own classes (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import layout_cpu, nets_cpu, synth
from oracle.nets_cpu import DecompValues

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def _vals(g, prefix):
    n = 0
    while f"{prefix}phase{n}" in g:
        n += 1
    t = lambda k: torch.from_numpy(g[prefix + k])
    return DecompValues(t("high"), [t(f"phase{k}") for k in range(n)], [t(f"amp{k}") for k in range(n)], t("low"))


def _assert_vals(a, b, tol):
    assert len(a.phase) == len(b.phase)
    for x, y in [(a.high_level, b.high_level), (a.low_level, b.low_level)] + list(zip(a.phase, b.phase)) + \
            list(zip(a.amplitude, b.amplitude)):
        assert x.shape == y.shape
        assert (x - y).abs().max().item() <= tol


def test_layout_helpers_exact():
    g = _load("layout_helpers")
    vals = _vals(g, "vals_")
    sep = layout_cpu.separate_vals(vals, 2)
    _assert_vals(sep[0], _vals(g, "sep0_"), 0.0)
    _assert_vals(sep[1], _vals(g, "sep1_"), 0.0)
    _assert_vals(layout_cpu.get_concat_layers_inf(sep), _vals(g, "cat_"), 0.0)
    _assert_vals(layout_cpu.get_last_value_levels(sep[0], 1), _vals(g, "last_"), 0.0)
    _assert_vals(layout_cpu.get_first_value_levels(sep[1], 2), _vals(g, "first_"), 0.0)
    _assert_vals(layout_cpu.subtract_values(sep[0], sep[1]), _vals(g, "sub_"), 0.0)


def test_calc_pyr_height():
    # reference src/train/utils.py:168-171; values quoted in SURVEY.md section 8
    assert [layout_cpu.calc_pyr_height(*s) for s in [(256, 256), (512, 512), (720, 1280), (1080, 1920)]] == \
        [12, 14, 15, 17]


def _run_phasenet(g):
    h, w, height = int(g["h"]), int(g["w"]), int(g["height"])
    sd = nets_cpu.phasenet_random_state_dict(int(g["weight_seed"]))
    batch = synth.synthetic_vals(int(g["input_seed"]), 6, h, w, height)
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(batch, 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    with torch.no_grad():
        out = nets_cpu.phasenet_forward(sd, normed, state, height)
    return normed, state, out


@pytest.mark.parametrize("tag", ["32x48", "96x112"])
def test_phasenet_matches_reference(tag):
    g = _load("phasenet_" + tag)
    step = int(g["step"])
    normed, (maxes, max_low), out = _run_phasenet(g)
    np.testing.assert_allclose(max_low.numpy(), g["max_low"], rtol=1e-6)
    for k, mx in enumerate(maxes):
        np.testing.assert_allclose(mx.numpy(), g[f"max_amp{k}"], rtol=1e-6)
    if step == 1:
        _assert_vals(normed, _vals(g, "norm_"), 1e-6)
    items = [("high", out.high_level), ("low", out.low_level)]
    items += [(f"phase{k}", p) for k, p in enumerate(out.phase)] + [(f"amp{k}", a) for k, a in enumerate(out.amplitude)]
    for name, t in items:
        ref = g["out_" + name]
        got = t[..., ::step, ::step].numpy()
        assert got.shape == ref.shape, name
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 2e-5 * scale, name
        np.testing.assert_allclose(t.double().sum(dim=(1, 2, 3)).numpy(), g["out_" + name + "_sum"],
                                   rtol=1e-4, atol=1e-2 * scale)
    # 96x112 has 8 band levels, so the last block (layers.7) is re-used (phase_net.py:148)
    if tag == "96x112":
        assert len(out.phase) >= 7


@pytest.mark.parametrize("tag", ["64x64", "40x72"])
def test_fusionnet_matches_reference(tag):
    g = _load("fusionnet_" + tag)
    sd = nets_cpu.fusionnet_random_state_dict(int(g["seed"]))
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        for variant in (0, 1):
            out = nets_cpu.fusionnet_forward(sd, t("base"), t("adacof"), t("phase"), t("other"), t("maps"), variant)
            assert np.abs(out.numpy() - g[f"out_variant{variant}"]).max() <= 1e-6


@pytest.mark.parametrize("tag", ["64x96", "40x50"])
def test_adacofnet_matches_reference(tag):
    g = _load("adacofnet_" + tag)
    sd = nets_cpu.adacofnet_random_state_dict(int(g["seed"]))
    f0, f2 = torch.from_numpy(g["frame0"]), torch.from_numpy(g["frame2"])
    with torch.no_grad():
        t1, t2, fr, mask = nets_cpu.adacofnet_forward(sd, f0, f2, faithful_crop_bug=True)
        np.testing.assert_allclose(t1.numpy(), g["t1"], atol=2e-6)
        np.testing.assert_allclose(t2.numpy(), g["t2"], atol=2e-6)
        np.testing.assert_allclose(fr.numpy(), g["frame1"], atol=2e-6)
        np.testing.assert_allclose(mask.numpy(), g["mask"], atol=2e-6)
        if "head_w1" in g:
            heads = nets_cpu.kernel_estimation(sd, nets_cpu.module_normalize(f0), nets_cpu.module_normalize(f2))
            for name, h in zip(("w1", "a1", "b1", "w2", "a2", "b2", "occ"), heads):
                np.testing.assert_allclose(h[..., ::4, ::4].numpy(), g["head_" + name], atol=2e-6)
                np.testing.assert_allclose(h.double().sum(dim=(2, 3)).numpy(), g["head_" + name + "_sum"], rtol=1e-5)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src/phase_net/phase_net.pt")),
                    reason="trained reference checkpoints only exist in the build container")
def test_trained_checkpoints_load_into_oracle_key_layout():
    # The reference's own trained checkpoints use exactly the keys/shapes the oracle (and the
    # product modules) expect.  Read as data only.
    sd = torch.load(os.path.join(REF, "src/phase_net/phase_net.pt"), map_location="cpu")
    want = dict(nets_cpu.phasenet_shapes())
    assert set(sd) == set(want)
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    sd = torch.load(os.path.join(REF, "src/fusion_net/fusion_net.pt"), map_location="cpu")
    want = dict(nets_cpu.fusionnet_shapes())
    assert set(sd) == set(want)
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src/phase_net/phase_net.pt")),
                    reason="the reference (and its trained checkpoints) only exist in the build container")
def test_oracle_matches_reference_classes_with_trained_weights():
    # Build-container only: the reference's OWN classes with its OWN trained checkpoints (read as data, never
    # committed) against the oracle restatements on seeded inputs -- realistic BatchNorm statistics / weight scales.
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg._placeholders()
    from src.fusion_net.fusion_net import FusionNet as RefFusion
    from src.phase_net.phase_net import PhaseNet as RefPhaseNet
    from src.train import utils as rutils
    from src.train.pyramid import DecompValues as RefVals
    # PhaseNet
    sd = torch.load(os.path.join(REF, "src/phase_net/phase_net.pt"), map_location="cpu")
    h, w = 48, 64
    height = layout_cpu.calc_pyr_height(h, w)
    pyr = types.SimpleNamespace(height=height, nbands=4)
    ref = RefPhaseNet(pyr, torch.device("cpu"), num_img=2)
    ref.load_state_dict(sd)
    ref.eval()
    batch = synth.synthetic_vals(11, 6, h, w, height)
    vin_ref = rutils.get_concat_layers_inf(pyr, rutils.separate_vals(RefVals(*batch), 2))
    with torch.no_grad():
        want = ref(ref.normalize_vals(vin_ref))
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(batch, 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    with torch.no_grad():
        got = nets_cpu.phasenet_forward(sd, normed, state, height)
    for a, b in zip(got.phase + got.amplitude + [got.low_level], want.phase + want.amplitude + [want.low_level]):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    # FusionNet
    sd = torch.load(os.path.join(REF, "src/fusion_net/fusion_net.pt"), map_location="cpu")
    ref = RefFusion()
    ref.load_state_dict(sd)
    ref.eval()
    g = torch.Generator().manual_seed(5)
    r = lambda c: torch.rand((1, c, 40, 56), generator=g)
    base, ada, ph, other, maps = r(3), r(3), r(3), r(6), r(3)
    with torch.no_grad():
        want = ref(base, ada, ph, other, maps, variant=0)
        got = nets_cpu.fusionnet_forward(sd, base, ada, ph, other, maps, 0)
    assert (got - want).abs().max().item() <= 1e-6
