"""End-to-end parity of the fused per-frame path (product, on the MI355X) against the oracle pipeline
(CPU) on seeded synthetic triplets, with the tolerances of BASELINE.md section 3."""
import math
import os
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


# per-stage dB of the full-size parity tests, written to gpurun_out/parity_full_size.json (copied to profiles/ by hand)
_PARITY_LOG = {}


def _write_parity_log():
    import json
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_full_size.json"), "w") as f:
            json.dump(_PARITY_LOG, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _models(device, weights):
    args = types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0)
    adacof = Model(args)
    adacof.load(weights["adacof"])
    adacof.eval()
    fusion = FusionNet().to(device)
    fusion.load_state_dict(weights["fusionnet"])
    fusion.eval()
    return FusionInterpolator(adacof, fusion, weights["phasenet"], device)


# The reference's formula for `ada_uncertainty` (src/fusion_net/interpolate_twoframe.py:217-225: |phase| and |amplitude|
# differences of the six coarsest levels -> reconstruction -> x150 -> distance from a 50x50 median) loses CONDITIONING_DB of
# agreement between its input images and the map (tests/test_oracle_conditioning.py measures 50-57 dB on the oracle alone).
CONDITIONING_DB = 57.0
ADA_FLOOR_DB = 48.0


def _assert_stage_parity(got, ref, f1_true, label, ada_floor=ADA_FLOOR_DB):
    """Every stage output of the fused frame against the oracle pipeline: >= 60 dB on [0,1] images.

    `ada_uncertainty` is checked in two parts, because its formula amplifies a perturbation of its INPUT images by
    50-57 dB (CONDITIONING_DB above).  (a) The stage itself -- the oracle's uncertainty_maps() applied to the product's OWN
    ada_pred / phase_pred -- must reproduce the product's map at >= 60 dB (it does at ~99 dB).  (b) End to end the bound is
    derived from what the stage was fed: min(60, PSNR(phase_pred) - CONDITIONING_DB), never below `ada_floor`
    (ADA_FLOOR_DB = 48 dB).  That bound cannot account for phase coefficients that come out on the other side of the
    +-pi cut (a 2 pi step in the stage's input, not a rounding error), so the 1280x720 and 1920x1080 tests, where such
    coefficients exist, pass ada_floor = 40 and call `_assert_ada_uncertainty_explained` below, which restores the
    60 dB bar on the WHOLE map once exactly those coefficients are accounted for.  Everything downstream of the map
    (`final`) is held to the 60 dB bar."""
    from oracle import layout_cpu, pyramid_cpu, uncertainty_cpu
    report = {k: _psnr(got[k].cpu(), ref[k]) for k in ref if not k.startswith("_")}
    h, w = ref["final"].shape[-2:]
    opyr = pyramid_cpu.Pyramid(layout_cpu.calc_pyr_height(h, w), 4, np.sqrt(2))
    pu, au = uncertainty_cpu.uncertainty_maps(opyr, got["ada_pred"][0].cpu(), got["phase_pred"][0].cpu())
    report["ada_uncertainty | own inputs"] = _psnr(got["ada_uncertainty"].cpu(), au)
    report["phase_uncertainty | own inputs"] = _psnr(got["phase_uncertainty"].cpu(), pu)
    print(label, report)
    ada_bound = max(ada_floor, min(60.0, min(report["phase_pred"], report["ada_pred"]) - CONDITIONING_DB)) if ada_floor >= ADA_FLOOR_DB else ada_floor
    for k, v in report.items():
        assert v >= (ada_bound if k == "ada_uncertainty" else 60.0), (k, ada_bound, report)
    # |PSNR(HIP, GT) - PSNR(CPU, GT)| <= 0.01 dB on the analytic middle frame
    assert abs(_psnr(got["final"].cpu()[0], f1_true) - _psnr(ref["final"][0], f1_true)) <= 0.01
    return report


def _wrapped(p_ref, a_ref, p_gpu, a_gpu):
    """Coefficients whose PHASE VALUE differs by about 2 pi between the two pipelines although the complex coefficient
    agrees to the pyramid's tolerance (tests/test_pyramid_gpu.py: 5e-5 of the level's largest amplitude): atan2 put them
    on opposite sides of the +-pi cut.  Returns (mask of those, number of coefficients whose phase differs by more than 1 rad
    WITHOUT being such a wrap)."""
    scale = max(1e-3, float(a_ref.max()))
    far = (p_ref - p_gpu).abs() > 1.0
    same_coeff = (torch.polar(a_ref, p_ref) - torch.polar(a_gpu, p_gpu)).abs() <= 5e-5 * scale
    near_cut = (p_ref.abs() > math.pi - 1e-2) & (p_gpu.abs() > math.pi - 1e-2)
    wrap = far & same_coeff & near_cut
    # a phase that is far off WITHOUT the wrap excuse is only acceptable where the coefficient is (numerically) zero:
    # there atan2 returns noise in both pipelines
    unexplained = far & ~wrap & (a_ref > 2e-4 * scale)
    return wrap, int(unexplained.sum())


def _assert_ada_uncertainty_explained(run, got, ref, f0, f2, weights, label):
    """The explained-set check of `ada_uncertainty` (reference src/fusion_net/interpolate_twoframe.py:168-225).

    atan2 is discontinuous at +-pi: a coefficient within rounding distance of the cut may come out as +pi in one fp32
    pipeline and -pi in the other.  Such a wrap is NOT an error of the coefficient, but it (1) changes PhaseNet's input by
    2 (in units of pi) at that coefficient, hence `phase_pred` slightly, and (2) changes |phase difference| of the second
    analysis by 2 pi; the map's formula then amplifies either by ~50 dB.  This test finds those coefficients by comparing the
    product's phases with the oracle's, asserts (i) that every large phase difference IS such a wrap (the complex
    coefficient agrees to 5e-5 of the level's scale and both phases lie within 1e-2 of the cut), (ii) that they are few
    (bound below), and (iii) that the oracle, re-run from PhaseNet's input onwards with exactly those coefficients' phases
    set to the product's side of the cut, reproduces the product's map at >= 60 dB on the WHOLE frame."""
    from vfi_amd import ops
    device = got["final"].device
    h, w = f0.shape[1:]
    pyr, phase_net = run._state(h, w)
    nlev = pyr.height - 2
    # (1) PhaseNet's input as the product computed it (Pyramid.filter into the concat layout, phase / pi)
    lab12 = torch.empty((6, h, w), dtype=torch.float32, device=device)
    ops.rgb2lab(f0.to(device), out=lab12[:3])
    ops.rgb2lab(f2.to(device), out=lab12[3:])
    vals, _bufs, _ = pyr.filter(lab12, concat_frames=2, phase_scale=1.0 / math.pi, amp_max_eps=phase_net.eps)
    rin = ref["_vals_input_raw"]
    n_coeff, wraps1, bad1, patched = 0, 0, 0, []
    for k in range(nlev):                                     # coarsest first, (3, 8, h_k, w_k)
        p_gpu, a_gpu = vals.phase[k].cpu() * math.pi, vals.amplitude[k].cpu()
        wrap, bad = _wrapped(rin.phase[k], rin.amplitude[k], p_gpu, a_gpu)
        n_coeff += wrap.numel()
        wraps1 += int(wrap.sum())
        bad1 += bad
        patched.append(torch.where(wrap, p_gpu, rin.phase[k]))
    del vals, _bufs
    type_ = type(rin)
    hook_in = lambda v: type_(v.high_level, patched, v.amplitude, v.low_level)

    # (2) the second analysis (ada_pred, rgb_pred), six coarsest levels, as the product computes it
    coarse = min(6, nlev)
    mask = ((1 << coarse) - 1) << (nlev - coarse)
    vb = pyr.filter(torch.cat((got["ada_pred"][0], got["phase_pred"][0]), 0), level_mask=mask, want_high=False)
    second = {k: (vb.phase[k].cpu(), vb.amplitude[k].cpu()) for k in range(nlev - coarse, nlev)}   # finest first, (24, 1, h_k, w_k)
    del vb
    stats = {"wraps2": 0, "bad2": 0, "n2": 0}

    def hook_second(v):
        phase = list(v.phase)
        for k, (p_gpu, a_gpu) in second.items():
            wrap, bad = _wrapped(v.phase[k], v.amplitude[k], p_gpu.reshape(v.phase[k].shape), a_gpu.reshape(v.phase[k].shape))
            stats["wraps2"] += int(wrap.sum())
            stats["bad2"] += bad
            stats["n2"] += wrap.numel()
            phase[k] = torch.where(wrap, p_gpu.reshape(v.phase[k].shape), v.phase[k])
        return type(v)(v.high_level, phase, v.amplitude, v.low_level)

    ref2 = pipeline_cpu.interp(f0, f2, weights, hooks={"vals_input": hook_in, "vals_second": hook_second}, reuse=ref,
                               stop_after="ada_uncertainty")
    rep = {"coefficients": n_coeff, "wrapped (PhaseNet input)": wraps1, "far but not wrapped (PhaseNet input)": bad1,
           "coefficients (2nd analysis, 6 coarsest)": stats["n2"], "wrapped (2nd analysis)": stats["wraps2"],
           "far but not wrapped (2nd analysis)": stats["bad2"],
           "phase_pred vs oracle": _psnr(got["phase_pred"].cpu(), ref["phase_pred"]),
           "phase_pred vs oracle with the wraps applied": _psnr(got["phase_pred"].cpu(), ref2["phase_pred"]),
           "ada_uncertainty vs oracle": _psnr(got["ada_uncertainty"].cpu(), ref["ada_uncertainty"]),
           "ada_uncertainty vs oracle with the wraps applied": _psnr(got["ada_uncertainty"].cpu(), ref2["ada_uncertainty"])}
    print(label, rep)
    assert bad1 == 0 and stats["bad2"] == 0, rep
    # the bound on the explained set.  Measured (profiles/r04_parity.txt): 5 459 of 44.3 M PhaseNet-input coefficients at
    # 1280x720 and 12 670 of 99.6 M at 1920x1080 (1.2-1.3e-4: the synthetic frames are sums of a few sinusoids, so many
    # band coefficients are real-negative up to rounding and their phase is +-pi by the sign of an imaginary part of a few
    # ulps), 0-1 of the 2-3e5 coefficients of the second analysis.  Allowed: 5e-4 of the coefficients (+ 4).
    assert wraps1 <= 5e-4 * n_coeff + 4 and stats["wraps2"] <= 5e-4 * stats["n2"] + 4, rep
    assert rep["ada_uncertainty vs oracle with the wraps applied"] >= 60.0, rep
    assert rep["phase_pred vs oracle with the wraps applied"] >= 60.0, rep
    return rep


@pytest.mark.parametrize("h,w", [(128, 160), (96, 96)])
def test_fused_frame_matches_oracle(h, w, device):
    weights = pipeline_cpu.seeded_weights(0)
    f0, f1_true, f2 = (torch.from_numpy(x) for x in synth.translating_pair(7, h, w))
    ref = pipeline_cpu.interp(f0, f2, weights, output_baseline=True)
    run = _models(device, weights)
    got = run(f0.to(device), f2.to(device), output_baseline=True)
    torch.cuda.synchronize()
    _assert_stage_parity(got, ref, f1_true, f"configs[0] fused frame {h}x{w} vs oracle:")


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
    # a second interpolator (own pyramid plan / workspace, same weights) must reproduce every stage output bit for bit:
    # nothing on the path depends on allocation addresses or on state left behind by an earlier frame
    out2 = _models(device, weights)(f0, f2, output_baseline=True)
    for k, v in out.items():
        assert torch.equal(v, out2[k]), k
    # the same interpolator called again (plans, packed weights and workspaces reused) gives the same bits
    again = run(f0, f2, output_baseline=True)
    for k, v in out.items():
        assert torch.equal(v, again[k]), k
    assert out["flow_var_map"].shape == (1, 1, h, w)


def test_fused_frame_runs_at_4k(device):
    # the largest frame the FFT engine's tables cover (8192-point Bluestein lines for the 1528 x 2716
    # level): one fused frame end to end, size-independent properties only
    h, w = 2160, 3840
    run = _models(device, pipeline_cpu.seeded_weights(4))
    g = torch.Generator().manual_seed(7)
    f0 = torch.rand((3, h, w), generator=g).to(device)
    f2 = (0.5 * f0 + 0.5 * torch.rand((3, h, w), generator=g).to(device)).contiguous()
    out = run(f0, f2, output_baseline=True)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    for k in ("final", "phase_pred", "baseline", "phase_uncertainty", "ada_uncertainty", "flow_var_map"):
        assert out[k].min().item() >= 0.0 and out[k].max().item() <= 1.0, k
    assert out["final"].shape == (1, 3, h, w)
    assert (out["final"] - out["base"].clamp(0, 1)).abs().max().item() <= 1.0


def test_frame_is_capturable_into_a_hip_graph(device):
    # every library call only enqueues on its stream (no allocation / synchronisation inside): a whole frame, the
    # pyramid's FFT passes included, can be captured into a hipGraph and replayed on new inputs with identical results
    weights = pipeline_cpu.seeded_weights(3)
    run = _models(device, weights)
    a0, _, a2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(5, 64, 96))
    b0, _, b2 = (torch.from_numpy(x).to(device) for x in synth.translating_pair(6, 64, 96))
    want_a = run(a0, a2)["final"].clone()          # also warms up plans / packed weights
    want_b = run(b0, b2)["final"].clone()          # both eager references BEFORE the capture (round-1 order restored)
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
    assert torch.equal(got_b, want_b)
    assert not torch.equal(got_b, want_a)
    # ... and an eager run after the replays still gives the same bits
    assert torch.equal(run(b0, b2)["final"], want_b)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1..3] at 1280x720 against the CPU oracle (the oracle needs ~30 s per fused frame at this size)
# ---------------------------------------------------------------------------------------------------------------------
H720, W720 = 720, 1280


@pytest.fixture(scope="module")
def pair_720():
    return tuple(torch.from_numpy(x) for x in synth.translating_pair(11, H720, W720))


@pytest.mark.parametrize("weights_kind", ["seeded", "trained-statistics"])
def test_phasenet_branch_720p_matches_oracle(pair_720, device, weights_kind):
    """configs[1]: steerable-pyramid decompose -> PhaseNet -> reconstruct at 1280x720 (6 Lab channel-images in, 3 out;
    reference src/fusion_net/interpolate_twoframe.py:168-188) -- HIP pyramid kernels + PhaseNet vs the oracle.
    "trained-statistics": every tensor of the state dict drawn with the mean / std / range of the reference's trained
    phase_net.pt (tests/golden/trained_weight_stats.json): BatchNorm running variances of 0.02-0.1 put a x3-x7 gain in front
    of every ELU, so the F(4x4) Winograd layers run on activations of the magnitude a trained network produces."""
    import numpy as np
    from oracle import color_cpu, layout_cpu, nets_cpu, pyramid_cpu
    from vfi_amd import ops
    from vfi_amd.phase_net.phase_net import PhaseNet
    from vfi_amd.train.pyramid import Pyramid
    from vfi_amd.values import DecompValues
    f0, _, f2 = pair_720
    sd = pipeline_cpu.seeded_weights(0)["phasenet"]
    if weights_kind == "trained-statistics":
        import trained_stats
        sd = trained_stats.state_dict_like_trained("phasenet", sd, seed=3)
    height = layout_cpu.calc_pyr_height(H720, W720)
    assert height == 15
    lab = torch.cat((color_cpu.rgb2lab_single(f0), color_cpu.rgb2lab_single(f2)), 0).float()
    opyr = pyramid_cpu.Pyramid(height, 4, np.sqrt(2))
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(opyr.filter(lab), 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    with torch.no_grad():
        ref = opyr.inv_filter(nets_cpu.phasenet_forward(sd, normed, state, height))
    pyr = Pyramid(height, 4, np.sqrt(2), device)
    pyr.set_full_size(H720, W720)
    net = PhaseNet(pyr, device)
    net.load_state_dict(sd)
    net.eval()
    vals, bufs = pyr.filter(lab.to(device), concat_frames=2, phase_scale=1.0 / math.pi)
    pred = net(net.normalize_vals(vals, concat=bufs))
    got = pyr.inv_filter(DecompValues(0, pred.phase, pred.amplitude, pred.low_level)).cpu()
    psnr = _psnr(got, ref)
    print("configs[1] PhaseNet branch 720p vs oracle (%s weights):" % weights_kind, psnr, "dB; output rms", float(ref.pow(2).mean().sqrt()))
    _PARITY_LOG["720p PhaseNet branch, %s weights" % weights_kind] = psnr
    _write_parity_log()
    assert psnr >= 60.0, psnr


def test_adacof_720p_matches_oracle(pair_720, device):
    """configs[2]: one AdaCoFNet forward (U-Net + fused sampling + occlusion blend + flow-variance mask) at 1280x720:
    the 1080p-class routing of the Winograd kernel (run length, split-K cost model) against the oracle."""
    from oracle import nets_cpu
    f0, _, f2 = pair_720
    sd = pipeline_cpu.seeded_weights(0)["adacof"]
    with torch.no_grad():
        _, t2_ref, frame_ref, mask_ref = nets_cpu.adacofnet_forward(sd, f0.unsqueeze(0), f2.unsqueeze(0))
    args = types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0)
    adacof = Model(args)
    adacof.load(sd)
    adacof.eval()
    _, t2, frame, mask = adacof(f0.unsqueeze(0).to(device), f2.unsqueeze(0).to(device))
    rep = {"frame1": _psnr(frame.cpu(), frame_ref), "t2": _psnr(t2.cpu(), t2_ref), "mask": _psnr(mask.cpu(), mask_ref)}
    print("configs[2] AdaCoF 720p vs oracle:", rep)
    assert min(rep.values()) >= 60.0, rep


def test_fused_frame_720p_matches_oracle(pair_720, device):
    """configs[3] (the full fused frame) at 1280x720, every stage output against the oracle pipeline."""
    f0, f1_true, f2 = pair_720
    weights = pipeline_cpu.seeded_weights(0)
    ref = pipeline_cpu.interp(f0, f2, weights, output_baseline=True, keep_stages=True)
    run = _models(device, weights)
    got = run(f0.to(device), f2.to(device), output_baseline=True)
    torch.cuda.synchronize()
    _PARITY_LOG["720p"] = _assert_stage_parity(got, ref, f1_true, "configs[3] fused frame 720p vs oracle:", ada_floor=40.0)
    _PARITY_LOG["720p explained"] = _assert_ada_uncertainty_explained(run, got, ref, f0, f2, weights, "720p ada_uncertainty, explained set:")
    _write_parity_log()


def test_fused_frame_1080p_matches_oracle(device):
    """configs[3] at ITS OWN size: the full fused frame (output_baseline) at 1920x1080 on one seeded pair, every stage
    output against the oracle pipeline (reference src/fusion_net/interpolate_twoframe.py:148-330).  This is the only
    size at which the 1080p-only routing is on the path: the F(4x4) Winograd selection of the large layers, the Winograd
    run length, the Bluestein transforms of the 764 x 1358 level.  One oracle frame costs ~70 s of CPU."""
    h, w = 1080, 1920
    f0, f1_true, f2 = (torch.from_numpy(x) for x in synth.translating_pair(11, h, w))
    weights = pipeline_cpu.seeded_weights(0)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    ref = pipeline_cpu.interp(f0, f2, weights, output_baseline=True, keep_stages=True)
    run = _models(device, weights)
    got = run(f0.to(device), f2.to(device), output_baseline=True)
    torch.cuda.synchronize()
    _PARITY_LOG["1080p"] = _assert_stage_parity(got, ref, f1_true, "configs[3] fused frame 1080p vs oracle:", ada_floor=40.0)
    _PARITY_LOG["1080p explained"] = _assert_ada_uncertainty_explained(run, got, ref, f0, f2, weights, "1080p ada_uncertainty, explained set:")
    _write_parity_log()


# ---------------------------------------------------------------------------------------------------------------------
# frames in flight (bench.py / interpolate_video): same bits as sequential execution
# ---------------------------------------------------------------------------------------------------------------------
def test_two_frames_in_flight_equal_sequential_1080p(device):
    """N frames (a) one after another on one stream with one interpolator and (b) two in flight on two streams with two
    interpolators sharing the weights, exactly as bench.py's timed loop does -- at 1920x1080, bit for bit."""
    h, w = 1080, 1920
    weights = pipeline_cpu.seeded_weights(4)
    args = types.SimpleNamespace(model="vfi_amd.fusion_net.fusion_adacofnet", kernel_size=5, dilation=1, gpu_id=0)
    adacof = Model(args); adacof.load(weights["adacof"]); adacof.eval()
    fusion = FusionNet().to(device); fusion.load_state_dict(weights["fusionnet"]); fusion.eval()
    g = torch.Generator().manual_seed(5)
    pairs = [(torch.rand((3, h, w), generator=g).to(device), torch.rand((3, h, w), generator=g).to(device)) for _ in range(2)]
    n = 6
    seq_runner = FusionInterpolator(adacof, fusion, weights["phasenet"], device)
    seq = [seq_runner(*pairs[i % 2], output_baseline=True)["final"].clone() for i in range(n)]
    torch.cuda.synchronize()
    runners = [FusionInterpolator(adacof, fusion, weights["phasenet"], device) for _ in range(2)]
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]
    outs = []
    for i in range(n):                      # no synchronisation inside the loop: frames i and i+1 overlap
        with torch.cuda.stream(streams[i % 2]):
            outs.append(runners[i % 2](*pairs[i % 2], output_baseline=True)["final"])
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(outs[i], seq[i]), i
    assert torch.equal(seq[0], seq[2]) and not torch.equal(seq[0], seq[1])


# ---------------------------------------------------------------------------------------------------------------------
# the file entry point interp(args) (reference src/fusion_net/interpolate_twoframe.py:82-119,280-334)
# ---------------------------------------------------------------------------------------------------------------------
def test_interp_png_entry_point_with_center_crop(tmp_path, device):
    import numpy as np
    from PIL import Image
    from vfi_amd.fusion_net import interpolate_twoframe as it
    weights = pipeline_cpu.seeded_weights(5)
    dim, hh, ww = 96, 120, 136
    f0, _, f2 = synth.translating_pair(9, hh, ww)
    u8 = lambda f: (f.transpose(1, 2, 0) * 255 + 0.5).astype(np.uint8)
    a, b = u8(f0), u8(f2)
    Image.fromarray(a).save(tmp_path / "a.png"); Image.fromarray(b).save(tmp_path / "b.png")
    torch.save(weights["fusionnet"], tmp_path / "fusion_net.pt")
    torch.save(weights["phasenet"], tmp_path / "phase_net.pt")
    torch.save({"epoch": 0, "state_dict": weights["adacof"]}, tmp_path / "ckpt.pth")
    out = {k: str(tmp_path / f"{k}.png") for k in ("final", "phase", "adacof", "baseline")}
    args = types.SimpleNamespace(
        gpu_id=0, adacof_model="vfi_amd.fusion_net.fusion_adacofnet", adacof_kernel_size=5, adacof_dilation=1,
        adacof_checkpoint=str(tmp_path / "ckpt.pth"), adacof_config=None, checkpoint=str(tmp_path / "fusion_net.pt"),
        phase_net_checkpoint=str(tmp_path / "phase_net.pt"), first_frame=str(tmp_path / "a.png"),
        second_frame=str(tmp_path / "b.png"), output_frame=out["final"], output_phase=True, output_frame_phase=out["phase"],
        output_adacof=True, output_frame_adacof=out["adacof"], output_baseline=True, output_frame_baseline=out["baseline"],
        dim=dim, high_level=False, model=1, mode="alpha")
    it.interp(args)
    # oracle on the same centre crop (interpolate_twoframe.py:75-79,109-113), quantised like save_image
    crop = lambda img: it.crop_center(img, dim, dim)
    t = lambda img: torch.from_numpy(crop(img).astype(np.float32) / 255).permute(2, 0, 1).contiguous()
    ref = pipeline_cpu.interp(t(a), t(b), weights, output_baseline=True)
    quant = lambda x: x[0].mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    for key, name in (("final", "final"), ("phase", "phase_pred"), ("adacof", "ada_pred"), ("baseline", "baseline")):
        got = np.array(Image.open(out[key]))
        want = quant(ref[name])
        assert got.shape == (dim, dim, 3), key
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d == 0).mean() >= 0.99, (key, d.max(), (d == 0).mean())
    # second call with the same checkpoints reuses the cached interpolator (no per-call rebuild of plans / packs)
    n_cached = len(it._INTERPOLATORS)
    os.remove(out["final"])
    it.interp(args)
    assert len(it._INTERPOLATORS) == n_cached and os.path.exists(out["final"])
