"""GPU parity of the steerable-pyramid kernels (through the C ABI) against the oracle restatement
(oracle/pyramid_cpu.py; parity of the SPEC itself is unpinned, see its header), plus properties."""
import math

import numpy as np
import pytest
import torch

from oracle import layout_cpu, pyramid_cpu, synth
from vfi_amd.train.pyramid import Pyramid
from vfi_amd.values import DecompValues

pytestmark = pytest.mark.gpu

S2 = math.sqrt(2)


def _images(seed, n_pairs, h, w):
    out = []
    for i in range(n_pairs):
        f0, _, f2 = synth.translating_pair(seed + i, h, w)
        out += [f0, f2]
    return torch.from_numpy(np.concatenate(out, 0))


def _psnr(a, b):
    return 10 * math.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-30))


# (720, 1280) and (1080, 1920): BASELINE.json's sizes at COEFFICIENT level, N = 6 -- the team kernels of the long columns,
# the Bluestein levels (764 x 1358 ...) and the multi-level launches of the small levels are on the path only there
@pytest.mark.parametrize("h,w", [(64, 96), (90, 120), (128, 128), (65, 77), (720, 1280), (1080, 1920)])
def test_filter_matches_oracle(h, w, device):
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(1, 1, h, w)                                     # 6 channel-images
    ref = pyramid_cpu.Pyramid(height).filter(img)
    pyr = Pyramid(height, 4, S2, device)
    got = pyr.filter(img.to(device))
    assert [tuple(p.shape) for p in got.phase] == [tuple(p.shape) for p in ref.phase]
    assert (got.high_level.cpu() - ref.high_level).abs().max() <= 2e-5
    assert (got.low_level.cpu() - ref.low_level).abs().max() <= 2e-5 * max(1.0, ref.low_level.abs().max().item())
    for k in range(len(ref.phase)):
        a_ref, p_ref = ref.amplitude[k], ref.phase[k]
        a, p = got.amplitude[k].cpu(), got.phase[k].cpu()
        scale = max(1e-3, a_ref.max().item())
        assert (a - a_ref).abs().max().item() <= 3e-5 * scale, k
        # compare phases through the coefficient (phase wraps at +-pi are not errors)
        z_ref = torch.polar(a_ref, p_ref)
        z = torch.polar(a, p)
        assert (z - z_ref).abs().max().item() <= 5e-5 * scale, k
        assert p.abs().max().item() <= math.pi + 1e-6


@pytest.mark.parametrize("h,w", [(64, 96), (90, 120), (256, 256), (720, 1280), (1080, 1920)])
def test_round_trip_and_inverse_vs_oracle(h, w, device):
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(2, 1, h, w)[:3]
    pyr = Pyramid(height, 4, S2, device)
    vals = pyr.filter(img.to(device))
    rec = pyr.inv_filter(vals).cpu()
    assert _psnr(rec, img) >= 90.0, _psnr(rec, img)                  # >= 90 dB pyramid round trip
    # synthesis alone: invert the ORACLE's values on the GPU and compare with the oracle's inverse
    opyr = pyramid_cpu.Pyramid(height)
    ovals = opyr.filter(img)
    orec = opyr.inv_filter(ovals)
    dv = DecompValues(ovals.high_level.to(device), [p.to(device) for p in ovals.phase],
                      [a.to(device) for a in ovals.amplitude], ovals.low_level.to(device))
    assert (pyr.inv_filter(dv).cpu() - orec).abs().max().item() <= 3e-5


def test_inverse_of_arbitrary_values_matches_oracle(device):
    # PhaseNet-like outputs: arbitrary (non-analytic) phase/amplitude, high = 0
    h, w = 64, 96
    height = layout_cpu.calc_pyr_height(h, w)
    v = synth.synthetic_vals(3, 3, h, w, height)
    v = DecompValues(torch.zeros_like(v.high_level), v.phase, v.amplitude, v.low_level)
    ref = pyramid_cpu.Pyramid(height).inv_filter(v)
    pyr = Pyramid(height, 4, S2, device)
    dv = DecompValues(v.high_level.to(device), [p.to(device) for p in v.phase], [a.to(device) for a in v.amplitude],
                      v.low_level.to(device))
    got = pyr.inv_filter(dv).cpu()
    assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # dropped levels (scalar 0, as phase_net.py:91-93 / utils.py:242-320) == explicit zeros
    keep = 2
    masked = DecompValues(0, [p if k < keep else 0 for k, p in enumerate(dv.phase)],
                          [a if k < keep else 0 for k, a in enumerate(dv.amplitude)], 0)
    pyr.set_full_size(h, w)
    ref2 = pyramid_cpu.Pyramid(height).inv_filter(layout_cpu.get_last_value_levels(
        DecompValues(torch.zeros_like(v.high_level), v.phase, v.amplitude, v.low_level), keep))
    assert (pyr.inv_filter(masked).cpu() - ref2).abs().max().item() <= 1e-4 * max(1.0, ref2.abs().max().item())


def test_concat_layout_equals_reference_shuffles(device):
    # filter(concat_frames=2) == get_concat_layers_inf(separate_vals(filter(x), 2)) with phase/pi
    h, w = 64, 96
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(4, 1, h, w).to(device)
    pyr = Pyramid(height, 4, S2, device)
    plain = pyr.filter(img)
    cpu = DecompValues(plain.high_level.cpu(), [p.cpu() for p in plain.phase], [a.cpu() for a in plain.amplitude],
                       plain.low_level.cpu())
    want = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(cpu, 2))
    got, bufs = pyr.filter(img, concat_frames=2, phase_scale=1.0 / math.pi)
    assert torch.equal(got.high_level.cpu(), want.high_level) and torch.equal(got.low_level.cpu(), want.low_level)
    for k in range(len(want.phase)):
        assert torch.equal(got.amplitude[k].cpu(), want.amplitude[k])
        assert (got.phase[k].cpu() - want.phase[k] / math.pi).abs().max().item() <= 1e-6
        assert bufs[k].shape[1] == (81 if k == 0 else 88)


def test_scfpyr_build_reconstruct_surface(device):
    from vfi_amd.steerable.SCFpyr_PyTorch import SCFpyr_PyTorch
    h, w = 64, 96
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(5, 1, h, w)[:2]
    sc = SCFpyr_PyTorch(height=height, nbands=4, scale_factor=S2, device=device)
    coeff = sc.build(img.to(device).unsqueeze(1))
    ref = pyramid_cpu.build(pyramid_cpu.PyramidSpec(h, w, height), img)
    assert len(coeff) == len(ref) and tuple(coeff[1][0].shape) == tuple(ref[1][0].shape)
    for k in range(1, len(ref) - 1):
        for b in range(4):
            scale = max(1e-3, ref[k][b].abs().max().item())
            assert (coeff[k][b].cpu() - ref[k][b]).abs().max().item() <= 5e-5 * scale
    rec = sc.reconstruct(coeff).cpu()
    assert _psnr(rec, img) >= 90.0


def test_full_size_1080p_properties(device):
    # BASELINE size: 6 channel-images at 1920x1080, height 17 (15 band levels)
    h, w = 1080, 1920
    height = layout_cpu.calc_pyr_height(h, w)
    assert height == 17
    g = torch.Generator().manual_seed(0)
    img = torch.rand((6, h, w), generator=g).to(device)
    pyr = Pyramid(height, 4, S2, device)
    vals = pyr.filter(img)
    assert [tuple(p.shape[2:]) for p in vals.phase][:4] == [(1080, 1920), (764, 1358), (540, 960), (382, 679)]
    rec = pyr.inv_filter(DecompValues(vals.high_level[:3], [p[:12] for p in vals.phase],
                                      [a[:12] for a in vals.amplitude], vals.low_level[:3]))
    assert _psnr(rec, img[:3]) >= 90.0
    # linearity of analysis->synthesis restricted to a level subset (a fixed linear band-pass filter)
    sub = lambda v: DecompValues(0, [p[:12] if k < 2 else 0 for k, p in enumerate(v.phase)],
                                 [a[:12] if k < 2 else 0 for k, a in enumerate(v.amplitude)], 0)
    pyr.set_full_size(h, w)
    y1 = pyr.inv_filter(sub(vals))
    vals2 = pyr.filter(0.5 * img)
    y2 = pyr.inv_filter(sub(vals2))
    assert (y1 * 0.5 - y2).abs().max().item() <= 2e-5


def test_band_filter_equals_analysis_plus_masked_synthesis(device):
    # inv_filter(get_last_value_levels(filter(x), 1)) == one radial gain (partition of unity of the 4 orientations)
    h, w = 90, 120
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(6, 1, h, w)[:3]
    opyr = pyramid_cpu.Pyramid(height)
    ref = opyr.inv_filter(layout_cpu.get_last_value_levels(opyr.filter(img), 1))
    ref2 = opyr.inv_filter(layout_cpu.get_first_value_levels(opyr.filter(img), 3))
    pyr = Pyramid(height, 4, S2, device)
    got = pyr.band_filter(img.to(device), level_mask=1, keep_high=True).cpu()
    assert (got - ref).abs().max().item() <= 2e-5
    nlev = height - 2
    got2 = pyr.band_filter(img.to(device), level_mask=0b111 << (nlev - 3), keep_low=True).cpu()
    assert (got2 - ref2).abs().max().item() <= 2e-5
    # all levels + both residuals == identity
    full = pyr.band_filter(img.to(device), level_mask=(1 << nlev) - 1, keep_high=True, keep_low=True).cpu()
    assert (full - img).abs().max().item() <= 2e-5


def test_amplitude_maxima_come_with_the_bands(device):
    # Pyramid.filter(concat_frames=2, amp_max_eps=eps): the per-(level, colour) maxima PhaseNet.normalize_vals needs
    # (src/phase_net/phase_net.py:55), reduced inside the kernel that writes the amplitudes -- equal, bit for bit, to a
    # separate max over the written amplitudes
    from vfi_amd import ops
    h, w = 90, 120
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(8, 1, h, w).to(device)
    pyr = Pyramid(height, 4, S2, device)
    vals, bufs, amp_max = pyr.filter(img, concat_frames=2, phase_scale=1.0 / math.pi, amp_max_eps=1e-8)
    assert amp_max.shape == (height - 2, 3)
    for k, amp in enumerate(vals.amplitude):                  # coarsest first, (3, 8, h_k, w_k) views
        assert torch.equal(amp_max[k], ops.batch_max(amp, 1e-8)), k
    plain, _ = pyr.filter(img, concat_frames=2, phase_scale=1.0 / math.pi)
    for a, b in zip(vals.amplitude, plain.amplitude):
        assert torch.equal(a, b)


def test_round_trip_4k_levels_with_8192_point_bluestein(device):
    # 3840x2160: level 1 is 1528 x 2716 = 4*7*97 wide -> Bluestein on 8192 points, one line per workgroup (the largest
    # transform the LDS engine takes); one image, reconstruction only
    h, w = 2160, 3840
    height = layout_cpu.calc_pyr_height(h, w)
    g = torch.Generator().manual_seed(1)
    img = torch.rand((1, h, w), generator=g).to(device)
    pyr = Pyramid(height, 4, S2, device)
    vals = pyr.filter(img)
    assert tuple(vals.phase[1].shape[2:]) == (1528, 2716)
    rec = pyr.inv_filter(vals)
    assert _psnr(rec.cpu(), img.cpu()) >= 90.0


@pytest.mark.parametrize("h,w", [(128, 160), (90, 120), (256, 256)])
def test_results_do_not_depend_on_stale_lds(h, w, device):
    """LDS is not cleared between workgroups.  With every CU's LDS filled with NaNs right before each call, analysis and
    synthesis (Bluestein levels included: their chirp tables have a tail that the zero-padding lanes read) must give the
    same bits as without -- a kernel that reads LDS it has not written would return NaNs here, instead of once in a
    while on a machine whose previous kernel happened to leave the wrong bytes behind."""
    from vfi_amd import _lib
    height = layout_cpu.calc_pyr_height(h, w)
    img = _images(3, 1, h, w).to(device)
    pyr = Pyramid(height, 4, S2, device)

    def run(poison):
        outs = []
        for concat in (None, 2):
            if poison:
                _lib.call("vfi_debug_poison_lds", _lib.stream_ptr())
            v = pyr.filter(img, concat_frames=concat) if concat is None else pyr.filter(img, concat_frames=concat, amp_max_eps=1e-8)[0]
            outs += [v.high_level.clone(), v.low_level.clone()] + [p.clone() for p in v.phase] + [a.clone() for a in v.amplitude]
            if concat is None:
                if poison:
                    _lib.call("vfi_debug_poison_lds", _lib.stream_ptr())
                outs.append(pyr.inv_filter(v).clone())
        return outs

    clean, dirty = run(False), run(True)
    for i, (a, b) in enumerate(zip(clean, dirty)):
        assert not torch.isnan(b).any(), i
        assert torch.equal(a, b), i
