"""CPU checks of the oracle pieces whose parity is pinned only by properties (pyramid spec, Lab)."""
import math

import numpy as np
import pytest
import torch

from oracle import color_cpu, layout_cpu, pyramid_cpu, synth


def test_lab_known_values_and_round_trip():
    white = torch.ones(3, 1, 1)
    lab = color_cpu.rgb2lab_single(white)
    np.testing.assert_allclose(lab[:, 0, 0].numpy(), [1.0, 128 / 255, 128 / 255], atol=2e-5)
    black = color_cpu.rgb2lab_single(torch.zeros(3, 1, 1))
    np.testing.assert_allclose(black[:, 0, 0].numpy(), [0.0, 128 / 255, 128 / 255], atol=1e-6)
    # sRGB red: L* 53.24, a* 80.09, b* 67.20 (standard D65 values)
    red = color_cpu.rgb2lab_single(torch.tensor([1.0, 0.0, 0.0]).view(3, 1, 1))
    np.testing.assert_allclose(red[:, 0, 0].numpy() * [100, 255, 255] - [0, 128, 128], [53.24, 80.09, 67.20], atol=0.02)
    # the other published D65 / 2-degree values of the sRGB primaries and of mid grey (CIE 1976 L*a*b* of IEC 61966-2-1 sRGB)
    for rgb1, want in (((0.0, 1.0, 0.0), (87.7347, -86.1827, 83.1793)), ((0.0, 0.0, 1.0), (32.2970, 79.1875, -107.8602)),
                       ((0.5, 0.5, 0.5), (53.3890, 0.0, 0.0)), ((1.0, 1.0, 0.0), (97.1393, -21.5537, 94.4780))):
        got = color_cpu.rgb2lab_single(torch.tensor(rgb1).view(3, 1, 1))[:, 0, 0].numpy() * [100, 255, 255] - [0, 128, 128]
        np.testing.assert_allclose(got, want, atol=0.02)
    rng = np.random.default_rng(0)
    rgb = torch.from_numpy(rng.random((3, 16, 16), dtype=np.float32))
    back = color_cpu.lab2rgb_single(color_cpu.rgb2lab_single(rgb))
    assert (back - rgb).abs().max().item() <= 1e-5      # round trip <= 1e-5


@pytest.mark.parametrize("h,w", [(64, 96), (90, 120), (256, 256)])
def test_pyramid_perfect_reconstruction_and_shapes(h, w):
    height = layout_cpu.calc_pyr_height(h, w)
    pyr = pyramid_cpu.Pyramid(height)
    f0, _, f2 = synth.translating_pair(0, h, w)
    img = torch.from_numpy(np.concatenate([f0, f2], 0))
    vals = pyr.filter(img)
    rec = pyr.inv_filter(vals)
    psnr = 10 * math.log10(1.0 / float(((rec - img) ** 2).mean()))
    assert psnr >= 100.0, psnr
    bands, low = synth.level_sizes(h, w, height)
    assert [tuple(p.shape[2:]) for p in vals.phase] == bands and tuple(vals.low_level.shape[2:]) == low
    assert vals.phase[0].shape[0] == 6 * 4 and vals.high_level.shape == (6, 1, h, w)


def test_pyramid_level_table_1080p():
    # ceil(d / 2^(k/2)); 17 levels at 1080p
    spec_sizes = synth.level_sizes(1080, 1920, 17)
    assert spec_sizes[0][:5] == [(1080, 1920), (764, 1358), (540, 960), (382, 679), (270, 480)]
    assert spec_sizes[1] == (6, 11) and len(spec_sizes[0]) == 15


def test_pyramid_scale2_even_sizes_match_upstream_crop_rule():
    # with scale_factor=2 and even sizes the windows are upstream's ceil((d-0.5)/2), centred on DC
    spec = pyramid_cpu.PyramidSpec(64, 96, 5, 4, 2.0)
    assert spec.sizes == [(64, 96), (32, 48), (16, 24), (8, 12)]
    dims = np.array([64, 96])
    start = (np.ceil((dims + 0.5) / 2) - np.ceil((np.ceil((dims - 0.5) / 2) + 0.5) / 2)).astype(int)
    assert spec.crop(0) == (slice(start[0], start[0] + 32), slice(start[1], start[1] + 48))
    img = torch.from_numpy(synth.translating_pair(1, 64, 96)[0])
    rec = pyramid_cpu.reconstruct(spec, pyramid_cpu.build(spec, img))
    assert (rec - img).abs().max().item() <= 1e-4


def test_pyramid_bands_are_analytic_and_energy_is_bounded():
    h, w = 64, 96
    height = layout_cpu.calc_pyr_height(h, w)
    spec = pyramid_cpu.PyramidSpec(h, w, height)
    img = torch.from_numpy(synth.translating_pair(2, h, w)[0][:1])
    coeff = pyramid_cpu.build(spec, img)
    band = torch.view_as_complex(coeff[1][0].contiguous())          # finest level, orientation 0
    spec_b = torch.fft.fftshift(torch.fft.fft2(band), dim=(-2, -1))[0]
    # analytic: energy only in the half plane |angle| < pi/2 (positive horizontal frequencies)
    left = spec_b[:, : w // 2 - 1].abs().pow(2).sum()
    right = spec_b[:, w // 2 + 1:].abs().pow(2).sum()
    assert left <= 1e-6 * right
