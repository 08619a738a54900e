"""GPU parity of the image-space stages (Lab, Gaussian, median, glue) against scipy (which the reference
calls directly) and the oracle's Lab restatement."""
import numpy as np
import pytest
import torch
from scipy.ndimage import gaussian_filter, median_filter

from oracle import color_cpu
from vfi_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", [(96, 128), (37, 61), (20, 200)])
def test_gaussian_matches_scipy(h, w, device):
    rng = np.random.default_rng(h)
    x = rng.random((2, h, w), dtype=np.float32)
    ref = np.stack([gaussian_filter(a, 5) for a in x])
    got = ops.gaussian_filter(torch.from_numpy(x).to(device), 5).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-6)


@pytest.mark.parametrize("h,w,size", [(96, 128, 50), (64, 72, 50), (30, 41, 50), (40, 40, 7), (33, 65, 4)])
def test_median_matches_scipy_exactly(h, w, size, device):
    rng = np.random.default_rng(w)
    x = (rng.standard_normal((1, h, w)) * 3).astype(np.float32)
    x[0, :5, :7] = 0.25                       # ties
    x[0, 10:14, :] = -0.0
    ref = np.stack([median_filter(a, size=size) for a in x])
    got = ops.median_filter(torch.from_numpy(x).to(device), size).cpu().numpy()
    assert np.array_equal(got, ref)          # a selection: bit exact


def test_median_smooth_map_1080p_window(device):
    # the real input is a smooth low-frequency map; check a crop of a 1080p launch against scipy
    h, w = 1080, 1920
    yy, xx = np.meshgrid(np.linspace(0, 6, h), np.linspace(0, 9, w), indexing="ij")
    x = (np.sin(yy) * np.cos(xx) + 0.1 * np.sin(7 * xx)).astype(np.float32)[None]
    got = ops.median_filter(torch.from_numpy(x).to(device), 50).cpu().numpy()
    ref = median_filter(x[0, :160, :200], size=50)
    assert np.array_equal(got[0, :100, :100], ref[:100, :100])   # interior of the crop unaffected by its border


def test_lab_matches_oracle_and_round_trips(device):
    rng = np.random.default_rng(0)
    rgb = rng.random((3, 48, 64), dtype=np.float32)
    rgb[:, 0, 0] = 0.0; rgb[:, 0, 1] = 1.0; rgb[:, 0, 2] = 0.04045; rgb[:, 0, 3] = 0.003
    t = torch.from_numpy(rgb)
    lab = ops.rgb2lab(t.to(device))
    np.testing.assert_allclose(lab.cpu().numpy(), color_cpu.rgb2lab_single(t).numpy(), atol=2e-6)
    back = ops.lab2rgb(lab)
    np.testing.assert_allclose(back.cpu().numpy(), rgb, atol=2e-5)
    # out-of-gamut Lab (as PhaseNet produces) clips like the oracle
    wild = torch.from_numpy((rng.random((3, 32, 32), dtype=np.float32) * 1.4 - 0.2))
    np.testing.assert_allclose(ops.lab2rgb(wild.to(device)).cpu().numpy(), color_cpu.lab2rgb_single(wild).numpy(), atol=3e-5)


def test_glue_ops(device):
    rng = np.random.default_rng(1)
    a = torch.from_numpy(rng.standard_normal((1, 3, 20, 30)).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal((1, 3, 20, 30)).astype(np.float32))
    got = ops.channel_mean_diff(a.to(device), b.to(device), 100.0, True).cpu()
    ref = ((a.mean(1) - b.mean(1)).abs() * 100).clamp(0, 1)
    assert (got - ref).abs().max().item() <= 1e-4
    got = ops.channel_mean_diff(a.to(device), None, 30.0, False).cpu()
    assert (got - a.mean(1) * 30).abs().max().item() <= 1e-5
    got = ops.absdiff(a.to(device), b.to(device), 5.0, True).cpu()
    assert (got - ((a - b).abs() * 5).clamp(0, 1)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("hi,wi,ho,wo,ac,relu", [(9, 15, 12, 20, False, False), (12, 20, 17, 29, False, False),
                                                 (10, 12, 20, 24, True, False), (8, 12, 16, 24, False, True),
                                                 (6, 11, 8, 15, False, False),
                                                 # >= 16 x 64 outputs, up-scaling: the LDS-staged tile kernel (float4 rows /
                                                 # ragged rows, several tiles in both directions, relu + residual)
                                                 (40, 70, 80, 140, True, False), (33, 45, 47, 66, False, False),
                                                 (20, 40, 40, 80, False, True), (100, 300, 200, 600, True, True),
                                                 (54, 96, 77, 135, False, False), (16, 64, 16, 64, True, False)])
def test_resize_bilinear_matches_torch(hi, wi, ho, wo, ac, relu, device):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(hi * wo)
    x = torch.randn((2, 5, hi, wi), generator=g)
    res = torch.randn((2, 5, ho, wo), generator=g)
    ref = F.interpolate(F.relu(x) if relu else x, size=(ho, wo), mode="bilinear", align_corners=ac) + res
    got = ops.resize_bilinear(x.to(device), (ho, wo), align_corners=ac, relu_input=relu, residual=res.to(device))
    assert (got.cpu() - ref).abs().max().item() <= 2e-6
    # into a channel slice of a wider tensor (PhaseNet block input)
    wide = torch.zeros((2, 9, ho, wo), device=device)
    ops.resize_bilinear(x.to(device), (ho, wo), align_corners=ac, relu_input=relu, out=wide[:, 2:7])
    ref2 = F.interpolate(F.relu(x) if relu else x, size=(ho, wo), mode="bilinear", align_corners=ac)
    assert (wide[:, 2:7].cpu() - ref2).abs().max().item() <= 2e-6 and wide[:, :2].abs().max().item() == 0
