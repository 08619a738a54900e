"""Scoring on device against plain torch-CPU statements of reference
src/evaluation/evaluate_image.py:22-28 and of piq's published psnr / ssim formulas."""
import math
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from vfi_amd.evaluation import evaluate_image as ei

pytestmark = pytest.mark.gpu


def _ssim_ref(x, y, ks=11, sigma=1.5, k1=0.01, k2=0.03):
    f = max(1, round(min(x.shape[-2:]) / 256))
    x, y = x.unsqueeze(0), y.unsqueeze(0)
    if f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    c = x.shape[1]
    co = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-co ** 2 / (2 * sigma ** 2)); g = g / g.sum()
    k = (g[:, None] * g[None, :]).expand(c, 1, ks, ks)
    conv = lambda t: F.conv2d(t, k, groups=c)
    mx, my = conv(x), conv(y)
    sxx, syy, sxy = conv(x * x) - mx * mx, conv(y * y) - my * my, conv(x * y) - mx * my
    cs = (2 * sxy + k2 ** 2) / (sxx + syy + k2 ** 2)
    return float(((2 * mx * my + k1 ** 2) / (mx * mx + my * my + k1 ** 2) * cs).mean())


@pytest.mark.parametrize("dim", [96, 512])
def test_evaluate_image_matches_reference_expressions(dim, device):
    g = torch.Generator().manual_seed(dim)
    pred = torch.rand((3, dim + 8, dim + 8), generator=g)
    tgt = (pred + 0.05 * torch.randn((3, dim + 8, dim + 8), generator=g)).clamp(0, 1)
    got = ei.evaluate_image(types.SimpleNamespace(dim=dim), pred.to(device), tgt.to(device))
    c = lambda t: t[:, 4:4 + dim, 4:4 + dim]
    p, t = c(pred).double(), c(tgt).double()
    d = p - t
    assert abs(got[0] - _ssim_ref(c(pred), c(tgt))) <= 2e-5
    assert math.isnan(got[1])
    assert abs(got[2] - 10 * math.log10(1.0 / (float((d * d).mean()) + 1e-8))) <= 1e-6      # piq.psnr
    assert abs(got[3] - float(torch.sqrt((d * d).sum()))) <= 1e-6                              # ssd (evaluate_image.py:24)
    assert abs(got[4] - float(d.sum())) <= 1e-6 and abs(got[5] - float(d.mean())) <= 1e-9      # signed l1 / mse (:25-26)
    assert abs(got[6] - float(torch.var(d))) <= 1e-9                                           # :27
