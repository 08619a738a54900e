"""GPU parity of the three networks (host mirrors over the C ABI) against the fixtures produced by the
reference's own classes, and against the oracle at a second size."""
import math
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import layout_cpu, nets_cpu, synth
from vfi_amd.fusion_net.fusion_adacofnet import AdaCoFNet
from vfi_amd.fusion_net.fusion_net import FusionNet
from vfi_amd.phase_net.phase_net import PhaseNet
from vfi_amd.values import DecompValues

pytestmark = pytest.mark.gpu

# fp32 tolerance: identical arithmetic in a different summation order (MFMA k-order vs oneDNN), over
# <= 4608-term dot products of O(1) values; errors stay ~1e-6, bound stated generously.
TOL = 3e-5


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.mark.parametrize("tag", ["64x64", "40x72"])
def test_fusionnet_matches_reference_fixture(tag, device):
    g = _load("fusionnet_" + tag)
    net = FusionNet().to(device)
    net.load_state_dict(nets_cpu.fusionnet_random_state_dict(int(g["seed"])))
    net.eval()
    t = lambda k: torch.from_numpy(g[k]).to(device)
    for variant in (0, 1):
        out = net(t("base"), t("adacof"), t("phase"), t("other"), t("maps"), variant=variant)
        assert np.abs(out.cpu().numpy() - g[f"out_variant{variant}"]).max() <= TOL


def test_fusionnet_with_trained_weight_statistics_matches_oracle(device):
    """FusionNet (5x5 direct-MFMA layers) with every tensor drawn to the mean / std / range of the reference's trained
    fusion_net.pt (tests/golden/trained_weight_stats.json) against the oracle, both variants, at the fixture's inputs."""
    import trained_stats
    g = _load("fusionnet_64x64")
    sd = trained_stats.state_dict_like_trained("fusionnet", nets_cpu.fusionnet_random_state_dict(0), seed=5)
    net = FusionNet().to(device)
    net.load_state_dict(sd)
    net.eval()
    c = lambda k: torch.from_numpy(g[k])
    for variant in (0, 1):
        with torch.no_grad():
            ref = nets_cpu.fusionnet_forward(sd, c("base"), c("adacof"), c("phase"), c("other"), c("maps"), variant)
        out = net(*(c(k).to(device) for k in ("base", "adacof", "phase", "other", "maps")), variant=variant)
        scale = max(1.0, float(ref.abs().max()))
        assert (out.cpu() - ref).abs().max().item() <= TOL * scale, (variant, scale)


@pytest.mark.parametrize("tag", ["64x96", "40x50"])
def test_adacofnet_matches_reference_fixture(tag, device):
    g = _load("adacofnet_" + tag)
    args = types.SimpleNamespace(kernel_size=5, dilation=1, gpu_id=0)
    net = AdaCoFNet(args).to(device)
    net.load_state_dict(nets_cpu.adacofnet_random_state_dict(int(g["seed"])))
    net.eval()
    f0, f2 = torch.from_numpy(g["frame0"]).to(device), torch.from_numpy(g["frame2"]).to(device)
    t1, t2, fr, mask = net(f0, f2)
    if tag == "64x96":  # no width padding -> t1 is comparable (see fusion_adacofnet.py:225)
        np.testing.assert_allclose(t1.cpu().numpy(), g["t1"], atol=TOL)
    np.testing.assert_allclose(t2.cpu().numpy(), g["t2"], atol=TOL)
    np.testing.assert_allclose(fr.cpu().numpy(), g["frame1"], atol=TOL)
    np.testing.assert_allclose(mask.cpu().numpy(), g["mask"], atol=1e-4)
    if "head_w1" in g:
        mean = torch.tensor(nets_cpu.CHANNEL_MEANS, device=device).view(1, 3, 1, 1)
        heads = net.get_kernel(f0 - mean, f2 - mean)
        for name, h in zip(("w1", "a1", "b1", "w2", "a2", "b2", "occ"), heads):
            np.testing.assert_allclose(h[..., ::4, ::4].cpu().numpy(), g["head_" + name], atol=TOL)


def _phasenet_inputs(g):
    h, w, height = int(g["h"]), int(g["w"]), int(g["height"])
    batch = synth.synthetic_vals(int(g["input_seed"]), 6, h, w, height)
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(batch, 2))
    return height, vin


@pytest.mark.parametrize("tag", ["32x48", "96x112"])
def test_phasenet_matches_reference_fixture(tag, device):
    g = _load("phasenet_" + tag)
    step = int(g["step"])
    height, vin = _phasenet_inputs(g)
    pyr = types.SimpleNamespace(height=height, nbands=4)
    net = PhaseNet(pyr, device)
    net.load_state_dict(nets_cpu.phasenet_random_state_dict(int(g["weight_seed"])))
    net.eval()
    dv = DecompValues(vin.high_level.to(device), [p.to(device) for p in vin.phase],
                      [a.to(device) for a in vin.amplitude], vin.low_level.to(device))
    normed = net.normalize_vals(dv)
    out = net(normed)
    torch.cuda.synchronize()
    np.testing.assert_allclose(net.max_low_level.cpu().numpy(), g["max_low"], rtol=1e-6)
    for k, mx in enumerate(net.max_amplitudes):
        np.testing.assert_allclose(mx.cpu().numpy(), g[f"max_amp{k}"], rtol=1e-6)
    if step == 1:
        for k in range(len(normed.phase)):
            np.testing.assert_allclose(normed.phase[k].cpu().numpy(), g[f"norm_phase{k}"], atol=1e-6)
            np.testing.assert_allclose(normed.amplitude[k].cpu().numpy(), g[f"norm_amp{k}"], atol=1e-6)
        np.testing.assert_allclose(normed.low_level.cpu().numpy(), g["norm_low"], atol=1e-6)
    items = [("high", out.high_level), ("low", out.low_level)]
    items += [(f"phase{k}", p) for k, p in enumerate(out.phase)] + [(f"amp{k}", a) for k, a in enumerate(out.amplitude)]
    for name, t in items:
        ref = g["out_" + name]
        got = t[..., ::step, ::step].cpu().numpy()
        assert got.shape == ref.shape, name
        scale = max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= 1e-4 * scale, (name, np.abs(got - ref).max(), scale)
    # the values may also arrive without the concat fast path (plain DecompValues)
    plain = DecompValues(normed.high_level, [p.contiguous() for p in normed.phase],
                         [a.contiguous() for a in normed.amplitude], normed.low_level)
    out2 = net(plain)
    for a, b in zip(out.phase + out.amplitude, out2.phase + out2.amplitude):
        assert torch.equal(a, b)


def test_phasenet_partial_levels_and_protocol(device):
    pyr = types.SimpleNamespace(height=6, nbands=4)
    net = PhaseNet(pyr, device)
    batch = synth.synthetic_vals(5, 6, 32, 48, 6)
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(batch, 2))
    dv = DecompValues(vin.high_level.to(device), [p.to(device) for p in vin.phase],
                      [a.to(device) for a in vin.amplitude], vin.low_level.to(device))
    with pytest.raises(RuntimeError):
        net(dv)                               # normalize_vals must come first
    out = net(net.normalize_vals(dv), m=2)    # hierarchical `m` of phase_net.py:107-110,91-93
    assert out.phase[0] == 0 and out.phase[1] == 0 and torch.is_tensor(out.phase[2])


def test_architecture_phasenet_image_in_image_out(device):
    # reference src/phase_net/architecture.py surface: (prediction, vals_pred, vals_target) from Lab channel-images
    from oracle import pyramid_cpu
    from vfi_amd.phase_net.architecture import PhaseNet as ArchPhaseNet
    h, w = 64, 96
    height = layout_cpu.calc_pyr_height(h, w)
    f0, _, f2 = synth.translating_pair(4, h, w)
    img = torch.from_numpy(np.concatenate([f0, f2], 0))                         # 6 channel-images
    sd = nets_cpu.phasenet_random_state_dict(1)
    # oracle: pyramid -> layout -> PhaseNet -> inverse pyramid
    opyr = pyramid_cpu.Pyramid(height)
    vin = layout_cpu.get_concat_layers_inf(layout_cpu.separate_vals(opyr.filter(img), 2))
    normed, state = nets_cpu.phasenet_normalize(vin)
    with torch.no_grad():
        ref = opyr.inv_filter(nets_cpu.phasenet_forward(sd, normed, state, height))
    net = ArchPhaseNet(height, device, num_img=2, scale_factor=math.sqrt(2), nbands=4)
    net.core.load_state_dict(sd)
    pred, vals_pred, target = net(img.to(device))
    assert target is None and pred.shape == (3, h, w) and len(vals_pred.phase) == height - 2
    assert (pred.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # hierarchical form: 3 frames (2 inputs + target), m levels predicted, the finest ones taken from the target
    img3 = torch.cat((img, torch.from_numpy(synth.translating_pair(4, h, w)[1])), 0)
    pred_m, vals_m, tgt = net(img3.to(device), m=2)
    assert pred_m.shape == (3, h, w) and tgt is not None and torch.isfinite(pred_m).all()
    # high_level=True copies the high residual of another prediction in
    pred_h, vals_h, _ = net(img.to(device), high_level=True, ada_pred=img[:3].to(device))
    assert vals_h.high_level.abs().max().item() > 0 and torch.isfinite(pred_h).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w", [(3, 96, 128), (2, 74, 83), (3, 9, 15), (1, 65, 67)])
def test_phasenet_predict_equals_conv_then_emit(n, h, w, device):
    """vfi_phasenet_predict (one pass: 1x1 tanh map + the level's outputs; phase_net.py:149-168) against the two calls it
    replaces -- vfi_conv2d(act = tanh) and vfi_phasenet_emit -- on channel slices of the block buffers, streaming sizes (4 and 2
    pixels per thread), a small level and an odd plane (both fall back to the two launches inside the library)."""
    from vfi_amd import _lib, ops
    g = torch.Generator().manual_seed(h * w + n)
    fp = torch.randn((n, 72, h, w), generator=g).to(device)
    x = torch.rand((n, 88, h, w), generator=g).to(device)
    mx = (torch.rand((n,), generator=g) + 0.5).to(device)
    pc = ops.PackedConv(torch.randn((8, 64, 1, 1), generator=g) / 8.0, torch.randn((8,), generator=g) * 0.1, device=device)
    amp_in = x[:, 80:]
    ref_fp = fp.clone()
    ops.conv2d(ref_fp[:, :64], pc, "zeros", "tanh", out=ref_fp[:, 64:])
    p_ref, a_ref = torch.empty((n, 4, h, w), device=device), torch.empty((n, 4, h, w), device=device)
    _lib.call("vfi_phasenet_emit", ref_fp[:, 64:].data_ptr(), ref_fp.stride(0), amp_in.data_ptr(), x.stride(0), mx.data_ptr(),
              p_ref.data_ptr(), a_ref.data_ptr(), n, h * w, _lib.stream_ptr())
    p_out, a_out = torch.empty_like(p_ref), torch.empty_like(a_ref)
    _lib.call("vfi_phasenet_predict", fp.data_ptr(), fp.stride(0), pc.packed.data_ptr(), pc.bias.data_ptr(), amp_in.data_ptr(), x.stride(0),
              mx.data_ptr(), fp[:, 64:].data_ptr(), fp.stride(0), p_out.data_ptr(), a_out.data_ptr(), n, 64, h, w, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(fp[:, :64], ref_fp[:, :64])                       # the features are only read
    if h * w >= 4096 and (h * w) % 2 == 0:      # the streaming kernels: the same FMA chain over the channels in both
        assert torch.equal(fp[:, 64:], ref_fp[:, 64:]) and torch.equal(p_out, p_ref) and torch.equal(a_out, a_ref)
    else:                                       # matrix-core kernel inside the library, without / with the caller's split-K workspace
        for got, ref in ((fp[:, 64:], ref_fp[:, 64:]), (p_out, p_ref), (a_out, a_ref)):
            assert (got - ref).abs().max().item() <= 2e-6
