"""GPU parity of the AdaCoF HIP kernels (through the C ABI) against the oracle and the
reference-generated fixtures, plus size-independent properties at BASELINE sizes."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import adacof_cpu
from vfi_amd.adacof.cupy_module.adacof import FunctionAdaCoF, adacof_fused

pytestmark = pytest.mark.gpu

# fp32 tolerance: the HIP kernel uses precomputed bilinear weights and fma contraction, the
# oracle the reference's left-to-right evaluation; both accumulate F*F terms of magnitude <= ~3.
ATOL = 2e-5

CASES = sorted(glob.glob(os.path.join(GOLDEN, "adacof_sampling_*.npz")))


def _dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[16:-4] for p in CASES])
def test_forward_matches_reference_fixture(path, device):
    g = np.load(path)
    out = FunctionAdaCoF.apply(_dev(g["input"], device), _dev(g["weight"], device),
                               _dev(g["offset_i"], device), _dev(g["offset_j"], device), int(g["dilation"]))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), g["output"], rtol=0, atol=ATOL)


def _rand_case(seed, n, h, w, f, amp=3.0):
    rng = np.random.default_rng(seed)
    k = f * f
    mk = lambda: (rng.standard_normal((n, k, h, w)) * amp).astype(np.float32)
    lg = rng.standard_normal((2, n, k, h, w)).astype(np.float32)
    W = (np.exp(lg) / np.exp(lg).sum(2, keepdims=True)).astype(np.float32)
    return dict(f0=rng.random((n, 3, h, w), dtype=np.float32), f2=rng.random((n, 3, h, w), dtype=np.float32),
                w1=W[0], a1=mk(), b1=mk(), w2=W[1], a2=mk(), b2=mk(),
                occ=rng.random((n, 1, h, w), dtype=np.float32))


@pytest.mark.parametrize("n,h,w,f,dil", [(1, 32, 64, 5, 1), (2, 20, 36, 5, 1), (1, 17, 23, 5, 1),
                                         (1, 16, 24, 3, 2), (1, 12, 16, 7, 1)])
def test_fused_matches_oracle(n, h, w, f, dil, device):
    c = _rand_case(7 + h, n, h, w, f)
    pad = (f - 1) * dil // 2
    o1 = adacof_cpu.adacof_forward(np.pad(c["f0"], ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="edge"),
                                   c["w1"], c["a1"], c["b1"], dil)
    o2 = adacof_cpu.adacof_forward(np.pad(c["f2"], ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="edge"),
                                   c["w2"], c["a2"], c["b2"], dil)
    frame, mask = adacof_cpu.blend_mask(o1, o2, c["occ"], c["w1"], c["a1"], c["b1"], c["w2"], c["a2"], c["b2"])
    d = {k: _dev(v, device) for k, v in c.items()}
    t1, t2, fr, mk = adacof_fused(d["f0"], d["f2"], d["w1"], d["a1"], d["b1"], d["w2"], d["a2"], d["b2"],
                                  d["occ"], dil)
    torch.cuda.synchronize()
    np.testing.assert_allclose(t1.cpu().numpy(), o1, rtol=0, atol=ATOL)
    np.testing.assert_allclose(t2.cpu().numpy(), o2, rtol=0, atol=ATOL)
    np.testing.assert_allclose(fr.cpu().numpy(), frame, rtol=0, atol=ATOL)
    np.testing.assert_allclose(mk.cpu().numpy(), mask, rtol=1e-4, atol=2e-5)
    # optional outputs may be omitted
    _, _, fr2, mk2 = adacof_fused(d["f0"], d["f2"], d["w1"], d["a1"], d["b1"], d["w2"], d["a2"], d["b2"],
                                  d["occ"], dil, want_sides=False, want_mask=False)
    assert mk2 is None and torch.equal(fr2, fr)


def test_fused_large_offsets_mask_conditioning(device):
    # offsets of ~100 px with small spread: the pivoted one-pass variance must stay accurate
    c = _rand_case(11, 1, 24, 32, 5, amp=0.5)
    for k in ("a1", "b1", "a2", "b2"):
        c[k] = c[k] + np.float32(97.3)
    _, mask = adacof_cpu.blend_mask(np.zeros_like(c["f0"]), np.zeros_like(c["f0"]), c["occ"], c["w1"], c["a1"],
                                    c["b1"], c["w2"], c["a2"], c["b2"])
    d = {k: _dev(v, device) for k, v in c.items()}
    _, _, _, mk = adacof_fused(d["f0"], d["f2"], d["w1"], d["a1"], d["b1"], d["w2"], d["a2"], d["b2"], d["occ"], 1)
    np.testing.assert_allclose(mk.cpu().numpy(), mask, rtol=1e-3, atol=2e-5)


def test_full_size_properties_1080p(device):
    # BASELINE size (1080p padded to 1088x1920): properties that need no CPU oracle.
    n, h, w, f = 1, 1088, 1920, 5
    g = torch.Generator(device="cpu").manual_seed(5)
    frame0 = torch.rand((n, 3, h, w), generator=g).to(device)
    frame2 = torch.rand((n, 3, h, w), generator=g).to(device)
    wlog = torch.randn((2, n, 25, h, w), generator=g).to(device)
    W = torch.softmax(wlog, 2).contiguous()
    a = (torch.randn((4, n, 25, h, w), generator=g) * 2).clamp(-8, 8).to(device)
    occ = torch.rand((n, 1, h, w), generator=g).to(device)
    t1, t2, fr, mask = adacof_fused(frame0, frame2, W[0], a[0], a[1], W[1], a[2], a[3], occ, 1)
    # (1) fused == two plain forwards on replication-padded frames + blend
    pad = torch.nn.functional.pad
    p0 = pad(frame0, (2, 2, 2, 2), mode="replicate").contiguous()
    p2 = pad(frame2, (2, 2, 2, 2), mode="replicate").contiguous()
    u1 = FunctionAdaCoF.apply(p0, W[0], a[0], a[1], 1)
    u2 = FunctionAdaCoF.apply(p2, W[1], a[2], a[3], 1)
    assert (u1 - t1).abs().max().item() <= 1e-6 and (u2 - t2).abs().max().item() <= 1e-6
    assert (occ * u1 + (1 - occ) * u2 - fr).abs().max().item() <= 1e-6
    # (2) constant frame is reproduced (bilinear weights and softmax weights both sum to 1)
    const = torch.full_like(frame0, 0.625)
    _, _, frc, _ = adacof_fused(const, const, W[0], a[0], a[1], W[1], a[2], a[3], occ, 1, False, False)
    assert (frc - 0.625).abs().max().item() <= 1e-5
    # (3) linearity in the frames
    _, _, fr_sum, _ = adacof_fused(frame0 + const, frame2 + const, W[0], a[0], a[1], W[1], a[2], a[3], occ, 1,
                                   False, False)
    assert (fr_sum - (fr + 0.625)).abs().max().item() <= 2e-5
    # (4) one-hot centre tap with zero offsets is the identity; mask is 0 there
    onehot = torch.zeros_like(W[0]); onehot[:, 12] = 1.0
    zero = torch.zeros_like(a[0])
    i1, i2, _, m0 = adacof_fused(frame0, frame2, onehot, zero, zero, onehot, zero, zero, occ, 1)
    assert torch.equal(i1, frame0) and torch.equal(i2, frame2) and m0.abs().max().item() == 0.0
    # (5) mask range
    assert mask.min().item() >= 0.0 and mask.max().item() <= 1.0
    # (6) a sub-window of the 1080p launch agrees with the oracle run on that window's rows
    rows = slice(0, 8)
    o = adacof_cpu.adacof_forward_window(p0[:, :, 0:32].cpu().numpy(), W[0][:, :, rows].cpu().numpy(),
                                         a[0][:, :, rows].cpu().numpy(), a[1][:, :, rows].cpu().numpy(), 1)
    np.testing.assert_allclose(t1[:, :, rows].cpu().numpy(), o, rtol=0, atol=ATOL)


def test_shape_errors_mirror_reference_asserts(device):
    x = torch.zeros(1, 3, 9, 8, device=device)
    w = torch.zeros(1, 25, 4, 4, device=device)
    with pytest.raises(AssertionError):     # adacof.py:326
        FunctionAdaCoF.apply(x, w, w, w, 1)
    with pytest.raises(AssertionError):     # adacof.py:329 (non-contiguous)
        FunctionAdaCoF.apply(torch.zeros(1, 3, 8, 16, device=device)[:, :, :, ::2], w, w, w, 1)


def test_fused_rgbx_equals_planar(device):
    # the pixel-interleaved variant (16-B gathers) computes exactly the same sums
    from vfi_amd import ops
    c = _rand_case(21, 2, 40, 72, 5)
    d = {k: _dev(v, device) for k, v in c.items()}
    ref = adacof_fused(d["f0"], d["f2"], d["w1"], d["a1"], d["b1"], d["w2"], d["a2"], d["b2"], d["occ"], 1)
    inter = lambda f: torch.cat((f, torch.full_like(f[:, :1], float("nan"))), 1).permute(0, 2, 3, 1).contiguous()
    got = adacof_fused(inter(d["f0"]), inter(d["f2"]), d["w1"], d["a1"], d["b1"], d["w2"], d["a2"], d["b2"], d["occ"], 1,
                       rgbx=True)
    for a, b in zip(ref, got):
        assert (a - b).abs().max().item() <= 5e-6
    # softmax folded into the sampler: logits in, same outputs
    lg1 = torch.log(d["w1"]) + 3.0 + torch.randn(2, 1, 40, 72, device=device) * 20.0   # shifts cancel in softmax
    lg2 = torch.log(d["w2"]) - 50.0
    got2 = adacof_fused(inter(d["f0"]), inter(d["f2"]), lg1.contiguous(), d["a1"], d["b1"], lg2.contiguous(), d["a2"], d["b2"],
                        d["occ"], 1, rgbx=True, weights_are_logits=True)
    for a, b in zip(ref, got2):
        assert (a - b).abs().max().item() <= 2e-5
    # and the prologue writes that layout (reflect pad to /32 + mean-subtracted concat)
    f0, f2 = d["f0"][:, :, :, :70].contiguous(), d["f2"][:, :, :, :70].contiguous()
    p0, p2, x6 = ops.adacof_prepare(f0, f2, rgbx=True)
    q0, q2, y6 = ops.adacof_prepare(f0, f2, rgbx=False)
    assert p0.shape == (2, 64, 96, 4) and q0.shape == (2, 3, 64, 96) and torch.equal(x6, y6)
    assert torch.equal(p0[..., :3].permute(0, 3, 1, 2), q0) and torch.equal(p2[..., :3].permute(0, 3, 1, 2), q2)
    want = torch.nn.functional.pad(torch.nn.functional.pad(f0, (0, 0, 0, 24), mode="reflect"), (0, 26, 0, 0), mode="reflect")
    assert torch.equal(q0, want)
