"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain PyTorch-CPU fp32 restatements of the three networks on the hot path, written as
functions over a state dict that uses the REFERENCE's checkpoint key names, so the
reference's own checkpoints (phase_net.pt, fusion_net.pt, AdaCoF ckpt.pth) load unchanged.

  phasenet_*          <- reference src/phase_net/phase_net.py:42-207
  kernel_estimation   <- reference src/fusion_net/fusion_adacofnet.py:109-155 (same net as
                         src/adacof/models/adacofnet.py:107-153)
  adacofnet_forward   <- reference src/fusion_net/fusion_adacofnet.py:172-240
  fusionnet_forward   <- reference src/fusion_net/fusion_net.py:46-77

Pinned by tests/golden/{phasenet,kernel_estimation,adacofnet,fusionnet}_*.npz (outputs of the
reference's own classes loaded with the seeded weights below; tests/golden/make_golden.py).
"""
import math
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

from . import adacof_cpu

# Field order as reference src/train/pyramid.py:12-18 (positional use matters).
DecompValues = namedtuple("values", "high_level, phase, amplitude, low_level")

EPS = 1e-8  # phase_net.py:19


# --------------------------------------------------------------------------------------------
# seeded weights with the reference's key names / shapes (numpy PCG64: platform independent)
# --------------------------------------------------------------------------------------------
def _fill(shapes, seed, bn_keys=()):
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape in shapes:
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.long)
            continue
        if key.endswith("running_var"):
            arr = rng.uniform(0.5, 1.5, shape)
        elif key.endswith("running_mean"):
            arr = rng.uniform(-0.2, 0.2, shape)
        elif len(shape) == 4:                      # conv weight: U(+-1/sqrt(fan_in)), torch's default scale
            bound = 1.0 / math.sqrt(shape[1] * shape[2] * shape[3])
            arr = rng.uniform(-bound, bound, shape)
        elif any(key.startswith(b) for b in bn_keys) and key.endswith("weight"):
            arr = rng.uniform(0.8, 1.2, shape)
        else:                                      # biases
            arr = rng.uniform(-0.05, 0.05, shape)
        sd[key] = torch.from_numpy(arr.astype(np.float32))
    return sd


def phasenet_shapes(num_img=2):
    """phase_net.py:21-35 / :190-200: 8 blocks; block 0 sees the low level, 1-2 are 1x1, 3-7 are 3x3."""
    cin = [num_img, 64 + 1 + 8 * num_img] + [64 + 8 + 8 * num_img] * 6
    pred = [1] + [8] * 7
    ks = [1, 1, 1, 3, 3, 3, 3, 3]
    out = []
    for i in range(8):
        p = f"layers.{i}."
        out += [(p + "feature_map.0.weight", (64, cin[i], ks[i], ks[i])), (p + "feature_map.0.bias", (64,)),
                (p + "feature_map.1.weight", (64,)), (p + "feature_map.1.bias", (64,)),
                (p + "feature_map.1.running_mean", (64,)), (p + "feature_map.1.running_var", (64,)),
                (p + "feature_map.1.num_batches_tracked", ()),
                (p + "feature_map.3.weight", (64, 64, ks[i], ks[i])), (p + "feature_map.3.bias", (64,)),
                (p + "prediction_map.0.weight", (pred[i], 64, 1, 1)), (p + "prediction_map.0.bias", (pred[i],))]
    return out


def phasenet_random_state_dict(seed=0):
    return _fill(phasenet_shapes(), seed, bn_keys=tuple(f"layers.{i}.feature_map.1" for i in range(8)))


def kernel_estimation_shapes(kernel_size=5, prefix="get_kernel."):
    """fusion_adacofnet.py:14-107."""
    k2 = kernel_size ** 2
    out = []

    def conv(name, cin, cout):
        out.append((prefix + name + ".weight", (cout, cin, 3, 3)))
        out.append((prefix + name + ".bias", (cout,)))

    def basic(name, cin, cout):
        conv(f"{name}.0", cin, cout); conv(f"{name}.2", cout, cout); conv(f"{name}.4", cout, cout)

    basic("moduleConv1", 6, 32); basic("moduleConv2", 32, 64); basic("moduleConv3", 64, 128)
    basic("moduleConv4", 128, 256); basic("moduleConv5", 256, 512)
    basic("moduleDeconv5", 512, 512); conv("moduleUpsample5.1", 512, 512)
    basic("moduleDeconv4", 512, 256); conv("moduleUpsample4.1", 256, 256)
    basic("moduleDeconv3", 256, 128); conv("moduleUpsample3.1", 128, 128)
    basic("moduleDeconv2", 128, 64); conv("moduleUpsample2.1", 64, 64)
    for head in ("moduleWeight1", "moduleAlpha1", "moduleBeta1", "moduleWeight2", "moduleAlpha2", "moduleBeta2"):
        conv(f"{head}.0", 64, 64); conv(f"{head}.2", 64, 64); conv(f"{head}.4", 64, k2); conv(f"{head}.7", k2, k2)
    conv("moduleOcclusion.0", 64, 64); conv("moduleOcclusion.2", 64, 64); conv("moduleOcclusion.4", 64, 64)
    conv("moduleOcclusion.7", 64, 1)
    return out


def adacofnet_random_state_dict(seed=0, kernel_size=5):
    return _fill(kernel_estimation_shapes(kernel_size), seed)


def fusionnet_shapes(num_imgs=5, uncertainty_maps=3):
    """fusion_net.py:8-43, including the dead `net.*` stack that sits in the checkpoints."""
    cin = 3 * num_imgs + uncertainty_maps
    out = []

    def conv(name, ci, co, k):
        out.append((name + ".weight", (co, ci, k, k))); out.append((name + ".bias", (co,)))

    conv("net.0", cin, 64, 3); conv("net.2", 64, 64, 3); conv("net.4", 64, 64, 3); conv("net.6", 64, 3, 3)
    conv("encoder_layers.0", cin, 32, 5); conv("encoder_layers.1", 32, 64, 5); conv("encoder_layers.2", 64, 128, 3)
    conv("bottleneck_layer", 128, 128, 3)
    conv("decoder_layers.0", 128, 64, 5); conv("decoder_layers.1", 64, 32, 5); conv("decoder_layers.2", 32, 3, 1)
    return out


def fusionnet_random_state_dict(seed=0, num_imgs=5, uncertainty_maps=3):
    return _fill(fusionnet_shapes(num_imgs, uncertainty_maps), seed)


# --------------------------------------------------------------------------------------------
# PhaseNet  (phase_net.py)
# --------------------------------------------------------------------------------------------
def _conv(sd, name, x, pad=0, mode="zeros"):
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    if pad and mode == "reflect":
        x = F.pad(x, (pad,) * 4, mode="reflect")
        pad = 0
    return F.conv2d(x, w, b, padding=pad)


def phasenet_block(sd, i, x):
    """phase_net.py:190-207: conv-BN(eval)-ELU-conv-ELU -> feature; 1x1 conv + tanh -> prediction."""
    p = f"layers.{i}."
    pad = (sd[p + "feature_map.0.weight"].shape[-1] - 1) // 2
    f = _conv(sd, p + "feature_map.0", x, pad, "reflect")
    f = F.batch_norm(f, sd[p + "feature_map.1.running_mean"], sd[p + "feature_map.1.running_var"],
                     sd[p + "feature_map.1.weight"], sd[p + "feature_map.1.bias"], training=False, eps=1e-5)
    f = F.elu(f)
    f = F.elu(_conv(sd, p + "feature_map.3", f, pad, "reflect"))
    c = torch.tanh(_conv(sd, p + "prediction_map.0", f))
    return f, c


def phasenet_normalize(vals):
    """phase_net.py:42-78.  Returns (normalised values, (max_amplitudes, max_low_level))."""
    b = vals.amplitude[0].shape[0]
    amps, maxes = [], []
    for a in vals.amplitude:
        mx = a.reshape(b, -1).max(1)[0] + EPS                     # :55
        maxes.append(mx)
        amps.append(a / mx.view(b, 1, 1, 1))                      # :61
    phases = [p / math.pi for p in vals.phase]                    # :64
    max_low = vals.low_level.reshape(b, -1).max(1)[0] + EPS       # :69
    low = vals.low_level / max_low.view(b, 1, 1, 1)               # :70
    return DecompValues(vals.high_level, phases, amps, low), (maxes, max_low)


def phasenet_forward(sd, vals, norm_state, height, nbands=4, m=None):
    """phase_net.py:107-177 (+ reverse_normalize :80-105), num_img == 2."""
    maxes, max_low = norm_state
    if m is None:
        m = height - 2
    feature, pred = phasenet_block(sd, 0, vals.low_level)                       # :113
    alpha = (pred[:, 0] + 1) / 2                                                # :115
    low = (alpha * vals.low_level[:, 0] + (1 - alpha) * vals.low_level[:, 1]).unsqueeze(1)
    hs = vals.high_level.shape
    high = torch.zeros((hs[0], 1, hs[2], hs[3]))                                # :127-128
    phases, amps = [], []
    for idx in range(m):
        size = vals.phase[idx].shape[2:]
        fr = F.interpolate(feature, size=tuple(size), mode="bilinear", align_corners=False)   # :138
        pr = F.interpolate(pred, size=tuple(size), mode="bilinear", align_corners=False)      # :139
        x = torch.cat((fr, vals.phase[idx], vals.amplitude[idx], pr), 1)        # :141
        i = idx + 1 if idx + 1 < 7 else 7                                       # :148
        feature, pred = phasenet_block(sd, i, x)
        beta = (pred[:, 4:8] + 1) / 2                                           # :155
        amp = beta * vals.amplitude[idx][:, 4:8] + (1 - beta) * vals.amplitude[idx][:, :4]
        h, w = pred.shape[2:]
        phases.append(pred[:, :4].reshape(-1, 1, h, w))                         # :167
        amps.append(amp.reshape(-1, 1, h, w))                                   # :168
    # reverse_normalize (:80-105)
    out_p = [p * math.pi for p in phases]
    out_a = []
    for i in range(m):
        shp = amps[i].shape
        bsz = shp[0] // nbands
        out_a.append((amps[i].reshape(bsz, -1) * maxes[i].view(bsz, 1)).reshape(shp))
    for _ in range(height - 2 - m):                                             # :91-93
        out_p.append(0)
        out_a.append(0)
    low = low * max_low.view(-1, 1, 1, 1)
    return DecompValues(high, out_p[::-1], out_a[::-1], low)


# --------------------------------------------------------------------------------------------
# AdaCoF network  (fusion_adacofnet.py)
# --------------------------------------------------------------------------------------------
def _basic(sd, name, x):
    for i in (0, 2, 4):
        x = F.relu(_conv(sd, f"{name}.{i}", x, 1))
    return x


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def _upsample(sd, name, x):
    return F.relu(_conv(sd, f"{name}.1", _up2(x), 1))


def _head(sd, name, x):
    for i in (0, 2, 4):
        x = F.relu(_conv(sd, f"{name}.{i}", x, 1))
    return _conv(sd, f"{name}.7", _up2(x), 1)


def kernel_estimation(sd, f0, f2, prefix="get_kernel."):
    """fusion_adacofnet.py:109-155 -> (W1, A1, B1, W2, A2, B2, Occ)."""
    g = lambda n: prefix + n
    c1 = _basic(sd, g("moduleConv1"), torch.cat([f0, f2], 1))
    c2 = _basic(sd, g("moduleConv2"), F.avg_pool2d(c1, 2, 2))
    c3 = _basic(sd, g("moduleConv3"), F.avg_pool2d(c2, 2, 2))
    c4 = _basic(sd, g("moduleConv4"), F.avg_pool2d(c3, 2, 2))
    c5 = _basic(sd, g("moduleConv5"), F.avg_pool2d(c4, 2, 2))
    x = _basic(sd, g("moduleDeconv5"), F.avg_pool2d(c5, 2, 2))
    x = _upsample(sd, g("moduleUpsample5"), x) + c5
    x = _upsample(sd, g("moduleUpsample4"), _basic(sd, g("moduleDeconv4"), x)) + c4
    x = _upsample(sd, g("moduleUpsample3"), _basic(sd, g("moduleDeconv3"), x)) + c3
    x = _upsample(sd, g("moduleUpsample2"), _basic(sd, g("moduleDeconv2"), x)) + c2
    w1 = torch.softmax(_head(sd, g("moduleWeight1"), x), 1)
    a1 = _head(sd, g("moduleAlpha1"), x)
    b1 = _head(sd, g("moduleBeta1"), x)
    w2 = torch.softmax(_head(sd, g("moduleWeight2"), x), 1)
    a2 = _head(sd, g("moduleAlpha2"), x)
    b2 = _head(sd, g("moduleBeta2"), x)
    occ = torch.sigmoid(_head(sd, g("moduleOcclusion"), x))
    return w1, a1, b1, w2, a2, b2, occ


CHANNEL_MEANS = (0.4631, 0.4352, 0.3990)  # src/adacof/utility.py:86-87


def module_normalize(frame):
    return frame - torch.tensor(CHANNEL_MEANS).view(1, 3, 1, 1)


def adacofnet_forward(sd, frame0, frame2, kernel_size=5, dilation=1, faithful_crop_bug=False):
    """fusion_adacofnet.py:172-240 -> (t1, t2, frame1, UncertaintyMask).

    `faithful_crop_bug`: line :225 of the reference assigns tensorAdaCoF1 from tensorAdaCoF2 when
    the width was padded; t1/t2 are unused downstream.  False returns the evident intent."""
    h0, w0 = frame0.shape[2:]
    if frame0.shape[2:] != frame2.shape[2:]:
        raise SystemExit("Frame sizes do not match")               # :175-176
    ph = (32 - h0 % 32) % 32
    pw = (32 - w0 % 32) % 32
    if ph:
        frame0 = F.pad(frame0, (0, 0, 0, ph), mode="reflect"); frame2 = F.pad(frame2, (0, 0, 0, ph), mode="reflect")
    if pw:
        frame0 = F.pad(frame0, (0, pw, 0, 0), mode="reflect"); frame2 = F.pad(frame2, (0, pw, 0, 0), mode="reflect")
    w1, a1, b1, w2, a2, b2, occ = kernel_estimation(sd, module_normalize(frame0), module_normalize(frame2))
    pad = int(((kernel_size - 1) * dilation) / 2.0)                # :161
    rp = lambda x: F.pad(x, (pad,) * 4, mode="replicate").numpy()
    n = lambda t: t.contiguous().numpy()
    t1 = adacof_cpu.adacof_forward(rp(frame0), n(w1), n(a1), n(b1), dilation)   # :195
    t2 = adacof_cpu.adacof_forward(rp(frame2), n(w2), n(a2), n(b2), dilation)   # :196
    frame1, mask = adacof_cpu.blend_mask(t1, t2, n(occ), n(w1), n(a1), n(b1), n(w2), n(a2), n(b2))
    t1, t2, frame1, mask = (torch.from_numpy(x) for x in (t1, t2, frame1, mask))
    if ph:
        t1, t2, frame1, mask = (x[:, :, :h0] for x in (t1, t2, frame1, mask))
    if pw:
        if faithful_crop_bug:
            t1 = t2
        t1, t2, frame1, mask = (x[:, :, :, :w0] for x in (t1, t2, frame1, mask))
    return t1, t2, frame1, mask


# --------------------------------------------------------------------------------------------
# FusionNet  (fusion_net.py)
# --------------------------------------------------------------------------------------------
def fusionnet_forward(sd, base, adacof, phase, other, maps, variant=0):
    """fusion_net.py:46-77."""
    parts = [base, adacof, phase, other] + ([maps] if maps is not None else [])
    x = torch.cat(parts, 1)
    skip = []
    for i in range(3):
        k = sd[f"encoder_layers.{i}.weight"].shape[-1]
        x = F.relu(_conv(sd, f"encoder_layers.{i}", x, (k - 1) // 2, "reflect"))
        skip.append(x)
        x = F.max_pool2d(x, 2, 2)
    x = _conv(sd, "bottleneck_layer", x, 1, "reflect")
    for i, s in enumerate(skip[::-1]):
        x = F.interpolate(F.relu(x), scale_factor=2, mode="bilinear", align_corners=False)
        x = x + s
        k = sd[f"decoder_layers.{i}.weight"].shape[-1]
        x = _conv(sd, f"decoder_layers.{i}", x, (k - 1) // 2, "reflect")
    res = torch.tanh(x)
    out = (phase if variant == 1 else base) + res
    return out.clamp(0, 1)
