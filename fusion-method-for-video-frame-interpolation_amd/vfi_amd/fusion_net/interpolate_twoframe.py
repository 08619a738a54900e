"""Per-frame fused interpolation -- counterpart of reference src/fusion_net/interpolate_twoframe.py.

`interp(args)` keeps the reference's file-in / file-out contract (interpolate_twoframe.py:82-334: PNG paths
in `args.first_frame` / `second_frame`, centre crop to `args.dim`, outputs written with torchvision's
`save_image` quantisation, optional `loaded_adacof_model` / `loaded_fusion_net`), so
src/evaluation/interpolate.py:63-90 can call it unchanged.  The arithmetic lives in `FusionInterpolator`,
which performs the sequence of interpolate_twoframe.py:148-330 entirely on the MI355X:

  rgb->Lab | AdaCoF #1 | pyramid(6 Lab images) -> PhaseNet -> inverse pyramid -> Lab->rgb |
  pyramid(ada_pred, rgb_pred) -> two band-limited reconstructions -> uncertainty maps (Gaussian, 50x50 median) |
  AdaCoF #2,#3 (one batch-2 call) and #4 | [baseline] | FusionNet

Differences from the reference's execution (not its results):
  * Pyramid, PhaseNet and their plans/weights are built once per frame size, not per frame (:124-137);
  * no host round trips (the reference leaves the GPU 5x per frame for skimage/scipy, :148-149,190,207-225);
  * wherever pyramid values are only moved, not modified, between analysis and synthesis (both `h_freq`
    reconstructions of the phase uncertainty, the `baseline` mix) the round trip is applied as ONE radial gain in
    the frequency domain; the ada-uncertainty pyramids only transform the levels that are kept (level masks);
  * AdaCoF #1, #2 and #3 are independent of each other and run as one batch of three (after the PhaseNet branch).
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from .. import ops
from ..adacof.models import Model
from ..phase_net.phase_net import PhaseNet
from ..train.pyramid import Pyramid
from ..train.utils import calc_pyr_height
from ..values import DecompValues
from .fusion_net import FusionNet

DEFAULT_ADACOF_MODEL = "vfi_amd.fusion_net.fusion_adacofnet"


def crop_center(img, cropx, cropy):
    """interpolate_twoframe.py:75-79."""
    y, x, _ = img.shape
    startx = x // 2 - (cropx // 2)
    starty = y // 2 - (cropy // 2)
    return img[starty:starty + cropy, startx:startx + cropx]


def imwrite(tensor, path, range=(0, 1)):
    """torchvision.utils.save_image for one (3,H,W) image (no normalisation unless asked)."""
    from PIL import Image
    arr = tensor.detach().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(path)


class FusionInterpolator:
    """Holds the three networks and the per-size pyramid state; `__call__(rgb1, rgb2)` -> dict of tensors."""

    def __init__(self, adacof_model, fusion_net, phase_net_state=None, device=None):
        self.adacof = adacof_model
        self.fusion_net = fusion_net
        self.device = torch.device(device) if device is not None else next(fusion_net.parameters()).device
        self.phase_net_state = phase_net_state
        self._per_size = {}
        # the fused path never uses the two sampled sides (:156,229-237): models whose forward takes the per-call
        # `return_sides` argument (the fusion variant) skip their stores
        import inspect
        net = getattr(adacof_model, "model", adacof_model)                      # the AdaCoFNet behind the Model wrapper
        self._adacof_kwargs = {"return_sides": False} if "return_sides" in inspect.signature(net.forward).parameters else {}

    def _state(self, h, w):
        key = (h, w)
        if key not in self._per_size:
            height = calc_pyr_height(torch.empty(3, h, w, device="meta"))
            pyr = Pyramid(height=height, nbands=4, scale_factor=np.sqrt(2), device=self.device)   # :124-129
            pyr.set_full_size(h, w)
            net = PhaseNet(pyr, self.device, num_img=2)                                           # :132-137
            if self.phase_net_state is not None:
                net.load_state_dict(self.phase_net_state)
            net.eval()
            self._per_size[key] = (pyr, net)
        return self._per_size[key]

    @torch.no_grad()
    def __call__(self, rgb_frame1, rgb_frame2, output_baseline=False):
        """rgb_frame1/2: (3,H,W) float32 in [0,1] on the device."""
        h, w = rgb_frame1.shape[1:]
        pyr, phase_net = self._state(h, w)
        nlev = pyr.height - 2
        # both Lab frames in ONE (6,H,W) buffer: it is the pyramid's input (:172) and FusionNet's `other` (:324-325) as is
        lab12 = torch.empty((6, h, w), dtype=torch.float32, device=rgb_frame1.device)
        ops.rgb2lab(rgb_frame1, out=lab12[:3])                                         # :148-149
        ops.rgb2lab(rgb_frame2, out=lab12[3:])
        f1, f2 = rgb_frame1.unsqueeze(0), rgb_frame2.unsqueeze(0)
        return self._run(rgb_frame1, rgb_frame2, output_baseline, pyr, phase_net, nlev, lab12, f1, f2)

    def _run(self, rgb_frame1, rgb_frame2, output_baseline, pyr, phase_net, nlev, lab12, f1, f2):
        h, w = rgb_frame1.shape[1:]
        # PhaseNet branch (:168-192)
        vals, bufs, amp_max = pyr.filter(lab12, concat_frames=2, phase_scale=1.0 / math.pi,
                                         amp_max_eps=phase_net.eps)       # (the per-level maxima of :55 come with the bands)
        vals_pred = phase_net(phase_net.normalize_vals(vals, concat=bufs, amp_max=amp_max))
        lab_pred = pyr.inv_filter(DecompValues(0, vals_pred.phase, vals_pred.amplitude, vals_pred.low_level))
        phase_pred = ops.lab2rgb(lab_pred)                                             # (3,H,W) rgb
        pp = phase_pred.unsqueeze(0)

        # AdaCoF #1 (rgb1, rgb2) (:156), #2 (rgb1, phase_pred) and #3 (phase_pred, rgb2) (:229-233) are independent of each
        # other: ONE batch of three samples (the deep, small U-Net levels fill the chip better)
        _, _, three, masks = self.adacof(torch.cat((f1, f1, pp), 0), torch.cat((f2, pp, f2), 0), **self._adacof_kwargs)
        ada_pred, flow_var_map, between = three[:1], masks[:1], three[1:]              # (1,3,H,W), (1,1,H,W), (2,3,H,W)

        # uncertainty maps (:198-225)
        coarse = min(6, nlev)
        # phase uncertainty (:205-214): h_freq - h_freq_ph = G * (mean_c(ada_pred) - mean_c(rgb_pred)) with G the radial
        # gain of "finest band level + high residual" -- both get_last_value_levels reconstructions are linear in
        # unmodified values, so 48 band FFTs collapse into one R2C / C2R pair on one image
        m = ops.channel_mean_diff(ada_pred, phase_pred.unsqueeze(0), 1.0, False, signed=True)      # (1,H,W)
        hf = pyr.band_filter(m, level_mask=1, keep_high=True)
        maps = torch.empty((1, 3, h, w), dtype=torch.float32, device=rgb_frame1.device)   # :326-327, filled in place below
        d = ops.absdiff(hf, None, 100.0, True)                                         # :210-211
        phase_uncertainty = ops.gaussian_filter(d, 5, out=maps[:, 1])                  # :212-214  (1,H,W)
        # ada uncertainty (:217-225): |phase|,|amp| differences are not linear -> transform the 6 coarsest levels
        mask = ((1 << coarse) - 1) << (nlev - coarse)
        vb = pyr.filter(torch.cat((ada_pred[0], phase_pred), 0), level_mask=mask, want_high=False)   # 6 images, RGB space
        half = lambda t: (t[:12], t[12:])                                              # ada planes | phase planes
        dp, da = [0] * nlev, [0] * nlev
        for k in range(nlev - coarse, nlev):
            dp[k] = ops.absdiff(*half(vb.phase[k])[::-1])                              # subtract_values(vals_ph, vals_ada)
            da[k] = ops.absdiff(*half(vb.amplitude[k])[::-1])
        dlow = ops.absdiff(vb.low_level[3:], vb.low_level[:3])
        freq = pyr.inv_filter(DecompValues(0, dp, da, dlow))                           # :217-219 (3,H,W)
        fd = ops.channel_mean_diff(freq.unsqueeze(0), None, 30.0, False)               # :220
        ada_uncertainty = ops.absdiff(fd, ops.median_filter(fd, 50), 5.0, True, out=maps[:, 0])   # :221-225 (1,H,W)

        # base (:234-238): AdaCoF #4 on the two intermediate results
        _, _, base, _ = self.adacof(between[:1], between[1:], **self._adacof_kwargs)

        out = {"phase_pred": pp, "ada_pred": ada_pred, "base": base, "flow_var_map": flow_var_map,
               "phase_uncertainty": phase_uncertainty, "ada_uncertainty": ada_uncertainty}
        if output_baseline:                                                            # :288-322
            # baseline = inv_filter(high + coarse half of filter(lab_ada), fine half + low of filter(lab_phase)): every
            # value is moved UNMODIFIED, so both analyses and the synthesis are linear radial filters (see
            # vfi_pyr_apply_filter_pair): G_a * lab_ada + G_p * lab_phase with one inverse transform
            split = nlev // 2
            fine = (1 << split) - 1
            lab = pyr.band_filter_pair(ops.rgb2lab(ada_pred[0]), dict(level_mask=((1 << nlev) - 1) & ~fine, keep_high=True),
                                       ops.rgb2lab(phase_pred), dict(level_mask=fine, keep_low=True))
            out["baseline"] = ops.lab2rgb(lab).unsqueeze(0)

        other = lab12.unsqueeze(0)                                                     # :324-325 (1,6,H,W)
        maps[:, 2].copy_(flow_var_map[:, 0])                                           # (the sampler's mask comes with a batch of three)
        out["final"] = self.fusion_net(base, ada_pred, pp, other, maps, variant=0)     # :330
        return out


def build_models(args, device):
    """The model construction of interpolate_twoframe.py:88-103,139-145 / src/evaluation/evaluate.py:225-242."""
    if getattr(args, "loaded_adacof_model", None) is not None:
        adacof_model = args.loaded_adacof_model
    else:
        adacof_model = Model(SimpleNamespace(gpu_id=args.gpu_id, model=getattr(args, "adacof_model", DEFAULT_ADACOF_MODEL),
                                             kernel_size=args.adacof_kernel_size, dilation=args.adacof_dilation,
                                             config=getattr(args, "adacof_config", None)))
        adacof_model.eval()
        ckpt = getattr(args, "adacof_checkpoint", None)
        if ckpt and os.path.exists(ckpt):
            adacof_model.load(torch.load(ckpt, map_location="cpu")["state_dict"])
    if getattr(args, "loaded_fusion_net", None) is not None:
        fusion_net = args.loaded_fusion_net
    else:
        fusion_net = FusionNet().to(device)
        fusion_net.load_state_dict(torch.load(args.checkpoint, map_location="cpu"))
        fusion_net.eval()
    return adacof_model, fusion_net


_INTERPOLATORS = {}      # key -> FusionInterpolator (holds its models), at most _MAX_INTERPOLATORS, least recently used first
_MAX_INTERPOLATORS = 4


def _interpolator_for(args, device):
    """One FusionInterpolator (models + per-size pyramid plans, ~1 GB of workspace at 1080p) per model identity:
    pre-loaded models (src/evaluation passes them, interpolate.py:78-79) are keyed by object; otherwise the key is the
    set of checkpoint paths, so repeated file-based calls build the models and plans ONCE instead of once per call
    (the reference rebuilds everything per frame, interpolate_twoframe.py:88-103,124-145).  Bounded LRU: an evicted
    entry releases its plans."""
    pn = getattr(args, "phase_net_checkpoint", "./src/phase_net/phase_net.pt")      # :134
    if getattr(args, "loaded_adacof_model", None) is not None and getattr(args, "loaded_fusion_net", None) is not None:
        key = ("loaded", id(args.loaded_adacof_model), id(args.loaded_fusion_net), pn, device.index)
    else:
        key = ("files", getattr(args, "adacof_model", DEFAULT_ADACOF_MODEL), getattr(args, "adacof_checkpoint", None),
               getattr(args, "checkpoint", None), args.adacof_kernel_size, args.adacof_dilation, pn, device.index,
               id(getattr(args, "loaded_adacof_model", None)), id(getattr(args, "loaded_fusion_net", None)))
    runner = _INTERPOLATORS.pop(key, None)
    if runner is None:
        adacof_model, fusion_net = build_models(args, device)
        state = torch.load(pn, map_location="cpu") if pn and os.path.exists(pn) else None
        runner = FusionInterpolator(adacof_model, fusion_net, state, device)     # (keeps the models alive: ids stay unique)
        while len(_INTERPOLATORS) >= _MAX_INTERPOLATORS:
            _INTERPOLATORS.pop(next(iter(_INTERPOLATORS)))
    _INTERPOLATORS[key] = runner                                                  # most recently used last
    return runner


def interp(args, high_level=False):
    """File-based entry point with the reference's argument namespace (interpolate_twoframe.py:82-334).
    `high_level` (and `args.high_level`) is accepted and ignored, as in the reference: its `interp` never reads either
    (the only `high_level` in its body is the DecompValues field, :289,307)."""
    from PIL import Image
    torch.cuda.set_device(args.gpu_id)
    device = torch.device("cuda:{}".format(args.gpu_id))
    runner = _interpolator_for(args, device)
    img1 = crop_center(np.array(Image.open(args.first_frame)), args.dim, args.dim)       # :106-113
    img2 = crop_center(np.array(Image.open(args.second_frame)), args.dim, args.dim)
    to_t = lambda a: torch.as_tensor(a[..., :3]).permute(2, 0, 1).float().to(device) / 255
    out = runner(to_t(img1), to_t(img2), output_baseline=getattr(args, "output_baseline", False))
    if getattr(args, "output_phase", False):
        imwrite(out["phase_pred"][0], args.output_frame_phase)
    if getattr(args, "output_adacof", False):
        imwrite(out["ada_pred"][0], args.output_frame_adacof)
    if getattr(args, "output_baseline", False):
        imwrite(out["baseline"][0], args.output_frame_baseline)
    imwrite(out["final"][0], args.output_frame)
    return out
