"""AdaCoF deformable sampling op -- mirror of reference src/adacof/cupy_module/adacof.py.

``FunctionAdaCoF.apply(input, weight, offset_i, offset_j, dilation)`` keeps the reference
signature (adacof.py:313-315) and its asserts (adacof.py:326-332); the work is one launch of
``vfi_adacof_forward`` (csrc/vfi_adacof.hip) on torch's current HIP stream instead of a
per-shape NVRTC compile.  Inference only: the reference's three backward kernels
(adacof.py:67-258) are training code and out of scope, so ``backward`` raises.
"""
import math

import torch

from ... import _lib


class FunctionAdaCoF(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weight, offset_i, offset_j, dilation):
        ctx.dilation = dilation
        n, c, hin, win = input.shape
        f = int(math.sqrt(weight.size(1)))
        h, w = weight.size(2), weight.size(3)
        assert hin - ((f - 1) * dilation + 1) == h - 1   # adacof.py:326
        assert win - ((f - 1) * dilation + 1) == w - 1   # adacof.py:327
        assert input.is_contiguous()                      # adacof.py:329-332
        assert weight.is_contiguous()
        assert offset_i.is_contiguous()
        assert offset_j.is_contiguous()
        if not input.is_cuda:
            raise NotImplementedError()                   # adacof.py:356-357
        output = torch.empty((n, c, h, w), dtype=input.dtype, device=input.device)
        _lib.call("vfi_adacof_forward", _lib.dptr(input, "input"), _lib.dptr(weight, "weight"),
                  _lib.dptr(offset_i, "offset_i"), _lib.dptr(offset_j, "offset_j"),
                  _lib.dptr(output), n, c, hin, win, h, w, f, int(dilation), _lib.stream_ptr())
        return output

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError("vfi_amd implements the inference path only")


def adacof_fused(frame0, frame2, w1, a1, b1, w2, a2, b2, occ, dilation,
                 want_sides=True, want_mask=True, rgbx=False, weights_are_logits=False):
    """Both sampling sides + occlusion blend + flow-variance mask in one launch
    (reference src/fusion_net/fusion_adacofnet.py:195-213).  Frames are UN-padded: planar (N,3,H,W), or
    pixel-interleaved (N,H,W,4) with rgbx=True (as ops.adacof_prepare writes them)."""
    if rgbx:
        n, h, w, c = frame0.shape[0], frame0.shape[1], frame0.shape[2], 3
    else:
        n, c, h, w = frame0.shape
    f = int(math.sqrt(w1.size(1)))
    new = lambda ch: torch.empty((n, ch, h, w), dtype=torch.float32, device=frame0.device)
    t1 = new(c) if want_sides else None
    t2 = new(c) if want_sides else None
    frame = new(c)
    mask = new(1) if want_mask else None
    d = _lib.dptr
    head = ("vfi_adacof_fused_rgbx",) if rgbx else ("vfi_adacof_fused",)
    dims = (n, h, w) if rgbx else (n, c, h, w)
    if weights_are_logits and not rgbx:
        raise _lib.VfiLibraryError("weights_are_logits needs rgbx frames")
    extra = (int(bool(weights_are_logits)),) if rgbx else ()
    _lib.call(*head, d(frame0, "frame0"), d(frame2, "frame2"), d(w1), d(a1), d(b1),
              d(w2), d(a2), d(b2), d(occ), d(t1), d(t2), d(frame), d(mask),
              *dims, f, int(dilation), *extra, _lib.stream_ptr(),
              work=("byte", float(n) * h * w * (6 * f * f * 4 + 4 + 2 * 4 * c + 4 * c * (3 if want_sides else 1)
                                                 + (4 if want_mask else 0)), "adacof_fused_kernel"))
    return t1, t2, frame, mask
