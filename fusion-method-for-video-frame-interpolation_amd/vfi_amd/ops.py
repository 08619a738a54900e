"""Host-side wrappers of the generic device ops of libvfi_hip.so (include/vfi_hip.h).

Tensors are torch HIP tensors used purely as device-memory handles; every wrapper enqueues one
library call on torch's current stream.  Inputs / outputs may be CHANNEL SLICES of wider NCHW
tensors (`t[:, a:b]`): only the batch stride is free, the (C, H, W) block must be dense.
"""
import torch

from . import _lib
from ._lib import VfiLibraryError

ACT = {None: 0, "none": 0, "relu": 1, "elu": 2, "tanh": 3, "sigmoid": 4}
PAD = {"zeros": 0, "zero": 0, "reflect": 1}


def _slice_ptr(t, name):
    """(device address, batch stride) of an NCHW tensor whose per-sample (C,H,W) block is dense."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise VfiLibraryError(f"{name} must be a tensor on a HIP device (vfi_amd has no CPU path)")
    if t.dtype != torch.float32 or t.dim() != 4:
        raise VfiLibraryError(f"{name} must be a 4-d float32 tensor")
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if not ((sw == 1 or w == 1) and (sh == w or h == 1) and (sc == h * w or c == 1)):
        raise VfiLibraryError(f"{name}: per-sample (C,H,W) block must be dense, strides {t.stride()}")
    if n == 1:
        sn = c * h * w
    return t.data_ptr(), sn


def new(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class PackedConv:
    """Weights of one nn.Conv2d in the matrix-core layout ([Cin_pad][KS*KS][Cout_pad]) plus bias.

    `bn` = (weight, bias, running_mean, running_var, eps) folds an eval-mode BatchNorm2d that
    follows the conv (reference src/phase_net/phase_net.py:191-193) into weights and bias."""

    def __init__(self, weight, bias=None, bn=None, device=None):
        device = torch.device(device) if device is not None else weight.device
        if device.type != "cuda":
            raise VfiLibraryError("PackedConv needs a HIP device (vfi_amd has no CPU path)")
        weight = weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self.cout, self.cin, kh, kw = weight.shape
        assert kh == kw, "square kernels only"
        self.ks = kh
        cout = self.cout
        b = (bias.detach().to(device=device, dtype=torch.float32) if bias is not None
             else torch.zeros(cout, device=device))
        scale = None
        if bn is not None:  # y = (conv + b - mean) * g / sqrt(var + eps) + beta
            g, beta, mean, var, eps = bn
            f = lambda t: t.detach().to(device=device, dtype=torch.float32)
            scale = (f(g) / torch.sqrt(f(var) + eps)).contiguous()
            b = (b - f(mean)) * scale + f(beta)
        self.bias = b.contiguous()
        n = _lib.lib().vfi_conv2d_packed_floats(cout, self.cin, self.ks)
        if n <= 0:
            raise VfiLibraryError(f"vfi_conv2d_packed_floats rejected {tuple(weight.shape)}")
        self.packed = torch.empty(n, dtype=torch.float32, device=device)
        _lib.call("vfi_conv2d_pack", weight.data_ptr(), scale.data_ptr() if scale is not None else None,
                  self.packed.data_ptr(), cout, self.cin, self.ks, _lib.stream_ptr())
        self._keep = (weight, scale)  # alive until the pack kernel has run (same stream ordering)


def conv2d(x, pc, pad_mode="zeros", act=None, residual=None, out=None):
    """act(conv(x) + bias) (+ residual) -> out.  One vfi_conv2d launch."""
    n, cin, h, w = x.shape
    if cin != pc.cin:
        raise VfiLibraryError(f"conv2d: input has {cin} channels, weights expect {pc.cin}")
    if out is None:
        out = new((n, pc.cout, h, w), x)
    elif tuple(out.shape) != (n, pc.cout, h, w):
        raise VfiLibraryError(f"conv2d: out shape {tuple(out.shape)} != {(n, pc.cout, h, w)}")
    xp, xs = _slice_ptr(x, "x")
    yp, ys = _slice_ptr(out, "out")
    rp, rs = (None, 0)
    if residual is not None:
        if tuple(residual.shape) != tuple(out.shape):
            raise VfiLibraryError("conv2d: residual shape mismatch")
        rp, rs = _slice_ptr(residual, "residual")
    _lib.call("vfi_conv2d", xp, xs, pc.packed.data_ptr(), pc.bias.data_ptr(), rp, rs, yp, ys,
              n, cin, h, w, pc.cout, pc.ks, PAD[pad_mode], ACT[act], _lib.stream_ptr())
    return out
