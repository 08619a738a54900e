"""Host-side wrappers of the generic device ops of libvfi_hip.so (include/vfi_hip.h).

Tensors are torch HIP tensors used purely as device-memory handles; every wrapper enqueues one
library call on torch's current stream.  Inputs / outputs may be CHANNEL SLICES of wider NCHW
tensors (`t[:, a:b]`): only the batch stride is free, the (C, H, W) block must be dense.
"""
import os

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
    _lib.check_device(t, name)
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


WINOGRAD = os.environ.get("VFI_CONV_WINOGRAD", "1") != "0"     # mirrors the library's switch (profiling labels only)
# 3x3 convs whose input is a bilinear resize: materialise the resize and use the Winograd kernel (default), or keep the
# direct kernels with the interpolating tile loader (VFI_CONV_FUSED_RESIZE=1; always when Winograd is off)
FUSED_RESIZE = (not WINOGRAD) or os.environ.get("VFI_CONV_FUSED_RESIZE", "0") == "1"
# (measured again in round 2 for the thin 25 -> 25 head convolutions alone: the fused loader loses there too,
# 77.5 vs 72.2 ms per frame)
def _winograd_work(n, cin, cout, h, w, residual=False, pooled=False, act="relu"):
    """Profiling label and the executed algorithm's own flop count of a 3x3 layer.  Which kernel runs is the library's
    decision (vfi_conv2d_algo): F(4x4,3x3) -- 36 multiply-adds per 4x4 outputs and channel pair -- or F(2x2,3x3) -- 16 per
    2x2 outputs."""
    algo = _lib.lib().vfi_conv2d_algo(n, cin, h, w, cout, 3, int(bool(residual)), int(bool(pooled)), ACT[act])
    if algo == 2:       # (the M = 32 kernel unless VFI_CONV_WINOGRAD4M=0: the library reads that switch at every call)
        name = "conv3x3_winograd4_kernel" if os.environ.get("VFI_CONV_WINOGRAD4M", "1") == "0" else "conv3x3_winograd4m_kernel"
        return ("flop", 2.0 * n * cin * cout * 36 * (h * w / 16.0), name)
    if algo == 1:
        return ("flop", 2.0 * n * cin * cout * 16 * (h * w / 4.0), "conv3x3_winograd_kernel")
    return ("flop", 2.0 * n * cin * cout * 9 * h * w, "conv2d_mfma_kernel<3,8,%d>" % (2 if ((cout + 31) // 32 * 32) % 64 == 0 else 1))


_WORKSPACES = {}
WORKSPACE_FLOATS = 48 * 1024 * 1024     # 192 MiB per (device, stream): split-K partial sums of the deep U-Net levels


def _workspace(device):
    """Split-K scratch, one per (device, stream) so frames in flight on different streams never share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = torch.empty(WORKSPACE_FLOATS, dtype=torch.float32, device=device)
    return ws


def conv2d(x, pc, pad_mode="zeros", act=None, residual=None, out=None, upsample2x=False):
    """act(conv(x) + bias) (+ residual) -> out.  One vfi_conv2d launch.  upsample2x: x is the low-resolution
    input of an `Upsample(x2, bilinear, align_corners=True) -> conv` pair (the upsampled tensor is not
    materialised)."""
    n, cin, h, w = x.shape
    if cin != pc.cin:
        raise VfiLibraryError(f"conv2d: input has {cin} channels, weights expect {pc.cin}")
    if upsample2x and pc.ks == 3 and not FUSED_RESIZE:
        # the Winograd kernel reads its input by LDS-DMA and cannot interpolate on the fly: one streaming resize pass
        # (HBM-bound, a few % of the conv) + the Winograd conv beats the direct kernel with the fused loader
        x = resize_bilinear(x, (2 * h, 2 * w), align_corners=True)
        h, w, upsample2x = 2 * h, 2 * w, False
    if upsample2x:
        h, w = 2 * h, 2 * w
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
    ws = _workspace(x.device)
    work = None
    if _lib.PROFILE is not None:
        if pc.ks == 3 and not upsample2x and WINOGRAD:
            work = _winograd_work(n, cin, pc.cout, h, w, residual is not None, False, act)
        else:
            label = (f"conv2d_mfma_kernel<{pc.ks},{4 if pc.ks == 5 else 8},{2 if ((pc.cout + 31) // 32 * 32) % 64 == 0 else 1}"
                     + (",ups>" if upsample2x else ">"))
            work = ("flop", 2.0 * n * cin * pc.cout * pc.ks * pc.ks * h * w, label)
            if pc.ks == 1:
                # a 1x1 layer (PhaseNet's 64 -> 8 prediction maps, the coarse 1x1 blocks, FusionNet's 32 -> 3 tail) does
                # 2*Cout flop per input byte: it is bound by reading the input once, not by the matrix cores
                if _lib.lib().vfi_conv2d_algo(n, cin, h, w, pc.cout, 1, int(residual is not None), 0, ACT[act]) == 3:
                    label = "conv1x1_stream_kernel"
                work = ("byte", 4.0 * n * (cin + pc.cout) * h * w, label)
    _lib.call("vfi_conv2d_upsample2x" if upsample2x else "vfi_conv2d", xp, xs, pc.packed.data_ptr(), pc.bias.data_ptr(), rp, rs, yp, ys,
              n, cin, h, w, pc.cout, pc.ks, PAD[pad_mode], ACT[act], ws.data_ptr(), ws.numel(), _lib.stream_ptr(), work=work)
    return out


def conv2d_pool2(x, pc, is_max, pad_mode="zeros", act="relu"):
    """(act(conv(x) + bias), pool2 of it): one vfi_conv2d_pool2 call (the pooled tensor comes out of the conv's epilogue
    for 3x3 ReLU layers)."""
    n, cin, h, w = x.shape
    if cin != pc.cin:
        raise VfiLibraryError(f"conv2d_pool2: input has {cin} channels, weights expect {pc.cin}")
    out, pooled = new((n, pc.cout, h, w), x), new((n, pc.cout, h // 2, w // 2), x)
    xp, xs = _slice_ptr(x, "x")
    ws = _workspace(x.device)
    work = None
    if _lib.PROFILE is not None:
        if pc.ks == 3 and WINOGRAD:
            work = _winograd_work(n, cin, pc.cout, h, w, False, True, act)
        else:
            work = ("flop", 2.0 * n * cin * pc.cout * pc.ks * pc.ks * h * w,
                    f"conv2d_mfma_kernel<{pc.ks},{4 if pc.ks == 5 else 8},{2 if ((pc.cout + 31) // 32 * 32) % 64 == 0 else 1}>")
    _lib.call("vfi_conv2d_pool2", xp, xs, pc.packed.data_ptr(), pc.bias.data_ptr(), out.data_ptr(), out.stride(0),
              pooled.data_ptr(), pooled.stride(0), int(bool(is_max)), n, cin, h, w, pc.cout, pc.ks, PAD[pad_mode], ACT[act],
              ws.data_ptr(), ws.numel(), _lib.stream_ptr(), work=work)
    return out, pooled


def conv2d_resized_prefix(x, x2, pc, pad_mode="reflect", act=None, out=None):
    """conv over [bilinear_resize(x2, align_corners=False) | x[:, C2:]] with C2 = x2.shape[1]; x is (N, Cin, H, W) whose
    first C2 channels are ignored (never written).  One launch, the resized maps are not materialised."""
    n, cin, h, w = x.shape
    n2, c2, hs, ws_ = x2.shape
    if cin != pc.cin or n2 != n:
        raise VfiLibraryError("conv2d_resized_prefix: shape mismatch")
    if pc.ks == 3 and not FUSED_RESIZE:      # see conv2d(upsample2x=True)
        resize_bilinear(x2, (h, w), align_corners=False, out=x[:, :c2])
        return conv2d(x, pc, pad_mode=pad_mode, act=act, out=out)
    if out is None:
        out = new((n, pc.cout, h, w), x)
    xp, xs = _slice_ptr(x, "x")
    x2p, x2s = _slice_ptr(x2, "x2")
    yp, ys = _slice_ptr(out, "out")
    ws = _workspace(x.device)
    work = None
    if _lib.PROFILE is not None:
        work = ("flop", 2.0 * n * cin * pc.cout * pc.ks * pc.ks * h * w, f"conv2d_mfma_kernel<{pc.ks},8,2,rsz>")
    _lib.call("vfi_conv2d_resized_prefix", xp, xs, x2p, x2s, c2, hs, ws_, pc.packed.data_ptr(), pc.bias.data_ptr(), yp, ys,
              n, cin, h, w, pc.cout, pc.ks, PAD[pad_mode], ACT[act], ws.data_ptr(), ws.numel(), _lib.stream_ptr(), work=work)
    return out


def adacof_prepare(frame0, frame2, rgbx=True):
    """-> (pad0, pad2, x6 (N,6,Hp,Wp)); Hp, Wp = sizes rounded up to multiples of 32.  pad0/pad2 are the
    reflect-padded raw frames: pixel-interleaved (N,Hp,Wp,4) when rgbx, else planar (N,3,Hp,Wp)."""
    n, c, h, w = frame0.shape
    if c != 3 or tuple(frame2.shape) != tuple(frame0.shape):
        raise VfiLibraryError("adacof_prepare: frames must both be (N,3,H,W)")
    hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    shape = (n, hp, wp, 4) if rgbx else (n, 3, hp, wp)
    pad0, pad2, x6 = new(shape, frame0), new(shape, frame0), new((n, 6, hp, wp), frame0)
    _lib.call("vfi_adacof_prepare", _lib.dptr(frame0, "frame0"), _lib.dptr(frame2, "frame2"), pad0.data_ptr(),
              pad2.data_ptr(), x6.data_ptr(), n, h, w, hp, wp, int(bool(rgbx)), _lib.stream_ptr())
    return pad0, pad2, x6


def pool2(x, is_max, out=None):
    n, c, h, w = x.shape
    if out is None:
        out = new((n, c, h // 2, w // 2), x)
    xp, xs = _slice_ptr(x, "x")
    yp, ys = _slice_ptr(out, "out")
    _lib.call("vfi_pool2", xp, xs, yp, ys, n, c, h, w, int(bool(is_max)), _lib.stream_ptr())
    return out


def resize_bilinear(x, size, align_corners, relu_input=False, residual=None, out=None):
    n, c, h, w = x.shape
    ho, wo = size
    if out is None:
        out = new((n, c, ho, wo), x)
    xp, xs = _slice_ptr(x, "x")
    yp, ys = _slice_ptr(out, "out")
    rp, rs = (None, 0) if residual is None else _slice_ptr(residual, "residual")
    _lib.call("vfi_resize_bilinear", xp, xs, rp, rs, yp, ys, n, c, h, w, ho, wo, int(bool(align_corners)),
              int(bool(relu_input)), _lib.stream_ptr())
    return out


def softmax_channels_(x):
    n, c, h, w = x.shape
    xp, xs = _slice_ptr(x, "x")
    _lib.call("vfi_softmax_channels", xp, xs, xp, xs, n, c, h * w, _lib.stream_ptr())
    return x


def affine_slice(src, dst, div=None, mul=1.0):
    """dst[n] = src[n] / div[n] * mul over the per-sample (C,H,W) block (dst may be a channel slice)."""
    n = src.shape[0]
    count = src[0].numel()
    if dst.shape[0] != n or dst[0].numel() != count:
        raise VfiLibraryError("affine_slice: shape mismatch")
    sp, ss = _slice_ptr(src, "src")
    dp, ds = _slice_ptr(dst, "dst")
    _lib.call("vfi_affine_slice", sp, ss, dp, ds, n, count, _lib.dptr(div, "div") if div is not None else None,
              float(mul), _lib.stream_ptr())
    return dst


def batch_max(x, eps):
    """max over each sample's (C,H,W) block, + eps -> (N,) tensor."""
    n = x.shape[0]
    xp, xs = _slice_ptr(x, "x")
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = torch.empty(n, dtype=torch.int32, device=x.device)
    _lib.call("vfi_batch_max", xp, xs, n, x[0].numel(), float(eps), out.data_ptr(), ws.data_ptr(), _lib.stream_ptr())
    return out


def tanh_residual_clamp(x, base):
    out = torch.empty_like(x)
    _lib.call("vfi_tanh_residual_clamp", _lib.dptr(x, "x"), _lib.dptr(base, "base"), out.data_ptr(), x.numel(),
              _lib.stream_ptr())
    return out


def _out_like(x, out, name):
    """`out`: None (a new tensor) or a contiguous tensor of x's shape the op writes into (a slice of a wider buffer: no
    concat copy afterwards)."""
    if out is None:
        return torch.empty_like(x)
    if tuple(out.shape) != tuple(x.shape) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != x.device:
        raise VfiLibraryError(f"{name}: out must be a contiguous float32 tensor of shape {tuple(x.shape)} on {x.device}")
    return out


def rgb2lab(rgb, out=None):
    """(N,3,H,W) or (3,H,W) rgb in [0,1] -> scaled Lab, same shape (reference src/train/transform.py:17-25)."""
    x = rgb.contiguous()
    hw = x.shape[-1] * x.shape[-2]
    out = _out_like(x, out, "rgb2lab")
    _lib.call("vfi_rgb2lab", _lib.dptr(x, "rgb"), out.data_ptr(), x.numel() // (3 * hw), hw, _lib.stream_ptr())
    return out


def lab2rgb(lab):
    x = lab.contiguous()
    hw = x.shape[-1] * x.shape[-2]
    out = torch.empty_like(x)
    _lib.call("vfi_lab2rgb", _lib.dptr(x, "lab"), out.data_ptr(), x.numel() // (3 * hw), hw, _lib.stream_ptr())
    return out


def channel_mean_diff(a, b=None, scale=1.0, clamp01=False, signed=False):
    """a, b (N,C,H,W) -> (N,H,W): mean over C of a (minus that of b; abs unless signed), * scale, optional clamp."""
    a = a.contiguous()
    n, c, h, w = a.shape
    out = torch.empty((n, h, w), dtype=torch.float32, device=a.device)
    _lib.call("vfi_channel_mean_diff", _lib.dptr(a, "a"), _lib.dptr(b.contiguous(), "b") if b is not None else None,
              out.data_ptr(), n, c, h * w, float(scale), int(bool(clamp01)) | (2 if signed else 0), _lib.stream_ptr())
    return out


def absdiff(x, y, scale=1.0, clamp01=False, out=None):
    """|x - y| * scale, or |x| * scale for y=None (no zero tensor is built to subtract)."""
    x = x.contiguous()
    if y is not None:
        y = y.contiguous()
        if x.shape != y.shape:
            raise VfiLibraryError("absdiff: shape mismatch")
    out = _out_like(x, out, "absdiff")
    _lib.call("vfi_absdiff", _lib.dptr(x, "x"), _lib.dptr(y, "y") if y is not None else None, out.data_ptr(), x.numel(),
              float(scale), int(bool(clamp01)), _lib.stream_ptr())
    return out


def gaussian_filter(x, sigma, truncate=4.0, out=None):
    """scipy.ndimage.gaussian_filter per (H,W) image of x (N,H,W)."""
    x = x.contiguous()
    n, h, w = x.shape
    tmp, out = torch.empty_like(x), _out_like(x, out, "gaussian_filter")
    _lib.call("vfi_gaussian_filter", _lib.dptr(x, "x"), tmp.data_ptr(), out.data_ptr(), n, h, w, float(sigma),
              float(truncate), _lib.stream_ptr())
    return out


def median_filter(x, size):
    """scipy.ndimage.median_filter(size=size) per (H,W) image of x (N,H,W)."""
    x = x.contiguous()
    n, h, w = x.shape
    out = torch.empty_like(x)
    _lib.call("vfi_median_filter", _lib.dptr(x, "x"), out.data_ptr(), n, h, w, int(size), _lib.stream_ptr())
    return out
