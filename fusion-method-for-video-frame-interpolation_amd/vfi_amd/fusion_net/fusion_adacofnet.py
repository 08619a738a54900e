"""AdaCoF network, fusion variant -- mirror of reference src/fusion_net/fusion_adacofnet.py.

`make_model(args)` (fusion_adacofnet.py:10-11) is the plugin entry the reference's
`src.adacof.models.Model` resolves by dotted module path; `AdaCoFNet.forward(frame0, frame2)` returns
`(tensorAdaCoF1, tensorAdaCoF2, frame1, UncertaintyMask)` (fusion_adacofnet.py:172-240).

Execution on the MI355X:
  * prologue (reflect pad to /32, mean subtraction, concat): one launch (vfi_adacof_prepare);
  * KernelEstimation U-Net (fusion_adacofnet.py:109-155): fp32-MFMA convs with ReLU and the additive skip
    fused in the epilogue; the seven heads' first convs share their input and run as ONE 64->448 conv,
    their second/third convs read channel slices of its output in place;
  * both AdaCoF samplings + occlusion blend + flow-variance mask: one launch (vfi_adacof_fused); the
    ReplicationPad2d is folded into the sampler's clamp.
"""
import sys

import torch

from .. import _lib, ops
from ..adacof.cupy_module.adacof import FunctionAdaCoF, adacof_fused
from ..nn_util import ConvParams, Indexed, PackedModule

HEADS = ("moduleWeight1", "moduleAlpha1", "moduleBeta1", "moduleWeight2", "moduleAlpha2", "moduleBeta2",
         "moduleOcclusion")


def make_model(args):
    return AdaCoFNet(args).to(torch.device("cuda:{}".format(args.gpu_id)))


def _basic(cin, cout):
    return Indexed({0: ConvParams(cin, cout, 3), 2: ConvParams(cout, cout, 3), 4: ConvParams(cout, cout, 3)})


class KernelEstimation(PackedModule):
    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        k2 = kernel_size ** 2
        self.moduleConv1 = _basic(6, 32)
        self.moduleConv2 = _basic(32, 64)
        self.moduleConv3 = _basic(64, 128)
        self.moduleConv4 = _basic(128, 256)
        self.moduleConv5 = _basic(256, 512)
        self.moduleDeconv5 = _basic(512, 512)
        self.moduleUpsample5 = Indexed({1: ConvParams(512, 512, 3)})
        self.moduleDeconv4 = _basic(512, 256)
        self.moduleUpsample4 = Indexed({1: ConvParams(256, 256, 3)})
        self.moduleDeconv3 = _basic(256, 128)
        self.moduleUpsample3 = Indexed({1: ConvParams(128, 128, 3)})
        self.moduleDeconv2 = _basic(128, 64)
        self.moduleUpsample2 = Indexed({1: ConvParams(64, 64, 3)})
        for name in HEADS[:6]:
            setattr(self, name, Indexed({0: ConvParams(64, 64, 3), 2: ConvParams(64, 64, 3),
                                         4: ConvParams(64, k2, 3), 7: ConvParams(k2, k2, 3)}))
        self.moduleOcclusion = Indexed({0: ConvParams(64, 64, 3), 2: ConvParams(64, 64, 3),
                                        4: ConvParams(64, 64, 3), 7: ConvParams(64, 1, 3)})
        self.train(False)

    def _build_packed(self):
        p = {}
        for name in ("moduleConv1", "moduleConv2", "moduleConv3", "moduleConv4", "moduleConv5", "moduleDeconv5",
                     "moduleDeconv4", "moduleDeconv3", "moduleDeconv2"):
            m = getattr(self, name)
            p[name] = [self.pack(m[i]) for i in (0, 2, 4)]
        for name in ("moduleUpsample5", "moduleUpsample4", "moduleUpsample3", "moduleUpsample2"):
            p[name] = self.pack(getattr(self, name)[1])
        # the seven heads' first convs share their input: one 64 -> 7*64 filter bank
        w = torch.cat([getattr(self, h)[0].weight for h in HEADS], 0)
        b = torch.cat([getattr(self, h)[0].bias for h in HEADS], 0)
        p["heads0"] = ops.PackedConv(w, b)
        for h in HEADS:
            m = getattr(self, h)
            p[h] = [self.pack(m[2]), self.pack(m[4]), self.pack(m[7])]
        # occlusion tail Upsample -> Conv2d(64, 1, 3): channel reduction as a 1x1 conv (taps as output channels) at
        # low resolution, finished by vfi_upsample2x_tapsum (exact by linearity)
        w7 = self.moduleOcclusion[7].weight                              # (1, 64, 3, 3)
        p["occ_taps"] = ops.PackedConv(w7[0].permute(1, 2, 0).reshape(9, -1, 1, 1).contiguous(), None)
        p["occ_bias"] = float(self.moduleOcclusion[7].bias.item())
        return p

    def _basic(self, convs, x):
        for pc in convs:
            x = ops.conv2d(x, pc, "zeros", "relu")
        return x

    def _basic_pooled(self, convs, x):
        """Basic block followed by AvgPool2d(2): -> (block output, pooled output), the pooling fused into the last conv."""
        for pc in convs[:-1]:
            x = ops.conv2d(x, pc, "zeros", "relu")
        return ops.conv2d_pool2(x, convs[-1], False, "zeros", "relu")

    def _up(self, pc, x, skip):
        """Upsample(x2, align_corners=True) -> conv -> ReLU, + skip (fusion_adacofnet.py:28-33,128-146)."""
        return ops.conv2d(x, pc, "zeros", "relu", residual=skip, upsample2x=True)

    def forward_x6(self, x6, softmax=True):
        """softmax=False returns the Subnet_weight LOGITS (the sampler folds the softmax in)."""
        p = self.packed()
        c1, q1 = self._basic_pooled(p["moduleConv1"], x6)          # (c1 itself is not a skip connection; q = AvgPool2d(c))
        c2, q2 = self._basic_pooled(p["moduleConv2"], q1)
        c3, q3 = self._basic_pooled(p["moduleConv3"], q2)
        c4, q4 = self._basic_pooled(p["moduleConv4"], q3)
        c5, q5 = self._basic_pooled(p["moduleConv5"], q4)
        x = self._basic(p["moduleDeconv5"], q5)
        x = self._up(p["moduleUpsample5"], x, c5)
        x = self._up(p["moduleUpsample4"], self._basic(p["moduleDeconv4"], x), c4)
        x = self._up(p["moduleUpsample3"], self._basic(p["moduleDeconv3"], x), c3)
        x = self._up(p["moduleUpsample2"], self._basic(p["moduleDeconv2"], x), c2)
        n, _, h, w = x.shape
        h0 = ops.conv2d(x, p["heads0"], "zeros", "relu")            # (N, 448, h, w)
        outs = []
        for i, name in enumerate(HEADS):
            c_mid, c_up, c_out = p[name]
            t = ops.conv2d(h0[:, 64 * i:64 * (i + 1)], c_mid, "zeros", "relu")
            t = ops.conv2d(t, c_up, "zeros", "relu")
            # Upsample(x2, align_corners=True) -> conv: one launch, the upsampled tensor is never written
            if name.startswith("moduleWeight"):
                t = ops.conv2d(t, c_out, "zeros", None, upsample2x=True)
                if softmax:
                    t = ops.softmax_channels_(t)
            elif name == "moduleOcclusion":
                taps = ops.conv2d(t, p["occ_taps"], "zeros", None)                  # (N, 9, h, w)
                t = ops.new((n, 1, 2 * h, 2 * w), t)
                _lib.call("vfi_upsample2x_tapsum", taps.data_ptr(), t.data_ptr(), n, h, w, p["occ_bias"], 4,
                          _lib.stream_ptr())
            else:
                t = ops.conv2d(t, c_out, "zeros", None, upsample2x=True)
            outs.append(t)
        return tuple(outs)

    def forward(self, rfield0, rfield2):
        """Reference signature (fusion_adacofnet.py:109): two mean-subtracted (N,3,H,W) frames."""
        return self.forward_x6(torch.cat([rfield0, rfield2], 1).contiguous())


class AdaCoFNet(torch.nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.kernel_size = args.kernel_size
        self.kernel_pad = int(((args.kernel_size - 1) * args.dilation) / 2.0)   # fusion_adacofnet.py:163
        self.dilation = args.dilation
        self.get_kernel = KernelEstimation(self.kernel_size)
        self.moduleAdaCoF = FunctionAdaCoF.apply
        self.train(False)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("vfi_amd implements the inference path only (eval mode)")
        return super().train(False)

    def forward(self, frame0, frame2, return_sides=True):
        """The reference returns both sampled sides (tensorAdaCoF1/2, fusion_adacofnet.py:240) although its fused caller
        drops them (src/fusion_net/interpolate_twoframe.py:156,229-237).  `return_sides=False` (a per-call argument, so
        callers sharing the module never see each other's choice) makes the sampler skip their 24 B/px of stores; forward
        then returns None in their place."""
        h0, w0 = int(frame0.shape[2]), int(frame0.shape[3])
        if h0 != int(frame2.shape[2]) or w0 != int(frame2.shape[3]):
            sys.exit("Frame sizes do not match")                                 # fusion_adacofnet.py:177-178
        pad0, pad2, x6 = ops.adacof_prepare(frame0.contiguous(), frame2.contiguous(), rgbx=True)
        w1, a1, b1, w2, a2, b2, occ = self.get_kernel.forward_x6(x6, softmax=False)
        t1, t2, frame1, mask = adacof_fused(pad0, pad2, w1, a1, b1, w2, a2, b2, occ, self.dilation, rgbx=True,
                                            weights_are_logits=True, want_sides=bool(return_sides))
        if x6.shape[2] != h0 or x6.shape[3] != w0:
            # the reference's width crop assigns tensorAdaCoF1 from tensorAdaCoF2 (fusion_adacofnet.py:225);
            # both are unused downstream -- we return the correctly cropped t1.
            t1, t2, frame1, mask = (t[:, :, :h0, :w0].contiguous() if t is not None else None for t in (t1, t2, frame1, mask))
        return t1, t2, frame1, mask
