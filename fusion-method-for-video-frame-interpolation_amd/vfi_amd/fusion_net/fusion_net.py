"""FusionNet -- mirror of reference src/fusion_net/fusion_net.py (same constructor, state-dict keys
and forward signature), executed as libvfi_hip.so calls on the MI355X.

forward (fusion_net.py:46-77): cat(base, adacof, phase, other, maps) -> 3 x [conv, ReLU, skip, maxpool]
-> bottleneck -> 3 x [ReLU, bilinear x2, + skip, conv] -> tanh -> base|phase + res -> clamp(0,1).
Fusions: ReLU in the conv epilogue; `deconvolution(relu(x)) + s` is one resize launch; tanh + add +
clamp is one launch.  The checkpoint's unused `net.*` stack (fusion_net.py:11-20) is kept as
parameters only so `load_state_dict` stays strict.
"""
import torch

from .. import ops
from ..nn_util import ConvParams, Indexed, PackedModule


class FusionNet(PackedModule):
    def __init__(self, num_imgs=5, uncertainty_maps=3, kernel=3, pad=3, dil=3):
        super().__init__()
        cin = 3 * num_imgs + uncertainty_maps
        self.in_channels = cin
        # dead stack of the reference (fusion_net.py:11-20): parameters only
        self.net = Indexed({0: ConvParams(cin, 64, kernel), 2: ConvParams(64, 64, kernel),
                            4: ConvParams(64, 64, kernel), 6: ConvParams(64, 3, kernel)})
        self.encoder_layers = Indexed({0: ConvParams(cin, 32, 5), 1: ConvParams(32, 64, 5), 2: ConvParams(64, 128, 3)})
        self.bottleneck_layer = ConvParams(128, 128, 3)
        self.decoder_layers = Indexed({0: ConvParams(128, 64, 5), 1: ConvParams(64, 32, 5), 2: ConvParams(32, 3, 1)})
        self.residuals = []
        self.train(False)

    def _build_packed(self):
        return {"enc": [self.pack(self.encoder_layers[i]) for i in range(3)],
                "mid": self.pack(self.bottleneck_layer),
                "dec": [self.pack(self.decoder_layers[i]) for i in range(3)]}

    def forward(self, base, adacof, phase, other, maps, save=False, variant=0):
        p = self.packed()
        parts = [base, adacof, phase, other] + ([maps] if maps is not None else [])
        n, _, h, w = base.shape
        if h % 8 or w % 8:
            raise ops.VfiLibraryError(f"FusionNet needs H, W multiples of 8, got {h}x{w}")
        x = ops.new((n, self.in_channels, h, w), base)
        c0 = 0
        for t in parts:                                   # torch.cat of fusion_net.py:49
            ops.affine_slice(t.contiguous(), x[:, c0:c0 + t.shape[1]])
            c0 += t.shape[1]
        assert c0 == self.in_channels
        skip = []
        for i in range(3):                                # fusion_net.py:55-59
            s, x = ops.conv2d_pool2(x, p["enc"][i], True, "reflect", "relu")     # conv + ReLU, and its MaxPool2d(2)
            skip.append(s)
        x = ops.conv2d(x, p["mid"], "reflect", None)      # :61
        for i, s in enumerate(skip[::-1]):                # :63-67
            x = ops.resize_bilinear(x, s.shape[2:], align_corners=False, relu_input=True, residual=s)
            x = ops.conv2d(x, p["dec"][i], "reflect", None)
        out = ops.tanh_residual_clamp(x, (phase if variant == 1 else base).contiguous())   # :69-77
        if save:
            self.residuals.append(float((out - (phase if variant == 1 else base)).sum().item()))
        return out
