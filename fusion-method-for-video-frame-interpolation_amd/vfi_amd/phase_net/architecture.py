"""Image-in / image-out PhaseNet -- mirror of reference src/phase_net/architecture.py:12-71:
`PhaseNet(height, device, num_img, scale_factor, nbands).forward(img_batch, high_level, ada_pred, m)
 -> (prediction, vals_pred, vals_target)` and `.load(path)`.

img_batch is (num_img*C [+ C target], H, W) Lab channel-images, frame-major.  The common inference call
(m=None, high_level=False) takes the fused route: the analysis writes phase/pi and amplitudes directly into
PhaseNet's block-input buffers (no separate_vals / get_concat_layers_inf copies)."""
import math

import numpy as np
import torch

from ..train.pyramid import Pyramid
from ..train.utils import calc_pyr_height, exchange_vals, get_concat_layers_inf, separate_vals
from ..values import DecompValues
from .core import PhaseNetCore


class PhaseNet(torch.nn.Module):
    def __init__(self, height, device, num_img=2, scale_factor=np.sqrt(2), nbands=4):
        super().__init__()
        self.core = PhaseNetCore(height, device, num_img=num_img, nbands=nbands)
        self.pyr = Pyramid(height=height, nbands=nbands, scale_factor=scale_factor, device=device)
        self.train(False)
        self.to(device)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("vfi_amd implements the inference path only (eval mode)")
        return super().train(False)

    def load(self, path="./src/phase_net/phase_net.pt"):
        """architecture.py:34-36 (map_location added: the reference file carries a cuda:0 storage tag)."""
        self.core.load_state_dict(torch.load(path, map_location="cpu"))

    @torch.no_grad()
    def forward(self, img_batch, high_level=False, ada_pred=None, m=None):
        img_batch = img_batch.float()
        if m is None:                                                    # fused route (architecture.py:40-59)
            vals, bufs = self.pyr.filter(img_batch, concat_frames=self.core.num_img, phase_scale=1.0 / math.pi)
            vals_pred = self.core(self.core.normalize_vals(vals, concat=bufs), None)
            vals_target = None
        else:                                                            # hierarchical form with a target image
            vals_batch = self.pyr.filter(img_batch)
            vals_list = separate_vals(vals_batch, self.core.num_img + 1)
            vals_target, vals_list = vals_list[-1], vals_list[:-1]
            vals_input = get_concat_layers_inf(self.pyr, vals_list)
            vals_pred = self.core(self.core.normalize_vals(vals_input), m)
            vals_pred = exchange_vals(DecompValues(vals_pred.high_level, list(vals_pred.phase),
                                                   list(vals_pred.amplitude), vals_pred.low_level),
                                      vals_target, 0, calc_pyr_height(img_batch) - m)   # architecture.py:58-60
        if high_level:                                                   # architecture.py:63-65
            vals_pred.high_level[:] = self.pyr.filter(ada_pred.float()).high_level
        prediction = self.pyr.inv_filter(vals_pred)                      # architecture.py:68
        return prediction, vals_pred, vals_target
