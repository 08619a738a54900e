"""Plugin loader -- mirror of reference src/adacof/models/__init__.py:5-21.

`Model(args)` imports `args.model.lower()` as a dotted module path and calls its `make_model(args)`;
pass `model='vfi_amd.fusion_net.fusion_adacofnet'` (fusion variant) or `'vfi_amd.adacof.models.adacofnet'`.
"""
from importlib import import_module

import torch.nn as nn


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.model = import_module(args.model.lower()).make_model(args)

    def forward(self, frame0, frame1, **kwargs):
        return self.model(frame0, frame1, **kwargs)     # (kwargs: the fusion variant's per-call `return_sides`)

    def load(self, state_dict):
        self.model.load_state_dict(state_dict)

    def get_state_dict(self):
        return self.model.state_dict()

    def get_kernel(self, frame0, frame1):
        return self.model.get_kernel(frame0, frame1)

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("vfi_amd implements the inference path only (eval mode)")
        return super().train(False)
