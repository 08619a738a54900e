"""Parameter containers for the host-side mirrors of the reference's nn.Modules.

The modules keep the reference's state-dict key names so its checkpoints load with
`load_state_dict`, but they never run a torch convolution: parameters are packed once into the
matrix-core layout (ops.PackedConv) and every forward is a sequence of libvfi_hip.so calls.
"""
import math

import torch
import torch.nn as nn

from . import ops


class ConvParams(nn.Module):
    """weight/bias of one nn.Conv2d (same names, shapes and default init scale as torch's)."""

    def __init__(self, cin, cout, ks):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, ks, ks))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cin * ks * ks)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("ConvParams only holds parameters; use vfi_amd.ops.conv2d")


class BatchNormParams(nn.Module):
    """Parameters/buffers of nn.BatchNorm2d (eval mode only; folded into the preceding conv)."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def fold_args(self):
        return (self.weight, self.bias, self.running_mean, self.running_var, self.eps)


class Indexed(nn.Module):
    """nn.Sequential-like holder whose children are named by the reference's indices ('0','2','4',...)."""

    def __init__(self, items):
        super().__init__()
        for idx, m in items.items():
            self.add_module(str(idx), m)

    def __getitem__(self, idx):
        return getattr(self, str(idx))


class PackedModule(nn.Module):
    """Base class: caches ops.PackedConv objects, rebuilt whenever parameters may have changed."""

    def __init__(self):
        super().__init__()
        self._packed = None

    def _invalidate(self):
        self._packed = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **k):          # .to(device) / .float() ...
        r = super()._apply(fn, *a, **k)
        self._invalidate()
        return r

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("vfi_amd implements the inference path only (eval mode)")
        return super().train(False)

    def packed(self):
        """Packed weights, built on first use.  The pack kernels are enqueued on the CALLER's current stream, and the
        modules are shared by frames in flight on other streams (bench.py, interpolate_video): the build therefore ends
        with a host wait for that stream, once per parameter load, so every later caller on ANY stream finds finished
        buffers (no per-call events; never happens inside a hipGraph capture because a captured frame is warmed up
        first).  `prepare()` is the explicit spelling for callers that want the cost up front."""
        if self._packed is None:
            with torch.no_grad():
                self._packed = self._build_packed()
            dev = next(self.parameters()).device
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
        return self._packed

    def prepare(self):
        self.packed()
        return self

    def _build_packed(self):  # pragma: no cover
        raise NotImplementedError

    @staticmethod
    def pack(conv, bn=None):
        return ops.PackedConv(conv.weight, conv.bias, bn=bn.fold_args() if bn is not None else None)
