"""vfi_amd -- MI355X (gfx950) implementation of the per-frame inference hot path of
"Fusion Method for Video Frame Interpolation" (steerable-pyramid PhaseNet + AdaCoF + FusionNet).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all
arithmetic on the path runs in hand-written HIP kernels reached through the C ABI of
``libvfi_hip.so`` (``include/vfi_hip.h``).  There is NO CPU fallback: every op raises
``VfiLibraryError`` when the library is missing or a tensor is not on a HIP device.

Sub-packages mirror the reference's module surfaces (reference path -> here):
    src/adacof/cupy_module/adacof.py   -> vfi_amd.adacof.cupy_module.adacof
    src/adacof/models/__init__.py      -> vfi_amd.adacof.models
    src/fusion_net/fusion_adacofnet.py -> vfi_amd.fusion_net.fusion_adacofnet
    src/fusion_net/fusion_net.py       -> vfi_amd.fusion_net.fusion_net
    src/phase_net/phase_net.py         -> vfi_amd.phase_net.phase_net
    src/train/pyramid.py, utils.py, transform.py -> vfi_amd.train.*
    steerable.SCFpyr_PyTorch           -> vfi_amd.steerable.SCFpyr_PyTorch
"""
from ._lib import VfiLibraryError, lib, library_path  # noqa: F401

__all__ = ["VfiLibraryError", "lib", "library_path"]
