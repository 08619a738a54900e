"""PhaseNetCore -- mirror of reference src/phase_net/core.py:9-22 (`PhaseNetCore(height, device, num_img, nbands)`;
same network as phase_net.PhaseNet, constructed from the pyramid height instead of a Pyramid object)."""
from types import SimpleNamespace

from .phase_net import PhaseNet as _PhaseNet


class PhaseNetCore(_PhaseNet):
    def __init__(self, height, device, num_img=2, nbands=4):
        super().__init__(SimpleNamespace(height=height, nbands=nbands), device, num_img=num_img)
        self.height = height
        self.nbands = nbands
