"""PhaseNetBlock -- mirror of reference src/phase_net/block.py:4-32 (parameter holder; the arithmetic runs in
vfi_amd.phase_net.phase_net.PhaseNet.forward as three fused fp32-MFMA conv launches per block)."""
from .phase_net import PhaseNetBlock  # noqa: F401
