"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's per-frame inference hot path (steerable-pyramid
phase decomposition + PhaseNet, AdaCoF deformable sampling, FusionNet blend and the
glue between them).  Each function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package, and only as the checker.  The product package
(``fusion-method-for-video-frame-interpolation_amd/``) never imports it and fails
loudly when its HIP library is missing.

Parity pins (what each restatement has been checked against, see tests/golden/):
  * PhaseNet core, FusionNet, KernelEstimation, AdaCoFNet glue, layout helpers:
    outputs of the reference's own Python classes imported in the build container.
  * AdaCoF sampling: output of the reference's own specialised kernel text run on host.
  * Gaussian / median tail: scipy (the reference calls scipy directly).
  * Steerable pyramid: PARITY UNPINNED -- the arithmetic lives in the third-party
    ``steerable`` package (fork/version unknown, absent from the reference tree and
    from this image); restated from the published upstream algorithm, generalised to
    scale_factor=sqrt(2), pinned only by properties (perfect reconstruction etc.).
  * rgb<->Lab: PARITY UNPINNED -- skimage is absent; restated from the published
    formulas (D65 / 2 degree observer).
"""
