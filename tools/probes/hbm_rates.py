"""Probe: achievable HBM rates of trivial torch kernels (pure write, copy) next to vfi_resize_bilinear."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops
dev = torch.device("cuda:0")
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 4
    a, b = torch.empty(n, device=dev), torch.empty(n, device=dev)
    print(f"{mb:5d} MiB  fill {n * 4 / t(lambda: a.fill_(1.0)) / 1e12:.2f} TB/s   copy (r+w) {2 * n * 4 / t(lambda: b.copy_(a)) / 1e12:.2f} TB/s")
for (n, c, h, w) in [(1, 25, 544, 960), (2, 25, 544, 960), (3, 64, 272, 480)]:
    x = torch.randn(n, c, h, w, device=dev)
    out = torch.empty(n, c, 2 * h, 2 * w, device=dev)
    s = t(lambda: ops.resize_bilinear(x, (2 * h, 2 * w), True, out=out))
    print(f"resize x2 N{n} C{c} {h}x{w}: {s * 1e3:.3f} ms  {(x.numel() + out.numel()) * 4 / s / 1e12:.2f} TB/s")
