"""Probe: every non-conv library call of one 1080p frame with its sizes, time and bytes/s (which of the small kernels are
far from the HBM rate).  Hooks vfi_amd._lib.call; one frame alone on the device, one stream."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
import bench
from vfi_amd import _lib
dev = torch.device("cuda:0")
runners, _ = bench.build_runner(dev, 1)
f0, f2 = torch.rand(3, 1080, 1920, device=dev), torch.rand(3, 1080, 1920, device=dev)
for _ in range(3):
    runners[0](f0, f2, output_baseline=True)
torch.cuda.synchronize()
rows = []
orig = _lib.call


def hooked(name, *args, work=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *args, work=work)
    e1.record()
    rows.append((name, args, e0, e1))


_lib.call = hooked
import vfi_amd.ops as ops
runners[0](f0, f2, output_baseline=True)
torch.cuda.synchronize()
_lib.call = orig
want = sys.argv[1:] or ["vfi_resize_bilinear", "vfi_median_filter", "vfi_affine_slice", "vfi_phasenet_emit", "vfi_pool2"]
tot = collections.Counter()
for name, args, e0, e1 in rows:
    ms = e0.elapsed_time(e1)
    tot[name] += ms
    if name == "vfi_conv2d" and "conv1x1" in want and args[13] == 1:      # 1x1 layers: bytes/s of reading the input + writing the output
        n, cin, h, w, cout = args[8:13]
        by = 4.0 * n * (cin + cout) * h * w
        print(f"vfi_conv2d 1x1 {n}x{cin}->{cout} @{h}x{w}: {1e3 * ms:8.1f} us  {by / 1e6:8.1f} MB  {by / ms / 1e6:7.1f} GB/s")
    if name not in want:
        continue
    ints = [a for a in args if isinstance(a, int) and not isinstance(a, bool) and abs(a) < (1 << 24)]
    line = f"{name:24s} {1e3 * ms:8.1f} us  ints={ints}"
    if name == "vfi_resize_bilinear":
        n, c, h, w, ho, wo = args[6:12]
        by = 4.0 * n * c * (h * w + ho * wo)
        line += f"  {n}x{c}x{h}x{w}->{ho}x{wo}  {by / 1e6:8.1f} MB  {by / ms / 1e6:7.1f} GB/s"
    print(line)
print("--- totals per entry point (ms, event time incl. launch gaps)")
for k, v in tot.most_common():
    print(f"{k:28s} {v:8.3f}")
