#!/usr/bin/env python3
"""Development aid: per-phase cycle sums of conv3x3_winograd_kernel's chunk loop (s_memtime stamps, summed over all
waves).  Needs a library built with -DVFI_WINO_STAMPS (csrc/vfi_conv_winograd.hip) at VFI_HIP_LIBRARY."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import _lib, ops  # noqa: E402

PHASES = ["loop top .. U23 fetch", "MFMA rows 0-1", "counted wait", "barrier", "fetch d,U01 + MFMA rows 2-3", "DMA issue + cursor",
          "transform", "epilogue"]


def main():
    dev = torch.device("cuda:0")
    h = _lib.lib()
    fn = h.vfi_debug_wino_stamps
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    cases = [("64->64 @1080p reflect N3", 3, 64, 64, 1080, 1920, "reflect"), ("64->64 @544x960 zeros N3", 3, 64, 64, 544, 960, "zeros"),
             ("25->25 @1088x1920 N3", 3, 25, 25, 1088, 1920, "zeros"), ("512->512 @68x120 N3", 3, 512, 512, 68, 120, "zeros")]
    for name, n, cin, cout, hh, ww, pad in cases:
        x = torch.randn(n, cin, hh, ww, device=dev)
        pc = ops.PackedConv(torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5, torch.zeros(cout), device=dev)
        out = torch.empty(n, cout, hh, ww, device=dev)
        for _ in range(2):
            ops.conv2d(x, pc, pad, "relu", out=out)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        fn(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d(x, pc, pad, "relu", out=out)
        e1.record()
        torch.cuda.synchronize()
        fn(buf, 0)
        v = list(buf)
        tot = sum(v[:8])
        print(f"== {name}: {e0.elapsed_time(e1):.3f} ms; wave-chunks {v[8]}, wave-items {v[9]}; cycles per wave-chunk {tot / max(v[8], 1):.0f}")
        for i, p in enumerate(PHASES):
            print(f"   {p:32s} {v[i] / max(v[8], 1):8.1f} cyc/chunk  {100.0 * v[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
