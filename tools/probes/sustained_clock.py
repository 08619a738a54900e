#!/usr/bin/env python3
"""Probe: what the device clocks at while the F(4x4) layer runs back to back (the fp32-MFMA peak of the roofline, 157.3
TFLOP/s, is quoted at the 2.4 GHz boost clock).  A sustained loop of the reference layer in this process; rocm-smi sampled
from a child process every half second (sclk, power); the kernel's own s_memtime cycles / HIP-event time as a second estimate
needs the -DW4M_STAMPS build and is in w4m_stamps.py."""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
from vfi_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn((3, 64, 1080, 1920), device=dev)
pc = ops.PackedConv(torch.randn(64, 64, 3, 3) / 24.0, torch.zeros(64), device=dev)
out = torch.empty_like(x)


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20)
        return " | ".join(l for l in r.stdout.splitlines() if l.strip())[:400]
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"


print("idle:", smi(), flush=True)
gf = 2.0 * 3 * 64 * 64 * 36 * (1080 * 1920 / 16) / 1e9
t_end = time.time() + float(os.environ.get("SECONDS_BUSY", 12))
k = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ops.conv2d(x, pc, "reflect", "elu", out=out)
    e1.record()
    s = smi()                      # (sampled while the 300 launches are in flight)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print(f"block {k}: {ms:.3f} ms/launch = {gf / ms:.1f} TFLOP/s; {s}", flush=True)
    k += 1
