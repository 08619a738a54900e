#!/usr/bin/env python3
"""HBM roofline of the hand-written pyramid kernels, one 1080p analysis (6 images) + synthesis (3 images) alone.

  step 1 (GPU box):  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/pyramid_roofline.py run
  step 2 (anywhere): python3 tools/pyramid_roofline.py report OUT/*/*_kernel_trace.csv
"""
import csv
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd")]
H, W, HEIGHT, NA, NS = 1080, 1920, 17, 6, 3


def run():
    import torch
    from vfi_amd.train.pyramid import Pyramid
    from vfi_amd.values import DecompValues
    dev = torch.device("cuda:0")
    pyr = Pyramid(HEIGHT, 4, math.sqrt(2), dev)
    img = torch.rand((NA, H, W), device=dev)
    for _ in range(3):
        v = pyr.filter(img)
        pyr.inv_filter(DecompValues(v.high_level[:NS], [p[:4 * NS] for p in v.phase], [a[:4 * NS] for a in v.amplitude],
                                    v.low_level[:NS]))
    torch.cuda.synchronize()


def report(path):
    rows = list(csv.DictReader(open(path)))
    sizes = [(math.ceil(H / 2 ** (k / 2) - 1e-9), math.ceil(W / 2 ** (k / 2) - 1e-9)) for k in range(HEIGHT - 1)]
    wh = W // 2 + 1
    # algorithmic HBM bytes of each kernel at level 0 (the level that dominates; 8 B = one complex64)
    byt = {
        "pyr_analysis_level_kernel<true": NA * H * wh * 8 * 2 + (4 + 2) * H * W * 4 + NA * 4 * H * W * 8 + NA * sizes[1][0] * sizes[1][1] * 8
                                          + sizes[1][0] * sizes[1][1] * 4,
        "pyr_polar_kernel": NA * 4 * H * W * (8 + 8),
        "pyr_to_complex_kernel": NS * 4 * H * W * (8 + 8),
        "pyr_combine_kernel": NS * 4 * H * W * 8 + 4 * H * W * 4 + NS * sizes[1][0] * sizes[1][1] * 8 + NS * H * W * 8,
        "pyr_final_kernel": NS * H * W * 16 + 2 * H * W * 4 + NS * H * wh * 8,
    }
    grid0 = {"pyr_polar_kernel": str(NA * 4), "pyr_to_complex_kernel": str(NS * 4)}
    print("| kernel (level 0, 1920x1080) | algorithmic MB | best us | median us | GB/s (best) | % of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for name, b in byt.items():
        d = []
        for r in rows:
            if name in r["Kernel_Name"]:
                gx, gy = int(r["Grid_Size_X"]), r["Grid_Size_Y"]
                full = (gy == str(H)) if name not in grid0 else (gy == grid0[name] and gx >= H * W)
                if full:
                    d.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if not d:
            continue
        d.sort()
        best, med = d[0], d[len(d) // 2]
        print(f"| `{name.rstrip('<true')}` | {b/1e6:.0f} | {best:.1f} | {med:.1f} | {b/best/1e3:.0f} | {b/best/1e3/8000*100:.0f} % |")
    tot = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "pyr_" in r["Kernel_Name"]) / 3e6
    fft = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
              if any(t in r["Kernel_Name"] for t in ("fft_", "bluestein", "transpose_", "real2complex", "complex2real", "r2c", "c2r"))) / 3e6
    print(f"\\nper (analysis of {NA} + synthesis of {NS} images): hand-written pyramid kernels {tot:.2f} ms, hipFFT/rocFFT kernels {fft:.2f} ms")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
