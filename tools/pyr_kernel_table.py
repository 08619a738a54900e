#!/usr/bin/env python3
"""Per-kernel / per-grid table of a rocprofv3 kernel trace of tools/pyramid_bench.py (development aid)."""
import collections
import csv
import re
import sys

csv.field_size_limit(sys.maxsize)
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "pyr" not in n and "fft" not in n:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = n.replace("void ", "").replace("vfi::fft::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    cfg = re.search(r"Cfg<(\d+), (\d+),.*?>, (true|false)>", n)      # wave-engine kernels: engine length, lines per wave, Bluestein
    short = short.split("(")[0].split("<")[0].replace("vfi::pyrw::", "")[:30]
    if cfg:
        short = f"{short[:22]} M{cfg.group(1)}/L{cfg.group(2)}{'b' if cfg.group(3) == 'true' else ''}"
    k = (short, int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]))
    agg[k][0] += 1
    agg[k][1] += d
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{k[0]:36s} grid {k[1]:6d} x {k[2]:2d}  calls {v[0]:3d}  avg {v[1] / v[0]:8.1f} us  share {100 * v[1] / tot:4.1f} %")
