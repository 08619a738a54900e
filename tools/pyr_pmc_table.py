#!/usr/bin/env python3
"""Per-kernel PMC table of tools/pyramid_bench.py runs under `rocprofv3 --pmc ...` (one directory per counter pass):
sums every counter per kernel name (wave-engine kernels are keyed by engine length) and prints per-wave / per-launch figures.

    python3 tools/pyr_pmc_table.py <pass dir> [<pass dir> ...]
"""
import collections
import csv
import glob
import os
import re
import sys

csv.field_size_limit(sys.maxsize)


def short(n):
    cfg = re.search(r"Cfg<(\d+), (\d+),.*?>, (true|false)", n)
    base = n.replace("void ", "").split("(")[0].split("<")[0].replace("vfi::pyrw::", "").replace("(anonymous namespace)::", "").replace("vfi::fft::", "")
    return f"{base} M{cfg.group(1)}/L{cfg.group(2)}{'b' if cfg.group(3) == 'true' else ''}" if cfg else base


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not any(s in k for s in ("_kernel",)) or ("pyr" not in row["Kernel_Name"] and "fft" not in row["Kernel_Name"]):
                    continue
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k][row["Counter_Name"]] += 1
rows = []
for k, c in agg.items():
    n = max(launches[k].values())
    waves = c.get("SQ_WAVES", 0.0)
    if not waves:
        continue
    cyc = c.get("SQ_WAVE_CYCLES", 0.0)
    rows.append((c.get("SQ_INSTS_VALU", 0) , k, n, waves / n,
                 c.get("SQ_INSTS_VALU", 0) / waves, c.get("SQ_INSTS_SALU", 0) / waves, c.get("SQ_INSTS_LDS", 0) / waves,
                 (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / waves,
                 100 * c.get("SQ_WAIT_ANY", 0) / cyc if cyc else float("nan"), 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / cyc if cyc else float("nan"),
                 100 * c.get("SQ_ACTIVE_INST_LDS", 0) / cyc if cyc else float("nan"),
                 100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_ACTIVE_INST_LDS", 0), 1.0),
                 c.get("FETCH_SIZE", 0) * 1024 / max(launches[k].get("FETCH_SIZE", 1), 1) / 1e6, c.get("WRITE_SIZE", 0) * 1024 / max(launches[k].get("WRITE_SIZE", 1), 1) / 1e6))
print(f"{'kernel':40s} {'launches':>8s} {'waves':>7s} {'VALU/w':>8s} {'SALU/w':>7s} {'LDS/w':>7s} {'VMEM/w':>7s} {'wait%':>6s} {'valu%':>6s} {'lds%':>5s} {'confl%':>6s} {'fetchMB':>8s} {'writeMB':>8s}")
for r in sorted(rows, reverse=True)[:int(os.environ.get("TOP", 16))]:
    print(f"{r[1]:40s} {r[2]:8d} {r[3]:7.0f} {r[4]:8.0f} {r[5]:7.0f} {r[6]:7.0f} {r[7]:7.0f} {r[8]:6.1f} {r[9]:6.1f} {r[10]:5.1f} {r[11]:6.1f} {r[12]:8.1f} {r[13]:8.1f}")
