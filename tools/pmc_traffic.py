#!/usr/bin/env python3
"""Turns rocprofv3 PMC passes (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of the same command, as
MI355X_MICROARCH.md prescribes: the TCC block cannot hold both in one pass) into per-launch HBM-side bytes per kernel.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [--out profiles/r02_traffic.json] [--command "..."]

Corrections (MI355X_MICROARCH.md, HBM [CDNA4]): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes; on gfx950
FETCH_SIZE tallies a wide (16 B/lane) coalesced read stream at exactly half its bytes, so reads of kernels that stream 16 B per lane
(`buffer_load ... lds` x4, float4 loads) are doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Both the raw and the corrected figure are written.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# source file of each kernel: bench.py only reports the traffic while the tree still holds the source the counters were taken on
KERNEL_SOURCES = {"conv3x3_winograd4m_kernel": "vfi_conv_winograd4m.hip", "conv3x3_winograd4_kernel": "vfi_conv_winograd4.hip",
                  "conv3x3_winograd_kernel": "vfi_conv_winograd.hip"}


def source_hash(label):
    with open(os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc", KERNEL_SOURCES[label]), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]

csv.field_size_limit(sys.maxsize)

# kernels whose global reads are 16-B-per-lane streams (FETCH_SIZE x 2); label -> substring of the kernel name
WIDE_READERS = {"conv3x3_winograd4m_kernel": "conv3x3_winograd4m_kernel", "conv3x3_winograd4_kernel": "conv3x3_winograd4_kernel<",
                "conv3x3_winograd_kernel": "conv3x3_winograd_kernel"}


def collect(directory, counter):
    per = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                name = row["Kernel_Name"]
                per[name][0] += 1
                per[name][1] += float(row["Counter_Value"])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--command", default="")
    args = ap.parse_args()
    fetch, write = collect(args.fetch_dir, "FETCH_SIZE"), collect(args.write_dir, "WRITE_SIZE")
    out = {}
    for label, sub in WIDE_READERS.items():
        fl = sum(v[0] for k, v in fetch.items() if sub in k)
        fk = sum(v[1] for k, v in fetch.items() if sub in k)
        wl = sum(v[0] for k, v in write.items() if sub in k)
        wk = sum(v[1] for k, v in write.items() if sub in k)
        if not fl or not wl:
            continue
        fetch_b, write_b = fk * 1024.0 / fl, wk * 1024.0 / wl
        out[label] = {"bytes_per_launch": 2.0 * fetch_b + write_b, "fetch_size_bytes_per_launch_raw": fetch_b,
                      "write_size_bytes_per_launch": write_b, "fetch_correction": "x2 (gfx950: 16-B-per-lane streams tallied at half)",
                      "launches_fetch_pass": fl, "launches_write_pass": wl, "command": args.command,
                      "kernel_source_sha16": source_hash(label)}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
