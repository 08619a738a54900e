"""bench.py command line (driver contract): a bare `python bench.py --gpus N` must start its N ranks itself, from a
parent that has not touched the GPU; on a node with fewer GPUs it must say so instead of dying on an assert."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def test_bare_multi_gpu_invocation_spawns_or_explains():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert re.search(r"--gpus 64 but this node shows \d+ GPU", r.stderr + r.stdout), r.stderr[-500:]
    assert "AssertionError" not in r.stderr
