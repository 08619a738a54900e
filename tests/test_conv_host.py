"""Host-side check of the Winograd kernel's item-decoding arithmetic (csrc/vfi_conv_common.h: exact division by a
precomputed reciprocal).  hipcc --cuda-host-only: no GPU involved."""
import os
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_division_by_reciprocal_is_exact(tmp_path):
    exe = str(tmp_path / "fastdiv_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-O2", "-std=c++17",
                           "-I", os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "fastdiv_check.cpp"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 wrong" in r.stdout, r.stdout[-2000:]
