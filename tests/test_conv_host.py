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


def test_generated_chunk_body_of_the_m32_winograd_kernel_on_the_cpu():
    """tools/gen_wino4m.py emits the hand-scheduled chunk body of conv3x3_winograd4m_kernel; tools/sim_wino4m.py interprets
    that text for one wave on the CPU (64 lanes; an LDS read only delivers at the `s_waitcnt lgkmcnt` that retires it, so
    a missing wait shows up as NaN) against a direct evaluation of the Winograd chunk sum -- plain tiles, and the border
    fix-up in its zero-padding and reflect-padding forms with the one-column shift of the piece before the buffer."""
    import importlib.util
    import sys
    tools = os.path.join(ROOT, "tools")
    sys.path.insert(0, tools)
    try:
        spec = importlib.util.spec_from_file_location("sim_wino4m", os.path.join(tools, "sim_wino4m.py"))
        sim = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sim)
        for mode in (0, 1, 2):
            err, counts = sim.simulate(mode=mode, nchunks=2 + (mode == 0), trow=mode + 1, wave=mode)
            assert err < 1e-5, (mode, err)
            assert counts["mfma"] == 72 * (2 + (mode == 0)) and counts["barrier"] == 3 + (mode == 0), counts
    finally:
        sys.path.remove(tools)


def test_generated_body_header_is_current(tmp_path):
    """csrc/vfi_conv_winograd4m_body.h is what tools/gen_wino4m.py generates from the tree's generator."""
    out = str(tmp_path / "body.h")
    env = {k: v for k, v in os.environ.items() if not k.startswith("W4M_")}
    subprocess.check_call([os.sys.executable, os.path.join(ROOT, "tools", "gen_wino4m.py"), out], env=env)
    with open(out) as a, open(os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc", "vfi_conv_winograd4m_body.h")) as b:
        assert a.read() == b.read(), "re-run tools/gen_wino4m.py"
