"""The LDS FFT engine's index arithmetic (csrc/vfi_fft.h: Stockham gather / scatter, LDS padding, Bluestein factors) run
on the CPU: tests/native/fft_host_check.cpp loops over the 256 "threads" between the synchronisation points and compares
every supported kind of length against a double-precision DFT.  hipcc --cuda-host-only: no GPU involved."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_fft_engine_on_host(tmp_path):
    exe = str(tmp_path / "fft_host_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-O2", "-std=c++17",
                           "-I", os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "fft_host_check.cpp"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("rel.err") >= 100


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_wave_fft_engine_on_host(tmp_path):
    """The wave-private register engine (csrc/vfi_wfft.h): tests/native/wfft_host_check.cpp runs the device's own per-lane
    stage / exchange code for every configuration of csrc/vfi_wfft_configs.h lane by lane on the CPU (plain transforms and
    Bluestein's form) against a double-precision DFT, and checks that no exchange leaves the wave's buffer."""
    exe = str(tmp_path / "wfft_host_check")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-O2", "-std=c++17", "-w",
                           "-I", os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "wfft_host_check.cpp"),
                           "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("rel.err") >= 90 and "all wave-engine checks passed" in r.stdout


def test_generated_wave_configurations_are_current():
    """csrc/vfi_wfft_configs.h is what tools/gen_wfft_configs.py generates (the generator validates every configuration
    against numpy.fft before it prints it)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_wfft_configs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(ROOT, "fusion-method-for-video-frame-interpolation_amd", "csrc", "vfi_wfft_configs.h")) as f:
        assert f.read() == r.stdout
