"""The packed-f32 instruction forms of csrc/vfi_fft.h's complex arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 with
op_sel / neg modifiers, written as inline asm where hipcc does not fold a swap-and-negate) against scalar arithmetic, on the
GPU: builds tools/probes/pk_complex.hip and runs it.  The transforms built on these forms are covered end to end by
tests/test_pyramid_gpu.py; this test pins the forms themselves."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_packed_complex_forms_match_scalar_arithmetic(tmp_path, device):
    exe = tmp_path / "pk_complex"
    src = os.path.join(ROOT, "tools", "probes", "pk_complex.hip")
    build = subprocess.run([HIPCC, "-O3", "--offload-arch=gfx950", src, "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    # (a child process of its own: it initialises the GPU itself, nothing is exec'ed from this one)
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "bad 0" in run.stdout, run.stdout
