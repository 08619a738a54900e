"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/vfi_hip.h declares, and the ctypes table binds exactly those symbols."""
import ctypes
import os
import re

from conftest import ROOT

import vfi_amd
from vfi_amd import _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "vfi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"(typedef )?enum.*?;", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vfi_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    h = vfi_amd.lib()
    assert h.vfi_abi_version() == 1
    names = _declared()
    assert "vfi_adacof_forward" in names and "vfi_adacof_fused" in names
    raw = ctypes.CDLL(vfi_amd.library_path())
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/vfi_hip.h but not exported"


def test_ctypes_table_matches_header():
    declared = set(_declared()) - {"vfi_abi_version", "vfi_status_string", "vfi_last_error"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))


def test_header_argument_counts_match_ctypes_table():
    text = open(os.path.join(ROOT, "include", "vfi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(argtypes), (name, nargs, len(argtypes))


def test_status_strings():
    h = vfi_amd.lib()
    assert h.vfi_status_string(0) == b"VFI_OK"
    assert h.vfi_status_string(-2) == b"VFI_ERR_SHAPE"


def test_argument_validation_needs_no_gpu():
    # Null pointers / bad shapes are rejected before anything touches a device.
    h = vfi_amd.lib()
    assert h.vfi_adacof_forward(None, None, None, None, None, 1, 3, 8, 8, 4, 4, 5, 1, None) == -1
    one = ctypes.c_void_p(16)
    assert h.vfi_adacof_forward(one, one, one, one, one, 1, 3, 9, 8, 4, 4, 5, 1, None) == -2
    assert b"does not match" in h.vfi_last_error()


def test_no_cpu_fallback():
    import pytest
    import torch
    from vfi_amd.adacof.cupy_module.adacof import FunctionAdaCoF
    x = torch.zeros(1, 3, 8, 8)
    w = torch.zeros(1, 25, 4, 4)
    with pytest.raises(NotImplementedError):   # reference adacof.py:356-357
        FunctionAdaCoF.apply(x, w, w, w, 1)


def test_conv_algorithm_query_reports_the_library_rule():
    """vfi_conv2d_algo (no GPU call): which kernel a layer gets -- the library's cost model of the two Winograd kernels
    (csrc/vfi_conv_winograd4.hip: winograd4_suits, fitted to profiles/r04_conv_layers.txt), mirrored nowhere else.  The
    expectations are measured winners of that table (256-CU default)."""
    from vfi_amd import _lib
    algo = _lib.lib().vfi_conv2d_algo
    ELU, RELU = 2, 1
    assert algo(3, 64, 1080, 1920, 64, 3, 0, 0, ELU) == 2          # PhaseNet level 7: 12240 items of 16 chunks -> F(4x4)
    assert algo(3, 88, 1080, 1920, 64, 3, 1, 0, ELU) == 2          # ... with a residual: still F(4x4) (RES instantiation)
    assert algo(1, 6, 1088, 1920, 32, 3, 0, 0, RELU) == 2          # first U-Net layer: 2 chunks per item, 8 rounds: F(4x4) by 20 %
    assert algo(3, 256, 136, 240, 256, 3, 0, 0, RELU) == 2         # 864 items (3.4 rounds) of 64 chunks: F(4x4) by 17 %
    assert algo(1, 128, 136, 240, 128, 3, 0, 0, RELU) == 2         # 144 items, one round: F(4x4) by 19 %
    assert algo(1, 64, 272, 480, 64, 3, 0, 0, RELU) == 1           # 272 items = 1.06 rounds: the second round is empty -> F(2x2)
    assert algo(1, 128, 272, 480, 128, 3, 0, 0, RELU) == 2         # 544 items = 2.1 rounds of 32 chunks: F(4x4) by 12 % (F(2x2) before the cheaper item epilogue)
    assert algo(1, 512, 34, 60, 512, 3, 0, 0, RELU) == 1           # 48 long items: F(2x2) with its K split
    assert algo(3, 64, 68, 120, 64, 3, 0, 0, ELU) == 1             # small PhaseNet level
    assert algo(1, 32, 1088, 1920, 32, 3, 0, 1, RELU) == 2         # pooled ReLU layer: F(4x4) POOL instantiation
    assert algo(1, 32, 1088, 1920, 32, 3, 0, 1, ELU) == 1          # pooled non-ReLU: only the F(2x2) path pools in the epilogue
    assert algo(1, 18, 1080, 1920, 32, 5, 0, 0, RELU) == 0 and algo(3, 64, 1080, 1920, 64, 1, 0, 0, 3) == 0
    # 1x1 layers with <= 16 output channels on >= 4096 pixels stream (conv1x1_stream_kernel); small levels and wide layers do not
    assert algo(3, 64, 1080, 1920, 8, 1, 0, 0, 3) == 3 and algo(1, 64, 544, 960, 9, 1, 0, 0, 0) == 3 and algo(3, 64, 382, 679, 8, 1, 0, 0, 3) == 3
    assert algo(3, 64, 9, 15, 8, 1, 0, 0, 3) == 0 and algo(3, 64, 135, 241, 8, 1, 0, 0, 3) == 0 and algo(1, 32, 1080, 1920, 3, 1, 1, 0, 0) == 0
    assert algo(0, 64, 16, 16, 64, 3, 0, 0, 0) < 0 and algo(1, 64, 16, 16, 64, 4, 0, 0, 0) < 0
