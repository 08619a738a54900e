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
    """vfi_conv2d_algo is the one place the convolution selection lives (profiling labels query it instead of restating
    it).  Without a GPU the library assumes 256 CUs, the part the windows were measured on: the query must reproduce them."""
    h = vfi_amd.lib()
    direct, wino2, wino4 = 0, 1, 2
    assert h.vfi_conv2d_algo(1, 32, 1088, 1920, 3, 5, 0, 0, 1) == direct          # 5x5: direct implicit GEMM
    assert h.vfi_conv2d_algo(1, 64, 1088, 1920, 64, 1, 0, 0, 0) == direct         # 1x1
    assert h.vfi_conv2d_algo(3, 64, 544, 960, 64, 3, 0, 0, 1) == wino4            # 3060 items: >= 7.8 rounds of 256 CUs
    assert h.vfi_conv2d_algo(1, 6, 1088, 1920, 32, 3, 0, 0, 1) == wino2           # Cin < 16
    assert h.vfi_conv2d_algo(3, 128, 272, 480, 128, 3, 0, 0, 1) == wino2          # 1632 items: 6.4 rounds, between the windows
    assert h.vfi_conv2d_algo(1, 64, 544, 960, 64, 3, 0, 0, 1) == wino4            # 1020 items: four full rounds
    assert h.vfi_conv2d_algo(1, 128, 272, 480, 128, 3, 0, 0, 1) == wino2          # 544 items: last round mostly empty
    assert h.vfi_conv2d_algo(0, 64, 64, 64, 64, 3, 0, 0, 1) < 0                   # bad arguments: a status, not an algorithm
