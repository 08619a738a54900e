"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/adacof_cpu.c.

Follows reference src/adacof/cupy_module/adacof.py:313-361 (FunctionAdaCoF.forward:
shape asserts, output allocation) and src/fusion_net/fusion_adacofnet.py:195-213.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_c.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_c.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def adacof_forward(inp, weight, offset_i, offset_j, dilation):
    """adacof.py:313-361.  inp (N,C,Hin,Win); weight/offsets (N,F*F,H,W) -> (N,C,H,W)."""
    inp, weight, offset_i, offset_j = map(_f32, (inp, weight, offset_i, offset_j))
    n, c, hin, win = inp.shape
    f = int(round(np.sqrt(weight.shape[1])))
    h, w = weight.shape[2:]
    assert hin - ((f - 1) * dilation + 1) == h - 1  # adacof.py:326
    assert win - ((f - 1) * dilation + 1) == w - 1  # adacof.py:327
    out = np.zeros((n, c, h, w), np.float32)
    _lib().oracle_adacof_forward(_p(inp), _p(weight), _p(offset_i), _p(offset_j), _p(out),
                                 n, c, hin, win, h, w, f, dilation)
    return out


def adacof_forward_window(inp, weight, offset_i, offset_j, dilation):
    """Same arithmetic on a row window: `inp` may hold MORE rows than the shape relation of
    adacof.py:326 allows (used to check a slice of a full-size launch; taps clamp to inp)."""
    inp, weight, offset_i, offset_j = map(_f32, (inp, weight, offset_i, offset_j))
    n, c, hin, win = inp.shape
    f = int(round(np.sqrt(weight.shape[1])))
    h, w = weight.shape[2:]
    assert win - ((f - 1) * dilation + 1) == w - 1 and hin >= h + (f - 1) * dilation
    out = np.zeros((n, c, h, w), np.float32)
    _lib().oracle_adacof_forward(_p(inp), _p(weight), _p(offset_i), _p(offset_j), _p(out),
                                 n, c, hin, win, h, w, f, dilation)
    return out


def blend_mask(t1, t2, occ, w1, a1, b1, w2, a2, b2):
    """fusion_adacofnet.py:198-213 -> (frame1 (N,C,H,W), UncertaintyMask (N,1,H,W))."""
    t1, t2, occ, w1, a1, b1, w2, a2, b2 = map(_f32, (t1, t2, occ, w1, a1, b1, w2, a2, b2))
    n, c, h, w = t1.shape
    k = w1.shape[1]
    frame = np.zeros((n, c, h, w), np.float32)
    mask = np.zeros((n, 1, h, w), np.float32)
    _lib().oracle_adacof_blend_mask(_p(t1), _p(t2), _p(occ), _p(w1), _p(a1), _p(b1),
                                    _p(w2), _p(a2), _p(b2), _p(frame), _p(mask), n, c, h, w, k)
    return frame, mask
