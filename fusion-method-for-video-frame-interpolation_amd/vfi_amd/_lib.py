"""ctypes binding of libvfi_hip.so (include/vfi_hip.h).  Loads lazily, fails loudly."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libvfi_hip.so"

c_f = ctypes.c_void_p   # device pointers travel as plain addresses
c_i = ctypes.c_int
c_d = ctypes.c_double
c_s = ctypes.c_void_p   # hipStream_t
c_l = ctypes.c_longlong
c_fl = ctypes.c_float


class VfiLibraryError(RuntimeError):
    """The HIP library is missing / failed, or a tensor is not usable by it."""


# name -> argtypes.  tests/test_abi.py checks this table against include/vfi_hip.h.
SIGNATURES = {
    "vfi_debug_poison_lds": [c_s],
    "vfi_adacof_forward": [c_f] * 5 + [c_i] * 8 + [c_s],
    "vfi_adacof_fused": [c_f] * 13 + [c_i] * 6 + [c_s],
    "vfi_conv2d_packed_floats": [c_i] * 3,
    "vfi_conv2d_algo": [c_i] * 9,
    "vfi_conv2d_pack": [c_f] * 3 + [c_i] * 3 + [c_s],
    "vfi_conv2d": [c_f, c_l, c_f, c_f, c_f, c_l, c_f, c_l] + [c_i] * 8 + [c_f, c_l, c_s],
    "vfi_conv2d_pool2": [c_f, c_l, c_f, c_f, c_f, c_l, c_f, c_l, c_i] + [c_i] * 8 + [c_f, c_l, c_s],
    "vfi_conv2d_upsample2x": [c_f, c_l, c_f, c_f, c_f, c_l, c_f, c_l] + [c_i] * 8 + [c_f, c_l, c_s],
    "vfi_conv2d_resized_prefix": [c_f, c_l, c_f, c_l, c_i, c_i, c_i, c_f, c_f, c_f, c_l] + [c_i] * 8 + [c_f, c_l, c_s],
    "vfi_adacof_prepare": [c_f] * 5 + [c_i] * 6 + [c_s],
    "vfi_adacof_fused_rgbx": [c_f] * 13 + [c_i] * 6 + [c_s],
    "vfi_pool2": [c_f, c_l, c_f, c_l] + [c_i] * 5 + [c_s],
    "vfi_resize_bilinear": [c_f, c_l, c_f, c_l, c_f, c_l] + [c_i] * 8 + [c_s],
    "vfi_upsample2x_tapsum": [c_f, c_f, c_i, c_i, c_i, c_fl, c_i, c_s],
    "vfi_softmax_channels": [c_f, c_l, c_f, c_l] + [c_i] * 3 + [c_s],
    "vfi_affine_slice": [c_f, c_l, c_f, c_l, c_i, c_l, c_f, c_fl, c_s],
    "vfi_batch_max": [c_f, c_l, c_i, c_l, c_fl, c_f, c_f, c_s],
    "vfi_phasenet_emit": [c_f, c_l, c_f, c_l, c_f, c_f, c_f, c_i, c_i, c_s],
    "vfi_phasenet_predict": [c_f, c_l, c_f, c_f, c_f, c_l, c_f, c_f, c_l, c_f, c_f, c_i, c_i, c_i, c_i, c_s],
    "vfi_phasenet_emit_low": [c_f, c_l, c_f, c_l, c_f, c_f, c_i, c_i, c_s],
    "vfi_tanh_residual_clamp": [c_f, c_f, c_f, c_l, c_s],
    "vfi_rgb2lab": [c_f, c_f, c_i, c_i, c_s],
    "vfi_lab2rgb": [c_f, c_f, c_i, c_i, c_s],
    "vfi_channel_mean_diff": [c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_i, c_s],
    "vfi_absdiff": [c_f, c_f, c_f, c_l, c_fl, c_i, c_s],
    "vfi_gaussian_filter": [c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_fl, c_s],
    "vfi_median_filter": [c_f, c_f, c_i, c_i, c_i, c_i, c_s],
    "vfi_diff_sums": [c_f, c_f, c_l, c_f, c_f, c_s],
    "vfi_ssim_sum": [c_f] * 5 + [c_i] * 4 + [c_fl, c_fl, c_f, c_f, c_s],
    "vfi_mul": [c_f, c_f, c_f, c_l, c_s],
    "vfi_avg_pool": [c_f, c_f, c_i, c_i, c_i, c_i, c_s],
    "vfi_pyr_plan_create": [c_i, c_i, c_i, c_i, c_d, c_i, ctypes.POINTER(ctypes.c_void_p)],
    "vfi_pyr_plan_destroy": [ctypes.c_void_p],
    "vfi_pyr_plan_level_size": [ctypes.c_void_p, c_i, ctypes.POINTER(c_i), ctypes.POINTER(c_i)],
    "vfi_pyr_plan_prepare_filter": [ctypes.c_void_p, ctypes.c_ulonglong, c_i, c_i, ctypes.POINTER(c_i)],
    "vfi_pyr_apply_filter": [ctypes.c_void_p, c_i, c_f, c_i, c_f, c_s],
    "vfi_pyr_apply_filter_pair": [ctypes.c_void_p, c_i, c_f, c_i, c_f, c_i, c_f, c_s],
    "vfi_pyr_analyze": [ctypes.c_void_p, c_f, c_i, c_f, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f,
                        c_fl, ctypes.c_ulonglong, c_i, c_s],
    "vfi_pyr_analyze_max": [ctypes.c_void_p, c_f, c_i, c_f, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f,
                            c_fl, ctypes.c_ulonglong, c_i, c_f, c_i, c_fl, c_s],
    "vfi_pyr_synthesize": [ctypes.c_void_p, c_f, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f,
                           ctypes.c_ulonglong, c_i, c_f, c_i, c_s],
}
# entry points that return a value instead of a vfi_status
RESTYPES = {"vfi_conv2d_packed_floats": c_l}

_lock = threading.Lock()
_lib = None


def library_path():
    return os.environ.get("VFI_HIP_LIBRARY", os.path.join(_HERE, _LIB_NAME))


def lib():
    """Returns the loaded library; raises VfiLibraryError if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (loads the process's HIP runtime first so both share it)
        path = library_path()
        if not os.path.exists(path):
            raise VfiLibraryError(
                f"{path} not found: build it with `make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` "
                "(or __graft_entry__.build()).  vfi_amd has no CPU fallback.")
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise VfiLibraryError(f"cannot load {path}: {e}") from e
        handle.vfi_abi_version.restype = c_i
        handle.vfi_status_string.restype = ctypes.c_char_p
        handle.vfi_status_string.argtypes = [c_i]
        handle.vfi_last_error.restype = ctypes.c_char_p
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                raise VfiLibraryError(f"{path} does not export {name}")
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, c_i)
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        h = lib()
        raise VfiLibraryError(
            f"{what}: {h.vfi_status_string(status).decode()} ({status}): {h.vfi_last_error().decode()}")


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def check_device(t, name="tensor"):
    """Every call enqueues on the CURRENT device's current stream (stream_ptr): a tensor that lives on another GPU
    would be handed to the wrong device's queue.  One process drives one GPU; a caller that
    really wants a second device in the same process wraps the call in `torch.cuda.device(t.device)`."""
    import torch
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        raise VfiLibraryError(f"{name} is on {t.device} but the current HIP device is cuda:{cur}; "
                              f"wrap the call in torch.cuda.device({t.device.index})")


def dptr(t, name="tensor", dtype=None):
    """Device address of a dense HIP tensor (None -> NULL)."""
    import torch
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise VfiLibraryError(f"{name} must be a tensor on a HIP device (vfi_amd has no CPU path)")
    check_device(t, name)
    if t.dtype != (dtype or torch.float32):
        raise VfiLibraryError(f"{name} must be {dtype or torch.float32}, got {t.dtype}")
    if not t.is_contiguous():
        raise VfiLibraryError(f"{name} must be contiguous")
    return t.data_ptr()


class Recorder:
    """Optional per-call timing (HIP events on the launch stream) used by bench.py / tools; off by default.
    `work` = (kind, amount, label): algorithmic FLOPs ("flop") or bytes ("byte") of the call."""

    def __init__(self):
        self.rows = []

    def summary(self):
        import torch
        torch.cuda.synchronize()
        agg = {}
        for name, work, e0, e1 in self.rows:
            kind, amount, label = work if work else (None, 0.0, name)
            a = agg.setdefault(label or name, dict(entry=name, kind=kind, calls=0, seconds=0.0, work=0.0))
            a["calls"] += 1
            a["seconds"] += e0.elapsed_time(e1) * 1e-3
            a["work"] += amount
        return agg


PROFILE = None


def call(name, *args, work=None):
    rec = PROFILE
    if rec is None:
        check(getattr(lib(), name)(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(lib(), name)(*args), name)
    e1.record()
    rec.rows.append((name, work, e0, e1))
