"""rgb <-> Lab -- mirror of reference src/train/transform.py (skimage D65/2deg + L/100, (ab+128)/255).
The reference bounces every call through the host (`.cpu()` -> skimage -> `torch.tensor`); here each
conversion is one kernel launch on the tensor's device (csrc/vfi_image.hip)."""
from .. import ops


def rgb2lab(img, light=100, ab_mul=255, ab_max=128):
    """[B, C, H, W] (transform.py:6-14)."""
    _check(light, ab_mul, ab_max)
    return ops.rgb2lab(img.float())


def rgb2lab_single(img, light=100, ab_mul=255, ab_max=128):
    """[C, H, W] (transform.py:17-25)."""
    _check(light, ab_mul, ab_max)
    return ops.rgb2lab(img.float())


def lab2rgb(img, light=100, ab_mul=255, ab_max=128):
    """[B, C, H, W] (transform.py:28-37)."""
    _check(light, ab_mul, ab_max)
    return ops.lab2rgb(img.float())


def lab2rgb_single(img, light=100, ab_mul=255, ab_max=128):
    """[C, H, W] (transform.py:40-49)."""
    _check(light, ab_mul, ab_max)
    return ops.lab2rgb(img.float())


def _check(light, ab_mul, ab_max):
    if (light, ab_mul, ab_max) != (100, 255, 128):
        raise ops.VfiLibraryError("only the reference's default Lab scaling (100, 255, 128) is implemented")
