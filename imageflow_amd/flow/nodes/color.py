"""Mirror of flow/nodes/color.rs: the ColorFilterSrgb matrices (:86-230) and the color_matrix_srgb mutate node
(:20-38), which runs graphics::color_matrix on the device-resident batch and leaves it in BlendWithSelf."""
import numpy as np

from ...graphics.bitmaps import Bitmap, BitmapCompositing
from ...graphics.bitmap_ops import window_bgra32_apply_color_matrix

f32 = np.float32


def _m(rows):
    return np.array(rows, dtype=np.float32)


def sepia():
    return _m([[0.393, 0.349, 0.272, 0, 0], [0.769, 0.686, 0.534, 0, 0], [0.189, 0.168, 0.131, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 0]])


def grayscale(r, g, b):
    return _m([[r, r, r, 0, 0], [g, g, g, 0, 0], [b, b, b, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]])


def grayscale_flat():
    return grayscale(0.5, 0.5, 0.5)


def grayscale_bt709():
    return grayscale(0.2125, 0.7154, 0.0721)


def grayscale_ry():
    return grayscale(0.5, 0.419, 0.081)


def grayscale_ntsc():
    return grayscale(0.229, 0.587, 0.114)


def invert():
    return _m([[-1, 0, 0, 0, 0], [0, -1, 0, 0, 0], [0, 0, -1, 0, 0], [0, 0, 0, 1, 0], [1, 1, 1, 0, 1]])


def alpha(a):
    return _m([[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, a, 0], [0, 0, 0, 0, 1]])


def contrast(c):
    c = f32(c) + f32(1)
    t = f32(0.5) * (f32(1) - c)
    return _m([[c, 0, 0, 0, 0], [0, c, 0, 0, 0], [0, 0, c, 0, 0], [0, 0, 0, 1, 0], [t, t, t, 0, 1]])


def brightness(factor):
    return _m([[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [factor, factor, factor, 0, 1]])


def saturation(s):
    s = max(f32(s) + f32(1), f32(0))
    c = f32(1) - s
    cr, cg, cb = f32(0.3086) * c, f32(0.6094) * c, f32(0.0820) * c
    return _m([[cr + s, cr, cr, 0, 0], [cg, cg + s, cg, 0, 0], [cb, cb, cb + s, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]])


_FILTERS = {"sepia": sepia, "grayscale_ntsc": grayscale_ntsc, "grayscale_ry": grayscale_ry, "grayscale_flat": grayscale_flat,
            "grayscale_bt709": grayscale_bt709, "invert": invert}
_PARAM_FILTERS = {"alpha": alpha, "contrast": contrast, "saturation": saturation, "brightness": brightness}


def color_matrix_srgb(b: Bitmap, matrix):
    """ColorMatrixSrgbMutDef::mutate (:20-38)."""
    window_bgra32_apply_color_matrix(b, matrix)
    b.compose = BitmapCompositing.BlendWithSelf


def color_filter_srgb(b: Bitmap, name, value=None):
    """ColorFilterSrgb::expand (:49-83): Alpha additionally enables transparency (the alpha channel becomes meaningful,
    after normalising an unused one to 255 -- EnableTransparency)."""
    if name in _PARAM_FILTERS:
        matrix = _PARAM_FILTERS[name](value)
    else:
        matrix = _FILTERS[name]()
    if name == "alpha" and not b.alpha_meaningful:
        from ...graphics.bitmap_ops import normalize_unused_alpha
        normalize_unused_alpha(b)
        b.alpha_meaningful = True
    color_matrix_srgb(b, matrix)
