"""imageflow_core/src/graphics/weights.rs mirror: Filter (:43-78), LobeRatio (:14-40), populate_weights (:681-788).
The tables are built by libimageflow_hip.so (csrc/weights.cpp); nothing is computed in Python."""
import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from .. import _native


class Filter(enum.IntEnum):
    RobidouxFast = 1
    Robidoux = 2
    RobidouxSharp = 3
    Ginseng = 4
    GinsengSharp = 5
    Lanczos = 6
    LanczosSharp = 7
    Lanczos2 = 8
    Lanczos2Sharp = 9
    CubicFast = 10
    Cubic = 11
    CubicSharp = 12
    CatmullRom = 13
    Mitchell = 14
    CubicBSpline = 15
    Hermite = 16
    Jinc = 17
    RawLanczos3 = 18
    RawLanczos3Sharp = 19
    RawLanczos2 = 20
    RawLanczos2Sharp = 21
    Triangle = 22
    Linear = 23
    Box = 24
    CatmullRomFast = 25
    CatmullRomFastSharp = 26
    Fastest = 27
    MitchellFast = 28
    NCubic = 29
    NCubicSharp = 30
    LegacyIDCTFilter = 31


# serde names, imageflow_types/src/lib.rs:144-205
JSON_FILTER_NAMES = {
    "robidoux_fast": Filter.RobidouxFast, "robidoux": Filter.Robidoux, "robidoux_sharp": Filter.RobidouxSharp,
    "ginseng": Filter.Ginseng, "ginseng_sharp": Filter.GinsengSharp, "lanczos": Filter.Lanczos,
    "lanczos_sharp": Filter.LanczosSharp, "lanczos_2": Filter.Lanczos2, "lanczos_2_sharp": Filter.Lanczos2Sharp,
    "cubic": Filter.Cubic, "cubic_sharp": Filter.CubicSharp, "catmull_rom": Filter.CatmullRom,
    "mitchell": Filter.Mitchell, "cubic_b_spline": Filter.CubicBSpline, "hermite": Filter.Hermite,
    "jinc": Filter.Jinc, "triangle": Filter.Triangle, "linear": Filter.Linear, "box": Filter.Box,
    "fastest": Filter.Fastest, "n_cubic": Filter.NCubic, "n_cubic_sharp": Filter.NCubicSharp,
}


class LobeRatio(enum.IntEnum):
    Natural = 0
    Exact = 1
    SharpenPercent = 2


@dataclass
class PixelRowWeights:
    """left_pixel[u], tap count[u] and the concatenated f32 weights (PixelWeightIndexes, weights.rs:555-571)."""
    left_pixel: np.ndarray
    count: np.ndarray
    weights: np.ndarray

    def row(self, u):
        off = int(self.count[:u].sum())
        return self.weights[off:off + int(self.count[u])]


def populate_weights(filter, output_line_size, input_line_size, lobe_ratio=LobeRatio.Natural, lobe_value=0.0,
                     kernel_width_scale=1.0):
    L = _native.lib()
    n = C.c_uint32(0)
    _native.check(L.ifhip_populate_weights(int(filter), int(lobe_ratio), lobe_value, kernel_width_scale,
                                           output_line_size, input_line_size, None, None, None, 0, C.byref(n)))
    left = np.zeros(output_line_size, np.uint32)
    count = np.zeros(output_line_size, np.uint32)
    w = np.zeros(n.value, np.float32)
    _native.check(L.ifhip_populate_weights(int(filter), int(lobe_ratio), lobe_value, kernel_width_scale,
                                           output_line_size, input_line_size,
                                           left.ctypes.data_as(_native.u32p), count.ctypes.data_as(_native.u32p),
                                           w.ctypes.data_as(_native.f32p), n.value, C.byref(n)))
    return PixelRowWeights(left, count, w)
