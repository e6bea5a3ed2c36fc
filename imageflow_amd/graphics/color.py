"""imageflow_core/src/graphics/color.rs + lut.rs mirror (tables come from libimageflow_hip.so)."""
import enum

import numpy as np

from .. import _native


class WorkingFloatspace(enum.IntEnum):   # color.rs:4-9 (Gamma is unreachable from the resample node)
    StandardRGB = 0
    LinearRGB = 1


def srgb_to_floatspace_table(space=WorkingFloatspace.LinearRGB):
    t = np.zeros(256, np.float32)
    _native.check(_native.lib().ifhip_table_srgb_to_floatspace(int(space), t.ctypes.data_as(_native.f32p)))
    return t


def linear_to_srgb_table():
    t = np.zeros(16384, np.uint8)
    _native.check(_native.lib().ifhip_table_linear_to_srgb(t.ctypes.data_as(_native.u8p)))
    return t
