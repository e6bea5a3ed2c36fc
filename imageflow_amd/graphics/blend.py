"""imageflow_core/src/graphics/blend.rs mirror: apply_matte (:6-59), in place on a device-resident Bitmap."""
import ctypes as C

import torch

from .. import _native
from .bitmaps import Bitmap


def apply_matte(b: Bitmap, matte_color32: int):
    if not b.alpha_meaningful:      # blend.rs:11-13
        return
    dev = b.data.device
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        _native.check(_native.lib().ifhip_apply_matte_batch_device(b.data.data_ptr(), b.image_bytes, b.n, b.w, b.h,
                                                                   b.stride, 1, matte_color32, C.c_void_p(stream)))
    if (matte_color32 >> 24) == 255:    # Bitmap::apply_matte marks alpha not meaningful for an opaque matte (bitmaps.rs:528-541)
        b.alpha_meaningful = False


def apply_matte_host(bgra, w, h, stride, matte_color32, alpha_meaningful=True):
    _native.check(_native.lib().ifhip_apply_matte(bgra.ctypes.data, w, h, stride, int(alpha_meaningful), matte_color32))
