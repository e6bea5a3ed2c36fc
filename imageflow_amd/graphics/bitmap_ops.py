"""Mirrors of the whole-bitmap primitives next to the resampler, on device-resident Bitmaps (a batch of n frames):
graphics/color_matrix.rs:5-29, graphics/copy_rect.rs:12-119, graphics/bitmaps.rs:1504-1548 (fill_rectangle),
graphics/flip.rs:10-38, graphics/transpose.rs:95-121.  Each function has the reference's name, argument meaning and
error behaviour; the work happens in libimageflow_hip.so (csrc/bitmap_ops.hip)."""
import ctypes as C

import numpy as np
import torch

from .. import _native
from .bitmaps import Bitmap, BitmapCompositing

_u32 = C.c_uint32


def _bind():
    L = _native.lib()
    if getattr(L, "_bitmap_ops_bound", False):
        return L
    frames = [C.c_void_p, C.c_size_t, _u32, _u32, _u32, _u32]          # ptr, image_bytes, n, w, h, stride
    L.ifhip_apply_color_matrix_batch_device.argtypes = frames + [C.c_void_p, C.c_void_p]
    L.ifhip_apply_color_matrix.argtypes = [C.c_void_p, _u32, _u32, _u32, C.c_void_p]
    L.ifhip_copy_rect_batch_device.argtypes = ([C.c_void_p, C.c_size_t, _u32, _u32, _u32, C.c_int] +
                                               [C.c_void_p, C.c_size_t, _u32, _u32, _u32, C.POINTER(C.c_int)] +
                                               [_u32] * 7 + [C.c_void_p])
    L.ifhip_copy_rect.argtypes = ([C.c_void_p, _u32, _u32, _u32, C.c_int, C.c_void_p, _u32, _u32, _u32, C.POINTER(C.c_int)] + [_u32] * 6)
    L.ifhip_fill_rect_batch_device.argtypes = frames + [C.c_int, _u32, _u32, _u32, _u32, _u32, C.c_void_p]
    L.ifhip_fill_rect.argtypes = [C.c_void_p, _u32, _u32, _u32, C.c_int, _u32, _u32, _u32, _u32, _u32]
    L.ifhip_normalize_unused_alpha_batch_device.argtypes = frames + [C.c_int, C.c_void_p]
    L.ifhip_flip_vertical_batch_device.argtypes = frames + [C.c_void_p]
    L.ifhip_flip_horizontal_batch_device.argtypes = frames + [C.c_void_p]
    L.ifhip_flip_vertical.argtypes = [C.c_void_p, _u32, _u32, _u32]
    L.ifhip_flip_horizontal.argtypes = [C.c_void_p, _u32, _u32, _u32]
    L.ifhip_transpose_batch_device.argtypes = ([C.c_void_p, C.c_size_t, _u32, _u32, _u32] * 2) + [_u32, C.c_void_p]
    L.ifhip_transpose.argtypes = [C.c_void_p, _u32, _u32, _u32, C.c_void_p, _u32, _u32, _u32]
    L._bitmap_ops_bound = True
    return L


def _stream(b: Bitmap):
    return C.c_void_p(torch.cuda.current_stream(b.data.device).cuda_stream)


def _frames(b: Bitmap):
    return (b.data.data_ptr(), b.image_bytes, b.n, b.w, b.h, b.stride)


def window_bgra32_apply_color_matrix(b: Bitmap, matrix):
    """color_matrix.rs:5-29; matrix = 5x5 float32 (rows as in flow/nodes/color.rs)."""
    m = np.ascontiguousarray(matrix, np.float32).reshape(25)
    with torch.cuda.device(b.data.device):
        _native.check(_bind().ifhip_apply_color_matrix_batch_device(*_frames(b), m.ctypes.data, _stream(b)))


def copy_rectangle(input: Bitmap, canvas: Bitmap, from_x, from_y, to_x, to_y, w, h):
    """copy_rect.rs:12-119.  Leaves the canvas in BlendWithSelf (:38) and updates the alpha flags like the reference."""
    if input.n != canvas.n:
        raise ValueError("input and canvas batches differ in length")
    flag = C.c_int(int(canvas.alpha_meaningful))
    with torch.cuda.device(canvas.data.device):
        _native.check(_bind().ifhip_copy_rect_batch_device(
            input.data.data_ptr(), input.image_bytes, input.w, input.h, input.stride, int(input.alpha_meaningful),
            canvas.data.data_ptr(), canvas.image_bytes, canvas.w, canvas.h, canvas.stride, C.byref(flag),
            from_x, from_y, to_x, to_y, w, h, canvas.n, _stream(canvas)))
    canvas.compose = BitmapCompositing.BlendWithSelf
    canvas.alpha_meaningful = bool(flag.value)


def fill_rectangle(b: Bitmap, color32, x, y, x2, y2):
    """bitmaps.rs:1504-1548; color32 = Color32 0xAARRGGBB."""
    with torch.cuda.device(b.data.device):
        _native.check(_bind().ifhip_fill_rect_batch_device(*_frames(b), int(b.compose.value), x, y, x2, y2, color32, _stream(b)))


def normalize_unused_alpha(b: Bitmap):
    """bitmaps.rs:1570-1576."""
    with torch.cuda.device(b.data.device):
        _native.check(_bind().ifhip_normalize_unused_alpha_batch_device(*_frames(b), int(b.alpha_meaningful), _stream(b)))


def flow_bitmap_bgra_flip_vertical_safe(b: Bitmap):
    with torch.cuda.device(b.data.device):
        _native.check(_bind().ifhip_flip_vertical_batch_device(*_frames(b), _stream(b)))


def flow_bitmap_bgra_flip_horizontal_safe(b: Bitmap):
    with torch.cuda.device(b.data.device):
        _native.check(_bind().ifhip_flip_horizontal_batch_device(*_frames(b), _stream(b)))


def bitmap_window_transpose(frm: Bitmap, to: Bitmap):
    """transpose.rs:95-121."""
    if frm.n != to.n:
        raise ValueError("input and canvas batches differ in length")
    with torch.cuda.device(to.data.device):
        _native.check(_bind().ifhip_transpose_batch_device(frm.data.data_ptr(), frm.image_bytes, frm.w, frm.h, frm.stride,
                                                           to.data.data_ptr(), to.image_bytes, to.w, to.h, to.stride,
                                                           to.n, _stream(to)))
