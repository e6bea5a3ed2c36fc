"""Mirror of the pixel half of flow/nodes/watermark.rs:100-196: given the placement the host logic computed (fit box,
constraint and gravity are layout arithmetic and stay with imageflow), a watermark is
crop -> ColorFilterSrgb::Alpha(opacity) when opacity < 1 -> DrawImageExact(Compose) onto the canvas."""
from typing import Optional

from ...graphics.bitmaps import Bitmap
from . import color as color_nodes
from .clone_crop_fill_expand import crop as crop_node
from .scale_render import CompositingMode, ResampleHints, render


def gravity1d(align_percentage, inner, outer):
    """WatermarkDef::gravity1d (:60-67)."""
    import math

    import numpy as np
    ratio = np.float32(min(max(align_percentage, 0.0), 100.0)) / np.float32(100.0)
    if (outer < inner and inner < 1) or outer < 1:
        raise ValueError("Watermark fit_box does not work")
    v = float(np.float32(outer - inner) * ratio)
    return int(math.copysign(math.floor(abs(v) + 0.5), v))           # f32::round: half away from zero


def obey_gravity(box, w, h, gravity=None):
    """WatermarkDef::obey_gravity (:69-86); gravity = None (centre) or (x%, y%)."""
    x1, y1, x2, y2 = box
    gx, gy = gravity if gravity is not None else (50.0, 50.0)
    return gravity1d(gx, w, x2 - x1) + x1, gravity1d(gy, h, y2 - y1) + y1


def draw_watermark(canvas: Bitmap, mark: Bitmap, x, y, w, h, opacity: Optional[float] = None, crop=None,
                   hints: Optional[ResampleHints] = None):
    """The node chain of :153-183 executed on device-resident batches.  `mark` is modified when opacity < 1 (the
    reference decodes a private copy of the watermark for every use)."""
    if crop is not None:
        mark = crop_node(mark, *crop)
    op = 1.0 if opacity is None else min(max(float(opacity), 0.0), 1.0)
    if op < 1.0:
        color_nodes.color_filter_srgb(mark, "alpha", op)
    render(canvas, mark, x, y, w, h, hints, CompositingMode.Compose)
    return canvas
