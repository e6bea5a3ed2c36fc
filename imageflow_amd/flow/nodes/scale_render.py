"""Mirror of the caller of the hot path: DrawImageDef::render (imageflow_core/src/flow/nodes/scale_render.rs:221-320).
Only the semantics that decide WHAT reaches scale_and_render are restated -- filter defaults, sharpen gating, working
space default, the compositing switch -- so that a job driving this library behaves like the reference node."""
import enum
from dataclasses import dataclass
from typing import Optional

from ...errors import ErrorKind, FlowError
from ...graphics.bitmaps import Bitmap, BitmapCompositing
from ...graphics.color import WorkingFloatspace
from ...graphics.scaling import ScaleAndRenderParams, scale_and_render
from ...graphics.weights import Filter


class SharpenWhen(enum.Enum):        # imageflow_types::SharpenWhen
    Downscaling = "downscaling"
    Upscaling = "upscaling"
    SizeDiffers = "size_differs"
    Always = "always"


class CompositingMode(enum.Enum):    # imageflow_types::CompositingMode
    Compose = "compose"
    Overwrite = "overwrite"


@dataclass
class ResampleHints:                 # imageflow_types::ResampleHints (lib.rs:925-933)
    sharpen_percent: Optional[float] = None
    down_filter: Optional[Filter] = None
    up_filter: Optional[Filter] = None
    scaling_colorspace: Optional[WorkingFloatspace] = None
    sharpen_when: Optional[SharpenWhen] = None
    resample_when: Optional[str] = None      # only "always" (or None) is legal on DrawImageExact (:246-251)


def resolve_draw_image_exact(canvas: Bitmap, input: Bitmap, x, y, w, h, hints: Optional[ResampleHints] = None,
                             blend: Optional[CompositingMode] = None):
    """Everything render() decides before calling scale_and_render.  Returns (ScaleAndRenderParams, compositing the
    canvas is switched to for the call).  scale_render.rs:237-313."""
    hints = hints or ResampleHints()
    if x + w > canvas.w or y + h > canvas.h:                                                  # :237-240
        raise FlowError(ErrorKind.InvalidArgument, f"DrawImageExact target rect x1={x},y1={y},w={w},h={h} does not fit "
                                                   f"canvas size {canvas.w}x{canvas.h}.")
    if hints.resample_when not in (None, "always"):                                           # :246-251
        raise FlowError(ErrorKind.InvalidArgument, "DrawImageExact already has a canvas and cannot honor ResampleWhen")
    upscaling = w > input.w or h > input.h                                                    # :253-255
    downscaling = w < input.w or h < input.h
    size_differs = w != input.w or h != input.h
    if upscaling:                                                                             # :257-261
        picked = hints.up_filter or Filter.Ginseng
    else:
        picked = hints.down_filter or Filter.Robidoux
    raw = hints.sharpen_percent if hints.sharpen_percent is not None else 0.0
    when = hints.sharpen_when or SharpenWhen.Always                                           # :265-274
    sharpen = raw if (when == SharpenWhen.Always or (when == SharpenWhen.Downscaling and downscaling)
                      or (when == SharpenWhen.Upscaling and upscaling)
                      or (when == SharpenWhen.SizeDiffers and size_differs)) else 0.0
    space = hints.scaling_colorspace if hints.scaling_colorspace is not None else WorkingFloatspace.LinearRGB   # :278-279
    compose = (blend or CompositingMode.Compose) == CompositingMode.Compose                   # :281-282
    mode = canvas.compose
    if mode == BitmapCompositing.ReplaceSelf and compose:                                     # :284-286
        mode = BitmapCompositing.BlendWithSelf
    if mode == BitmapCompositing.BlendWithMatte and not compose and canvas.alpha_meaningful:  # :287-292 (fmt == Bgra32)
        mode = BitmapCompositing.ReplaceSelf
    return ScaleAndRenderParams(x, y, w, h, sharpen, picked, space), mode


def render(canvas: Bitmap, input: Bitmap, x, y, w, h, hints=None, blend=None):
    """DrawImageDef::render: resolve, call the hot path, leave the canvas in BlendWithSelf (:314)."""
    params, mode = resolve_draw_image_exact(canvas, input, x, y, w, h, hints, blend)
    canvas.compose = mode
    scale_and_render(input, canvas, params)
    canvas.compose = BitmapCompositing.BlendWithSelf
    return params
