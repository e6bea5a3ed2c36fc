"""imageflow_core/src/graphics/scaling.rs mirror: ScaleAndRenderParams (:8-17) and scale_and_render (:19-90),
batched over the frames of a device-resident Bitmap.  All arithmetic happens in libimageflow_hip.so."""
import ctypes as C
from dataclasses import dataclass

import torch

from .. import _native
from .bitmaps import Bitmap, BitmapCompositing
from .color import WorkingFloatspace
from .weights import Filter


@dataclass
class ScaleAndRenderParams:
    x: int
    y: int
    w: int
    h: int
    sharpen_percent_goal: float = 0.0
    interpolation_filter: Filter = Filter.Robidoux
    scale_in_colorspace: WorkingFloatspace = WorkingFloatspace.LinearRGB


class ResamplePlan:
    """Per-shape tables in HBM (ifhip_resample_plan): PixelRowWeights for both axes + the vertical schedule."""

    def __init__(self, in_w, in_h, w, h, filter=Filter.Robidoux, sharpen_percent_goal=0.0):
        self._h = C.c_void_p()
        _native.check(_native.lib().ifhip_resample_plan_create(C.byref(self._h), in_w, in_h, w, h, int(filter),
                                                               float(sharpen_percent_goal)))
        self.key = (in_w, in_h, w, h, int(filter), float(sharpen_percent_goal))

    @property
    def handle(self):
        return self._h

    def horizontal_groups(self):
        """(groups of four source columns, groups of two) of the fused kernel's fast horizontal pass; 0 = not available."""
        import ctypes as C
        four, two = C.c_uint32(), C.c_uint32()
        _native.check(_native.lib().ifhip_resample_plan_horizontal_groups(self._h, C.byref(four), C.byref(two)))
        return four.value, two.value

    def kernel_kind(self, alpha_meaningful=False):
        return _native.lib().ifhip_resample_plan_kernel_kind(self._h, int(alpha_meaningful))

    def __del__(self):
        try:
            if self._h:
                _native.lib().ifhip_resample_plan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


_plans = {}


def plan_for(in_w, in_h, w, h, filter, sharpen, device):
    key = (in_w, in_h, w, h, int(filter), float(sharpen), str(device))
    p = _plans.get(key)
    if p is None:
        with torch.cuda.device(device):
            p = ResamplePlan(in_w, in_h, w, h, filter, sharpen)
        _plans[key] = p
    return p


def _batch_args(plan, input, canvas, info):
    return [plan.handle, input.data.data_ptr(), input.image_bytes, input.stride, int(input.alpha_meaningful), input.n,
            canvas.data.data_ptr(), canvas.image_bytes, canvas.w, canvas.h, canvas.stride, info.x, info.y,
            int(info.scale_in_colorspace), int(canvas.compose), int(canvas.matte)]


def scale_and_render(input: Bitmap, canvas: Bitmap, info: ScaleAndRenderParams, f32_out=None, force_kernel=-1,
                     plan=None):
    """Render every frame of `input` into the (x, y, w, h) rect of the matching frame of `canvas`.
    f32_out: optional float32 cuda tensor [n, h, w, 4] receiving the premultiplied working buffer."""
    assert input.n == canvas.n and input.data.device == canvas.data.device
    dev = input.data.device
    plan = plan or plan_for(input.w, input.h, info.w, info.h, info.interpolation_filter, info.sharpen_percent_goal, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if torch.cuda.current_device() == dev.index:
        _native.check(_native.lib().ifhip_scale_and_render_batch_device(
            *_batch_args(plan, input, canvas, info),
            f32_out.data_ptr() if f32_out is not None else None, force_kernel, C.c_void_p(stream)))
    else:
        with torch.cuda.device(dev):
            _native.check(_native.lib().ifhip_scale_and_render_batch_device(
                *_batch_args(plan, input, canvas, info),
                f32_out.data_ptr() if f32_out is not None else None, force_kernel, C.c_void_p(stream)))
    return plan


def time_scale_and_render(input: Bitmap, canvas: Bitmap, info: ScaleAndRenderParams, launches, plan=None,
                          force_kernel=-1):
    """Average ms per launch measured with hipEvents on the launch stream (bench.py's kernel-duration probe)."""
    dev = input.data.device
    plan = plan or plan_for(input.w, input.h, info.w, info.h, info.interpolation_filter, info.sharpen_percent_goal, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ms = C.c_float(0)
    with torch.cuda.device(dev):
        _native.check(_native.lib().ifhip_time_scale_and_render_batch_device(
            *_batch_args(plan, input, canvas, info), force_kernel, C.c_void_p(stream), launches, C.byref(ms)))
    return ms.value


def scale_and_render_host(inp, in_w, in_h, in_stride, in_alpha_meaningful, canvas, cw, ch, c_stride, info,
                          compositing=BitmapCompositing.ReplaceSelf, matte=0):
    """Host-buffer drop-in (numpy uint8 arrays), exactly the signature the Rust caller would bind."""
    _native.check(_native.lib().ifhip_scale_and_render(
        inp.ctypes.data, in_w, in_h, in_stride, int(in_alpha_meaningful), canvas.ctypes.data, cw, ch, c_stride, 0,
        info.x, info.y, info.w, info.h, int(info.interpolation_filter), float(info.sharpen_percent_goal),
        int(info.scale_in_colorspace), int(compositing), int(matte)))
