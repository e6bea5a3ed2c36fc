"""Caller-side semantics restated from the reference (no GPU): DrawImageDef::render's decisions
(flow/nodes/scale_render.rs:237-314) and the JPEG decoder's pre-shrink choice (codecs/mozjpeg_decoder.rs:588-618)."""
import types

import pytest

from imageflow_amd.codecs.mozjpeg_decoder import apply_downscaling, idct_method_for_luma
from imageflow_amd.errors import ErrorKind, FlowError
from imageflow_amd.flow.nodes.scale_render import (CompositingMode, ResampleHints, SharpenWhen,
                                                  resolve_draw_image_exact)
from imageflow_amd.graphics.bitmaps import BitmapCompositing
from imageflow_amd.graphics.color import WorkingFloatspace
from imageflow_amd.graphics.weights import Filter


def bm(w, h, compose=BitmapCompositing.ReplaceSelf, alpha=False):
    return types.SimpleNamespace(w=w, h=h, compose=compose, alpha_meaningful=alpha)


def test_defaults_down_robidoux_up_ginseng_linear():
    p, mode = resolve_draw_image_exact(bm(200, 200), bm(3840, 2160), 0, 0, 200, 200)
    assert p.interpolation_filter == Filter.Robidoux and p.scale_in_colorspace == WorkingFloatspace.LinearRGB
    assert p.sharpen_percent_goal == 0.0
    assert mode == BitmapCompositing.BlendWithSelf            # ReplaceSelf + default blend=compose (:284-286)
    p, _ = resolve_draw_image_exact(bm(800, 800), bm(100, 100), 0, 0, 800, 800)
    assert p.interpolation_filter == Filter.Ginseng
    p, _ = resolve_draw_image_exact(bm(800, 800), bm(100, 100), 0, 0, 800, 800, ResampleHints(up_filter=Filter.Box))
    assert p.interpolation_filter == Filter.Box


def test_overwrite_keeps_replace_self_and_demotes_matte_on_bgra():
    _, mode = resolve_draw_image_exact(bm(10, 10), bm(40, 40), 0, 0, 10, 10, blend=CompositingMode.Overwrite)
    assert mode == BitmapCompositing.ReplaceSelf
    _, mode = resolve_draw_image_exact(bm(10, 10, BitmapCompositing.BlendWithMatte, alpha=True), bm(40, 40), 0, 0, 10, 10,
                                       blend=CompositingMode.Overwrite)
    assert mode == BitmapCompositing.ReplaceSelf
    _, mode = resolve_draw_image_exact(bm(10, 10, BitmapCompositing.BlendWithMatte), bm(40, 40), 0, 0, 10, 10)
    assert mode == BitmapCompositing.BlendWithMatte


def test_sharpen_when_gating():
    h = ResampleHints(sharpen_percent=15.0, sharpen_when=SharpenWhen.Upscaling)
    p, _ = resolve_draw_image_exact(bm(10, 10), bm(40, 40), 0, 0, 10, 10, h)
    assert p.sharpen_percent_goal == 0.0
    h.sharpen_when = SharpenWhen.Downscaling
    p, _ = resolve_draw_image_exact(bm(10, 10), bm(40, 40), 0, 0, 10, 10, h)
    assert p.sharpen_percent_goal == 15.0
    h.sharpen_when = SharpenWhen.SizeDiffers
    p, _ = resolve_draw_image_exact(bm(40, 40), bm(40, 40), 0, 0, 40, 40, h)
    assert p.sharpen_percent_goal == 0.0
    p, _ = resolve_draw_image_exact(bm(40, 40), bm(40, 40), 0, 0, 40, 40, ResampleHints(sharpen_percent=7.0))
    assert p.sharpen_percent_goal == 7.0                      # default SharpenWhen::Always


def test_rect_and_resample_when_errors():
    with pytest.raises(FlowError) as e:
        resolve_draw_image_exact(bm(10, 10), bm(40, 40), 5, 0, 10, 10)
    assert e.value.kind == ErrorKind.InvalidArgument
    with pytest.raises(FlowError):
        resolve_draw_image_exact(bm(10, 10), bm(40, 40), 0, 0, 10, 10, ResampleHints(resample_when="size_differs"))


def test_apply_downscaling_worked_examples_from_the_survey():
    # 3840x2160 width=200 -> hint 420x236 -> i = 1, libjpeg decodes 480x270
    assert apply_downscaling(3840, 2160, 420, 236, 420, 236) == (1, 480, 270)
    # width=800 -> hint 1680x945 -> i = 4, 1920x1080
    assert apply_downscaling(3840, 2160, 1680, 945, 1680, 945) == (4, 1920, 1080)
    assert apply_downscaling(3840, 2160, 3400, 1900, 3400, 1900) == (8, 3840, 2160)      # 7/8 is skipped
    assert apply_downscaling(3840, 2160, 5000, 5000, 100, 100) == (8, 3840, 2160)        # not wider/taller than limits
    assert apply_downscaling(100, 100, 50, 50, 0, 0) == (8, 100, 100)


def test_luma_idct_selector():
    assert idct_method_for_luma(1, True, True) == ("spatial_srgb", 1)
    assert idct_method_for_luma(4, True, False) == ("spatial", 4)
    assert idct_method_for_luma(4, False, True) == ("islow", 8)
    assert idct_method_for_luma(8, True, True) == ("islow", 8)
