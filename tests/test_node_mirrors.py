"""Host logic of the node mirrors added around the hot path (no GPU): ColorFilterSrgb matrices (flow/nodes/color.rs),
watermark placement arithmetic (flow/nodes/watermark.rs:60-86), crop windows (bitmaps.rs:841-859), orientation sizes
(rotate_flip_transpose.rs:30-39), libjpeg quality scaling and sampling factors (codecs/mozjpeg.rs)."""
import numpy as np
import pytest
import torch

from imageflow_amd.codecs import mozjpeg as MJ
from imageflow_amd.errors import FlowError
from imageflow_amd.flow.nodes import clone_crop_fill_expand as CC
from imageflow_amd.flow.nodes import color as CN
from imageflow_amd.flow.nodes import rotate_flip_transpose as RT
from imageflow_amd.flow.nodes import watermark as WM
from imageflow_amd.graphics.bitmaps import Bitmap


def test_color_filter_matrices_have_the_reference_entries():
    f = np.float32
    assert CN.sepia()[1, 0] == f(0.769) and CN.sepia()[4, 4] == 0            # color.rs:86-94: last row all zero
    g = CN.grayscale_bt709()
    assert np.all(g[0, :3] == f(0.2125)) and np.all(g[1, :3] == f(0.7154)) and np.all(g[2, :3] == f(0.0721)) and g[3, 3] == 1
    assert np.all(CN.grayscale_ntsc()[0, :3] == f(0.229))                   # grayscale_y, color.rs:114-119
    inv = CN.invert()
    assert np.all(np.diag(inv)[:3] == -1) and np.all(inv[4, :3] == 1) and inv[3, 3] == 1
    a = CN.alpha(0.25)
    assert a[3, 3] == f(0.25) and np.array_equal(np.delete(np.diag(a), 3), np.ones(4, np.float32))
    c = CN.contrast(0.5)                                                     # c = 1.5, t = 0.5 * (1 - 1.5)
    assert c[0, 0] == f(1.5) and c[4, 0] == f(-0.25)
    s = CN.saturation(-1.0)                                                  # fully desaturated: luminance weights only
    assert np.allclose(s[:3, 0], [0.3086, 0.6094, 0.0820]) and s[0, 0] == s[0, 1]
    assert CN.saturation(-5.0)[0, 0] == CN.saturation(-1.0)[0, 0]           # "Stop at -1"
    assert CN.brightness(0.1)[4, 2] == f(0.1) and CN.brightness(0.1)[4, 3] == 0


def test_watermark_gravity():
    assert WM.gravity1d(50, 10, 100) == 45 and WM.gravity1d(100, 10, 100) == 90 and WM.gravity1d(0, 10, 100) == 0
    assert WM.gravity1d(150, 10, 100) == 90 and WM.gravity1d(-3, 10, 100) == 0       # percentage clamped to 0..100
    assert WM.gravity1d(50, 11, 100) == 45                                            # 44.5 rounds away from zero (f32::round)
    assert WM.gravity1d(50, 120, 100) == -10                                          # larger than the box: negative offset
    with pytest.raises(ValueError):
        WM.gravity1d(50, 10, 0)
    assert WM.obey_gravity((10, 20, 110, 80), 50, 20) == (35, 40)
    assert WM.obey_gravity((10, 20, 110, 80), 50, 20, (100.0, 0.0)) == (60, 20)


def test_crop_is_a_window_onto_the_same_bytes():
    n, w, h, stride = 2, 10, 6, 64
    data = torch.arange(n * h * stride, dtype=torch.int32).to(torch.uint8).reshape(n, h * stride)
    b = Bitmap(data, w, h, stride)
    c = CC.crop(b, 2, 1, 7, 5)
    assert (c.w, c.h, c.stride, c.n) == (5, 4, stride, n) and c.image_bytes == h * stride
    assert c.data.data_ptr() == data.data_ptr() + 1 * stride + 2 * 4
    assert torch.equal(c.data[1, :20], data[1, stride + 8: stride + 28])
    for bad in ((3, 1, 3, 5), (0, 0, 11, 6), (0, 4, 5, 4), (0, 0, 10, 7)):
        with pytest.raises(FlowError):
            CC.crop(b, *bad)
    # to_numpy of a window: rows at the parent's stride, the (short) last row zero-padded
    got = c.to_numpy()
    full = data.numpy().reshape(n, h, stride)
    assert got.shape == (n, 4, stride)
    assert np.array_equal(got[:, :, :20], full[:, 1:5, 8:28])
    assert np.array_equal(got[:, :3], full.reshape(n, -1)[:, stride + 8: stride + 8 + 3 * stride].reshape(n, 3, stride))


def test_orientation_sizes_and_quality_tables():
    assert [RT.oriented_size(30, 20, f) for f in range(1, 9)] == [(30, 20)] * 4 + [(20, 30)] * 4
    assert RT.oriented_size(30, 20, 0) == (30, 20) and RT.oriented_size(30, 20, 9) == (30, 20)
    q50, q75 = MJ.quant_tables_for_quality(50), MJ.quant_tables_for_quality(75)
    assert q50[0, 0] == 16 and q50[1, 0] == 17 and np.array_equal(q50[1], q50[2])      # scale 100: the Annex K tables
    assert q75[0, 0] == 8 and q75[0, 1] == 6                                            # (16*50+50)/100, (11*50+50)/100
    assert MJ.sampling_factors((2, 2), (1, 1)) == ([2, 1, 2], [2, 1, 2])                # mixed chroma sizes, mozjpeg.rs:141-149


def test_region_percent_coordinates_round_in_f32_away_from_zero():
    """RegionPercentDef::get_coords (clone_crop_fill_expand.rs:265-286)."""
    assert CC.region_percent_coords(50, 30, 10, 20, 110, 60) == (5, 6, 55, 18)
    assert CC.region_percent_coords(50, 30, 33, 0, 33.5, 100) == (17, 0, 17, 30)      # 16.5 and 16.75 -> 17: equal corners stay (`<`, not `<=`)
    assert CC.region_percent_coords(50, 30, -5, -50, 100, 100) == (-3, -15, 50, 30)   # -2.5 -> -3 (half away from zero)
    assert CC.region_percent_coords(3, 3, 0, 0, 0.1, 0.1) == (0, 0, 0, 0)
    assert CC.region_percent_coords(10, 10, 60, 0, 50, 100)[2] == 7                    # inverted side: one pixel
    b = Bitmap(torch.zeros((1, 4 * 64), dtype=torch.uint8), 8, 4, 64)
    for bad in ((2, 0, 2, 4), (0, 3, 8, 3), (5, 0, 1, 4)):
        with pytest.raises(FlowError):
            CC.region(b, *bad, 0)
    with pytest.raises(FlowError):
        CC.region_percent(b, 50, 0, 50, 100, 0)


def test_copy_rect_to_canvas_refuses_rectangles_outside_either_bitmap():
    """CopyRectNodeDef::render (clone_crop_fill_expand.rs:44-60): a FlowError, before any device call."""
    a = Bitmap(torch.zeros((1, 4 * 64), dtype=torch.uint8), 8, 4, 64)
    c = Bitmap(torch.zeros((1, 4 * 64), dtype=torch.uint8), 6, 4, 64)
    for args in ((8, 0, 1, 1, 0, 0), (0, 4, 1, 1, 0, 0), (4, 0, 5, 1, 0, 0), (0, 0, 7, 1, 0, 0), (0, 0, 2, 2, 5, 0), (0, 0, 2, 2, 0, 3)):
        with pytest.raises(FlowError, match="Invalid coordinates"):
            CC.copy_rect_to_canvas(a, c, *args)
