"""Mirror of flow/nodes/clone_crop_fill_expand.rs + create_canvas.rs on device-resident batches:
crop = a window (Bitmap::crop, bitmaps.rs:841-859: no bytes move), clone / expand_canvas / copy_rect_to_canvas =
create_canvas + graphics::copy_rect, fill_rect = BitmapWindowMut::fill_rectangle."""
from ...errors import ErrorKind, FlowError
from ...graphics.bitmaps import Bitmap, BitmapCompositing
from ...graphics import bitmap_ops as G


def create_canvas(n, w, h, device, color32=0, bgra32=True, keyword_transparent=None) -> Bitmap:
    """CreateCanvasDef::execute (create_canvas.rs:77-103): the enum value Color::Transparent -> ReplaceSelf and zero fill,
    any other colour -- also an srgb colour whose alpha is 0 -- -> BlendWithMatte(colour), pre-filled with it unless it is
    transparent (bitmaps.rs:829-837).  keyword_transparent: the JSON colour was the enum value (None: decide by the
    colour's alpha, for callers that only hold a Color32)."""
    if keyword_transparent is None:
        keyword_transparent = (color32 >> 24) == 0
    compose = BitmapCompositing.ReplaceSelf if keyword_transparent else BitmapCompositing.BlendWithMatte
    return Bitmap.create_u8(n, w, h, device, alpha_meaningful=bgra32, compose=compose, matte=0 if keyword_transparent else color32)


def crop(b: Bitmap, x1, y1, x2, y2) -> Bitmap:
    """CropMutNodeDef::execute (:519-541) -> Bitmap::crop: a window onto the same frames."""
    if x2 <= x1 or y2 <= y1 or x2 > b.w or y2 > b.h:
        raise FlowError(ErrorKind.InvalidArgument, f"Invalid crop bounds (({x1}, {y1}), ({x2}, {y2})) (image {b.w}x{b.h})")
    first = y1 * b.stride + 4 * x1
    last = (y2 - 1) * b.stride + 4 * x2
    return Bitmap(b.data[:, first:last], x2 - x1, y2 - y1, b.stride, b.alpha_meaningful, b.compose, b.matte)


def copy_rect_to_canvas(input: Bitmap, canvas: Bitmap, from_x, from_y, w, h, x, y) -> Bitmap:
    """CopyRectNodeDef::render (:31-90)."""
    if (input.w <= from_x or input.h <= from_y or input.w < from_x + w or input.h < from_y + h
            or canvas.w < x + w or canvas.h < y + h):
        raise FlowError(ErrorKind.InvalidArgument, f"InvalidNodeParams: Invalid coordinates. Canvas is {canvas.w}x{canvas.h}, Input is {input.w}x{input.h}")
    G.copy_rectangle(input, canvas, from_x, from_y, x, y, w, h)
    return canvas


def clone(b: Bitmap) -> Bitmap:
    """CloneDef::expand (:151-181)."""
    canvas = create_canvas(b.n, b.w, b.h, b.data.device, 0, b.alpha_meaningful)
    return copy_rect_to_canvas(b, canvas, 0, 0, b.w, b.h, 0, 0)


def expand_canvas(b: Bitmap, left, top, right, bottom, color32, keyword_transparent=None) -> Bitmap:
    """ExpandCanvasDef::expand (:224-262): the canvas is Bgra32 unless the colour is opaque."""
    opaque = (color32 >> 24) == 255
    canvas = create_canvas(b.n, b.w + left + right, b.h + top + bottom, b.data.device, color32,
                           b.alpha_meaningful if opaque else True, keyword_transparent)
    return copy_rect_to_canvas(b, canvas, 0, 0, b.w, b.h, left, top)


def fill_rect(b: Bitmap, x1, y1, x2, y2, color32) -> Bitmap:
    """FillRectNodeDef::mutate (:107-137): the bitmap becomes BlendWithSelf first (:112), so a matte canvas accepts a
    sub-rectangle."""
    b.compose = BitmapCompositing.BlendWithSelf
    if x2 <= x1 or y2 <= y1 or x1 >= 1 << 31 or y1 >= 1 << 31 or x2 > b.w or y2 > b.h:      # the node's own check (:114-127): an empty
        raise FlowError(ErrorKind.InvalidArgument, f"InvalidCoordinates: Invalid coordinates for {b.w}x{b.h} bitmap")     # rectangle is an error here
    G.fill_rectangle(b, color32, x1, y1, x2, y2)
    return b


def region_percent_coords(w, h, left, top, right, bottom):
    """RegionPercentDef::get_coords (:265-286): `(side as f32 * pct / 100f32).round() as i32` -- f32 arithmetic, half away
    from zero -- and a side the percentages invert gets one pixel (`x2 < x1`; equal corners stay equal)."""
    import numpy as np

    def px(side, pct):
        v = np.float32(side) * np.float32(pct) / np.float32(100)
        r = float(np.copysign(np.floor(np.abs(v) + np.float32(0.5)), v))
        return int(max(-2 ** 31, min(2 ** 31 - 1, r)))
    x1, y1, x2, y2 = px(w, left), px(h, top), px(w, right), px(h, bottom)
    if x2 < x1:
        x2 = x1 + 1
    if y2 < y1:
        y2 = y1 + 1
    return x1, y1, x2, y2


def region(b: Bitmap, x1, y1, x2, y2, color32, keyword_transparent=None) -> Bitmap:
    """RegionDef::expand (:390-452): Crop to the part of the rectangle inside the frame, then ExpandCanvas by the rest; a
    rectangle that misses the frame is a canvas of the colour in the parent's format."""
    if y2 <= y1 or x2 <= x1:
        raise FlowError(ErrorKind.InvalidArgument, f"InvalidNodeParams: Invalid coordinates: {x1},{y1} {x2},{y2} should describe the top-left and "
                        "bottom-right corners of the region in pixels. Not a rectangle.")
    if x1 >= b.w or y1 >= b.h or x2 <= 0 or y2 <= 0:
        return create_canvas(b.n, x2 - x1, y2 - y1, b.data.device, color32, b.alpha_meaningful, keyword_transparent)
    part = crop(b, min(b.w, max(0, x1)), min(b.h, max(0, y1)), min(b.w, max(0, x2)), min(b.h, max(0, y2)))
    return expand_canvas(part, max(0, -x1), max(0, -y1), max(0, x2 - b.w), max(0, y2 - b.h), color32, keyword_transparent)


def region_percent(b: Bitmap, left, top, right, bottom, color32, keyword_transparent=None) -> Bitmap:
    """RegionPercentDef::expand (:316-352)."""
    if bottom <= top or right <= left:
        raise FlowError(ErrorKind.InvalidArgument, f"InvalidNodeParams: Invalid coordinates: {left},{top} {right},{bottom} should describe the top-left "
                        "and bottom-right corners of the region in percentages. Not a rectangle.")
    return region(b, *region_percent_coords(b.w, b.h, left, top, right, bottom), color32, keyword_transparent)
