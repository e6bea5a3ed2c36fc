"""Mirror of flow/nodes/clone_crop_fill_expand.rs + create_canvas.rs on device-resident batches:
crop = a window (Bitmap::crop, bitmaps.rs:841-859: no bytes move), clone / expand_canvas / copy_rect_to_canvas =
create_canvas + graphics::copy_rect, fill_rect = BitmapWindowMut::fill_rectangle."""
from ...errors import ErrorKind, FlowError
from ...graphics.bitmaps import Bitmap, BitmapCompositing
from ...graphics import bitmap_ops as G


def create_canvas(n, w, h, device, color32=0, bgra32=True) -> Bitmap:
    """CreateCanvasDef::execute (create_canvas.rs:77-103): transparent -> ReplaceSelf and zero fill, any other colour
    -> BlendWithMatte(colour), pre-filled with it (bitmaps.rs:829-837)."""
    transparent = (color32 >> 24) == 0
    compose = BitmapCompositing.ReplaceSelf if transparent else BitmapCompositing.BlendWithMatte
    return Bitmap.create_u8(n, w, h, device, alpha_meaningful=bgra32, compose=compose, matte=0 if transparent else color32)


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
        raise FlowError(ErrorKind.InvalidNodeParams, f"Invalid coordinates. Canvas is {canvas.w}x{canvas.h}, Input is {input.w}x{input.h}")
    G.copy_rectangle(input, canvas, from_x, from_y, x, y, w, h)
    return canvas


def clone(b: Bitmap) -> Bitmap:
    """CloneDef::expand (:151-181)."""
    canvas = create_canvas(b.n, b.w, b.h, b.data.device, 0, b.alpha_meaningful)
    return copy_rect_to_canvas(b, canvas, 0, 0, b.w, b.h, 0, 0)


def expand_canvas(b: Bitmap, left, top, right, bottom, color32) -> Bitmap:
    """ExpandCanvasDef::expand (:224-262): the canvas is Bgra32 unless the colour is opaque."""
    opaque = (color32 >> 24) == 255
    canvas = create_canvas(b.n, b.w + left + right, b.h + top + bottom, b.data.device, color32,
                           b.alpha_meaningful if opaque else True)
    return copy_rect_to_canvas(b, canvas, 0, 0, b.w, b.h, left, top)


def fill_rect(b: Bitmap, x1, y1, x2, y2, color32) -> Bitmap:
    """FillRectNodeDef::mutate (:107-137): the bitmap becomes BlendWithSelf first (:112), so a matte canvas accepts a
    sub-rectangle."""
    b.compose = BitmapCompositing.BlendWithSelf
    G.fill_rectangle(b, color32, x1, y1, x2, y2)
    return b
