"""Mirror of flow/nodes/rotate_flip_transpose.rs: how apply_orientation / rotate_* decompose into the two flips and
the transpose (:51-66, :230-309), executed on a device-resident batch.  Functions return the resulting Bitmap (the
transposing ones allocate the swapped canvas like TransposeDef::expand :99-114, transparent = zero filled)."""
from ...graphics.bitmaps import Bitmap, BitmapCompositing
from ...graphics import bitmap_ops as G


def flip_v(b: Bitmap) -> Bitmap:
    G.flow_bitmap_bgra_flip_vertical_safe(b)
    return b


def flip_h(b: Bitmap) -> Bitmap:
    G.flow_bitmap_bgra_flip_horizontal_safe(b)
    return b


def transpose(b: Bitmap) -> Bitmap:
    canvas = Bitmap.create_u8(b.n, b.h, b.w, b.data.device, alpha_meaningful=b.alpha_meaningful,
                              compose=BitmapCompositing.ReplaceSelf)
    G.bitmap_window_transpose(b, canvas)
    return canvas


def rotate_90(b: Bitmap) -> Bitmap:         # :252-256  [FLIP_V, TRANSPOSE]
    return transpose(flip_v(b))


def rotate_270(b: Bitmap) -> Bitmap:        # :279-283  [TRANSPOSE, FLIP_V]
    return flip_v(transpose(b))


def rotate_180(b: Bitmap) -> Bitmap:        # :300-309  [FLIP_V, FLIP_H]
    return flip_h(flip_v(b))


def apply_orientation(b: Bitmap, flag: int) -> Bitmap:
    """ApplyOrientationDef::expand (:51-66): EXIF orientation flag 1..8; anything else is a no-op."""
    if flag == 7:
        return transpose(rotate_180(b))
    if flag == 8:
        return rotate_270(b)
    if flag == 6:
        return rotate_90(b)
    if flag == 5:
        return transpose(b)
    if flag == 4:
        return flip_v(b)
    if flag == 3:
        return rotate_180(b)
    if flag == 2:
        return flip_h(b)
    return b


def oriented_size(w, h, flag):              # estimate (:30-39)
    return (h, w) if 5 <= flag <= 8 else (w, h)
