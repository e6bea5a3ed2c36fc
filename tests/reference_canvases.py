"""The reference's visual tests whose inputs are synthetic (no S3 image needed), restated as (source, job) pairs, with
the checksum ids the reference stores for their outputs.  Used by the CPU oracle test and by the GPU parity test.

test_fill_rect       imageflow_core/tests/integration/visuals/canvas.rs:8-33   id canvas.checksums:71-73
test_expand_rect     visuals/canvas.rs:53-90                                   id canvas.checksums:51-53
test_trim_then_resize visuals/trim.rs:131-158                                  id trim.checksums (trimmed_then_300x300)
Node semantics: create_canvas Bgra32 keeps alpha meaningful, fill_rect writes the colour bytes (B,G,R,A of 0xAARRGGBB),
expand_canvas pads with the colour, crop_whitespace(threshold 80) of an orange square on white leaves the square,
resample_2d without background colour renders with ReplaceSelf in linear light (scale_render.rs:95-112,276-290).
"""
import numpy as np

HERMITE, ROBIDOUX = 16, 2


def _fill_source():
    src = np.zeros((200, 200, 4), np.uint8)
    src[:100, :100] = [0xFF, 0xCC, 0xEE, 0xFF]          # "EECCFFFF" = R EE, G CC, B FF, A FF -> bytes B,G,R,A
    return src


def _expand_source():
    src = np.zeros((200 + 15 + 25, 200 + 10 + 20, 4), np.uint8)
    src[:] = [0xAA, 0x33, 0x22, 0xFF]                   # "2233AAFF"
    src[15:215, 10:210] = _fill_source()
    return src


def _trim_source():
    src = np.zeros((100, 100, 4), np.uint8)
    src[:] = [0x00, 0x55, 0xFF, 0xFF]                   # "FF5500FF", what crop_whitespace leaves of the white canvas
    return src


# name -> (source BGRA [h][w][4], out_w, out_h, filter id, reference checksum id digits)
RESAMPLE_CASES = {
    "test_fill_rect eeccff_hermite_400x400": (_fill_source, 400, 400, HERMITE, "967914e71e"),
    "test_expand_rect fill_expand_hermite_linear": (_expand_source, 400, 400, HERMITE, "dd2079bbc7"),
    "test_trim_then_resize trimmed_then_300x300": (_trim_source, 300, 300, ROBIDOUX, "a185811359"),
}


def calibration_canvases():
    """Canvases the reference checksums WITHOUT any resampling: they fix the hash layout and the id format."""
    z = np.zeros((200, 200, 4), np.uint8)                                    # canvas.rs:146-158
    b = np.zeros((300, 400, 4), np.uint8)
    b[:100, :50] = [255, 0, 0, 255]                                          # canvas.rs:35-50
    c = np.zeros((50, 100, 4), np.uint8)
    c[:] = [0x55, 0x55, 0xFF, 0xFF]
    c[:, :10] = [255, 0, 0, 255]                                             # canvas.rs:92-114 (crop 0,50..100,100)
    return {"test_transparent_canvas 200x200": (z, "8cb229c079"),
            "test_fill_rect_original blue_on_transparent": (b, "103bc946d6"),
            "test_crop red_canvas_blue_strip": (c, "e833f82320")}
