"""Jobs through the OUTER boundary (libimageflow C-ABI subset + v1/build / v1/execute JSON, csrc/abi_shim.cpp) on the GPU:
  * the reference's own synthetic-canvas tests, sent as the JSON the reference's tests build (visuals/canvas.rs:8-90),
    hash to the checksums the reference stored (canvas.checksums) -- end to end through the ABI;
  * BASELINE config 1 (querystring `width=200` on a 4K JPEG), config 3 in its graph form (export_4_sizes) and the plain
    decode -> resample_2d job equal the CPU oracle chain byte for byte."""
import io

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.abi import Context, pack_raw_bgra, unpack_raw_bgra  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import util as U  # noqa: E402
from tests.seahash import bitmap_checksum, checksum_id_digits  # noqa: E402


def _pixels(buf):
    rows, w, h, alpha = unpack_raw_bgra(buf)
    return rows[:, :4 * w].reshape(h, w, 4), alpha


def _run(ctx, method, job, expect=200):
    status, r = ctx.send_json(method, job)
    assert status == expect, (status, r, ctx.error_message())
    return r


FILL = [{"create_canvas": {"w": 200, "h": 200, "format": "bgra_32", "color": "transparent"}},
        {"fill_rect": {"x1": 0, "y1": 0, "x2": 100, "y2": 100, "color": {"srgb": {"hex": "EECCFFFF"}}}}]


@pytest.mark.parametrize("steps,want", [
    (FILL + [{"resample_2d": {"w": 400, "h": 400, "hints": {"down_filter": "hermite", "up_filter": "hermite"}}}], "967914e71e"),
    (FILL + [{"expand_canvas": {"left": 10, "top": 15, "right": 20, "bottom": 25, "color": {"srgb": {"hex": "2233AAFF"}}}},
             {"resample_2d": {"w": 400, "h": 400, "hints": {"down_filter": "hermite", "up_filter": "hermite", "scaling_colorspace": "linear"}}}], "dd2079bbc7"),
    ([{"create_canvas": {"w": 200, "h": 200, "format": "bgra_32", "color": {"srgb": {"hex": "00000000"}}}}], "8cb229c079"),
    ([{"create_canvas": {"w": 400, "h": 300, "format": "bgra_32", "color": "transparent"}},
      {"fill_rect": {"x1": 0, "y1": 0, "x2": 50, "y2": 100, "color": {"srgb": {"hex": "0000FFFF"}}}}], "103bc946d6"),
    ([{"create_canvas": {"w": 200, "h": 200, "format": "bgra_32", "color": {"srgb": {"hex": "FF5555FF"}}}},
      {"fill_rect": {"x1": 0, "y1": 0, "x2": 10, "y2": 100, "color": {"srgb": {"hex": "0000FFFF"}}}},
      {"crop": {"x1": 0, "y1": 50, "x2": 100, "y2": 100}}], "e833f82320"),
])
def test_reference_canvas_jobs_hash_to_the_reference_checksums(steps, want):
    with Context() as c:
        c.add_output_buffer(1)
        r = _run(c, "v1/execute", {"framewise": {"steps": steps + [{"encode": {"io_id": 1, "preset": {"lodepng": {"maximum_deflate": False}}}}]}})
        px, _ = _pixels(c.get_output_buffer(1))
        assert r["data"]["job_result"]["encodes"][0]["w"] == px.shape[1]
        assert checksum_id_digits(px) == want, bitmap_checksum(px)


def _jpeg(w, h, seed=3, quality=85, subsampling="4:2:0"):
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 255 // max(w + h - 2, 1))], -1).astype(np.int16)
    img = np.clip(img + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    PIL.fromarray(img).save(b, "JPEG", quality=quality, subsampling=subsampling, optimize=False)
    return b.getvalue()


def _oracle_resize(rows, w, h, ow, oh, **kw):
    can = np.zeros((oh, U.stride_for(ow)), np.uint8)
    rc, _ = O.scale_and_render(np.ascontiguousarray(rows), w, h, can, ow, oh, 0, 0, ow, oh, **kw)
    assert rc == 0
    return can


def test_decode_resample_job_equals_the_oracle_chain():
    data = _jpeg(1000, 700)
    with Context() as c:
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        r = _run(c, "v1/build", {"io": [{"io_id": 0, "direction": "in", "io": "placeholder"}, {"io_id": 1, "direction": "out", "io": "placeholder"}],
                                 "framewise": {"steps": [{"decode": {"io_id": 0}},
                                                         {"resample_2d": {"w": 200, "h": 140, "hints": {"down_filter": "robidoux", "scaling_colorspace": "linear", "sharpen_percent": 0}}},
                                                         {"encode": {"io_id": 1, "preset": {"lodepng": {"maximum_deflate": False}}}}]}})
        assert r["data"]["build_result"]["decodes"][0]["w"] == 1000
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
    j = O.jpeg_read_coefficients(data)
    exp = _oracle_resize(O.jpeg_idct_color(j), 1000, 700, 200, 140, filter_id=2)
    assert (w, h, alpha) == (200, 140, False) and np.array_equal(rows, exp)


def test_config1_querystring_width_200_on_a_4k_jpeg():
    """BASELINE cfg1: command_string width=200.  3840x2160 -> target 200x113; pre-shrink hint as ir4/mod.rs:155-210
    (min(3840/200, 2160/200) = 10.8 -> 2.1/10.8 -> 746x420 -> scale_num 2 -> 960x540, luma through
    flow_scale_spatial_srgb_2x2), then Robidoux 960x540 -> 200x113 in linear light."""
    data = _jpeg(3840, 2160)
    with Context() as c:
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        _run(c, "v1/build", {"io": [{"io_id": 0, "direction": "in", "io": "placeholder"}, {"io_id": 1, "direction": "out", "io": "placeholder"}],
                             "framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=200", "decode": 0, "encode": 1}}]}})
        rows, w, h, _ = unpack_raw_bgra(c.get_output_buffer(1))
    assert (w, h) == (200, 113)
    j = O.jpeg_read_coefficients(data)
    small = O.jpeg_idct_color_scaled(j, 2, 2)
    exp = _oracle_resize(small, 960, 540, 200, 113, filter_id=2)
    assert np.array_equal(rows, exp)


def test_config3_graph_form_four_outputs():
    """export_4_sizes as a graph (composition.rs:400-482 shape): decode -> constrain 1600 -> {constrain 1200 -> constrain 400,
    constrain 800}, one encode per size."""
    src = U.random_frames(1, 3840, 2160, seed0=9, alpha=False)[0]
    nodes = {"0": {"decode": {"io_id": 0}},
             "1": {"constrain": {"mode": "within", "w": 1600}}, "2": {"constrain": {"mode": "within", "w": 1200}},
             "3": {"constrain": {"mode": "within", "w": 800}}, "4": {"constrain": {"mode": "within", "w": 400}},
             "5": {"encode": {"io_id": 1, "preset": "gif"}}, "6": {"encode": {"io_id": 2, "preset": "gif"}},
             "7": {"encode": {"io_id": 3, "preset": "gif"}}, "8": {"encode": {"io_id": 4, "preset": "gif"}}}
    edges = [{"from": a, "to": b, "kind": "input"} for a, b in ((0, 1), (1, 2), (1, 3), (2, 4), (1, 5), (2, 6), (3, 7), (4, 8))]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 3840, 2160, alpha_meaningful=False))
        for i in (1, 2, 3, 4):
            c.add_output_buffer(i)
        r = _run(c, "v1/execute", {"framewise": {"graph": {"nodes": nodes, "edges": edges}}})
        assert len(r["data"]["job_result"]["encodes"]) == 4
        outs = {i: unpack_raw_bgra(c.get_output_buffer(i)) for i in (1, 2, 3, 4)}
    l1600 = _oracle_resize(src, 3840, 2160, 1600, 900)
    l1200 = _oracle_resize(l1600, 1600, 900, 1200, 675)
    l800 = _oracle_resize(l1600, 1600, 900, 800, 450)
    l400 = _oracle_resize(l1200, 1200, 675, 400, 225)
    for i, exp in ((1, l1600), (2, l1200), (3, l800), (4, l400)):
        rows, w, h, _ = outs[i]
        assert np.array_equal(rows, exp), i


def test_raw_container_round_trip_and_matte_hint():
    src = U.random_frames(1, 333, 211, seed0=21, alpha=True)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 333, 211, alpha_meaningful=True))
        c.add_output_buffer(1)
        c.add_output_buffer(2)
        _run(c, "v1/execute", {"framewise": {"graph": {
            "nodes": {"0": {"decode": {"io_id": 0}}, "1": {"encode": {"io_id": 1, "preset": "gif"}},
                      "2": {"resample_2d": {"w": 40, "h": 25, "hints": {"down_filter": "lanczos", "sharpen_percent": 15,
                                                                       "background_color": {"srgb": {"hex": "FFFFFFFF"}}}}},
                      "3": {"encode": {"io_id": 2, "preset": "gif"}}},
            "edges": [{"from": 0, "to": 1, "kind": "input"}, {"from": 0, "to": 2, "kind": "input"}, {"from": 2, "to": 3, "kind": "input"}]}}})
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
        assert (w, h, alpha) == (333, 211, True) and np.array_equal(rows[:, :4 * w], src[:, :4 * w])
        rows2, w2, h2, alpha2 = unpack_raw_bgra(c.get_output_buffer(2))
    can = np.zeros((25, U.stride_for(40)), np.uint8)
    can[:, :160] = 255                                                          # BlendWithMatte canvases start filled (bitmaps.rs:829-837)
    rc, _ = O.scale_and_render(np.ascontiguousarray(src), 333, 211, can, 40, 25, 0, 0, 40, 25, filter_id=6, sharpen=15.0,
                               compositing=O.BLEND_WITH_MATTE, matte_bgra=0xFFFFFFFF, alpha_meaningful=True)
    assert rc == 0 and (w2, h2, alpha2) == (40, 25, False) and np.array_equal(rows2, can)


def test_progressive_jpeg_is_refused_not_mis_decoded():
    PIL = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    PIL.fromarray(np.zeros((32, 32, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with Context() as c:
        c.add_input_buffer(0, b.getvalue())
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}]}})
        assert status == 400 and c.error_code() == 5 and "ImageTypeNotSupported" in r["message"]


@pytest.mark.parametrize("size,quality", [((1, 1), 75), ((17, 9), 90), ((203, 131), None), ((640, 427), 60)])
def test_libjpeg_turbo_preset_writes_the_file_libjpeg_turbo_writes(size, quality):
    """decode(raw frame) -> encode(libjpeg_turbo): matte, colour conversion, down-sampling, DCT and quantisation on the GPU,
    markers + Huffman coding on the host -- byte for byte the file libjpeg-turbo (Pillow, same 4:2:0 sampling, standard
    tables) writes from the same pixels.  Default quality is the reference's 75 (codecs/mozjpeg.rs:32)."""
    Image = pytest.importorskip("PIL.Image")
    w, h = size
    rng = np.random.default_rng(w * 7 + h)
    y, x = np.mgrid[0:h, 0:w]
    rgb = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x * 5 + y * 3) % 256)], -1).astype(np.int32)
    rgb = np.clip(rgb + rng.integers(-30, 30, rgb.shape), 0, 255).astype(np.uint8)
    bgra = np.zeros((h, U.stride_for(w)), np.uint8)
    bgra[:, :4 * w] = np.concatenate([rgb[:, :, ::-1], np.full((h, w, 1), 255, np.uint8)], -1).reshape(h, 4 * w)
    preset = {"libjpeg_turbo": {} if quality is None else {"quality": quality}}
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(bgra, w, h, alpha_meaningful=False))
        c.add_output_buffer(1)
        r = _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"encode": {"io_id": 1, "preset": preset}}]}})
        got = bytes(c.get_output_buffer(1))
    enc = r["data"]["job_result"]["encodes"][0]
    assert (enc["preferred_mime_type"], enc["preferred_extension"], enc["w"], enc["h"]) == ("image/jpeg", "jpg", w, h)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, "JPEG", quality=75 if quality is None else quality, subsampling="4:2:0", optimize=False)
    assert got == buf.getvalue()


def test_libjpeg_turbo_preset_flattens_alpha_onto_the_matte_first():
    """mozjpeg.rs:88-94: apply_matte(matte or white) before compression; the flattened pixels come from the oracle."""
    Image = pytest.importorskip("PIL.Image")
    w, h = 97, 61
    fr = U.random_frames(1, w, h, seed0=11, alpha=True)
    stride = fr.shape[2]
    for matte_json, matte32 in ((None, 0xFFFFFFFF), ({"srgb": {"hex": "336699FF"}}, 0xFF336699)):
        flat = fr[0].copy()
        O.apply_matte(flat, w, h, stride, matte32, True)
        rgb = np.ascontiguousarray(flat[:, :4 * w].reshape(h, w, 4)[:, :, 2::-1])
        params = {"quality": 85}
        if matte_json:
            params["matte"] = matte_json
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(fr[0], w, h, alpha_meaningful=True))
            c.add_output_buffer(1)
            _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": params}}}]}})
            got = bytes(c.get_output_buffer(1))
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
        assert got == buf.getvalue()


@pytest.mark.parametrize("extra,pillow", [({"progressive": True}, {"progressive": True}),
                                          ({"optimize_huffman_coding": True}, {"optimize": True}),
                                          ({"progressive": True, "optimize_huffman_coding": True}, {"progressive": True, "optimize": True})])
def test_libjpeg_turbo_preset_progressive_and_optimised_tables(extra, pillow):
    """mozjpeg.rs:121-129 set_progressive_mode / set_optimize_coding over set_fastest_defaults: the files equal
    libjpeg-turbo's (Pillow) byte for byte -- optimal Huffman tables, the standard scan script, end-of-band runs."""
    PIL = pytest.importorskip("PIL.Image")
    from PIL import ImageFile
    ImageFile.MAXBLOCK = 1 << 24
    w, h = 203, 131
    rng = np.random.default_rng(7)
    y, x = np.mgrid[0:h, 0:w]
    rgb = np.stack([x * 255 // (w - 1), y * 255 // (h - 1), (x + y) * 3 % 256], -1).astype(np.int32)
    rgb = np.clip(rgb + rng.integers(-30, 30, rgb.shape), 0, 255).astype(np.uint8)
    bgra = np.zeros((h, U.stride_for(w)), np.uint8)
    bgra[:, :4 * w] = np.concatenate([rgb[:, :, ::-1], np.full((h, w, 1), 255, np.uint8)], -1).reshape(h, 4 * w)
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(bgra, w, h, alpha_meaningful=False))
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}},
                                                        {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": dict(quality=88, **extra)}}}]}})
        got = bytes(c.get_output_buffer(1))
    buf = io.BytesIO()
    PIL.fromarray(rgb).save(buf, "JPEG", quality=88, subsampling="4:2:0", **pillow)
    assert got == buf.getvalue()
