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
        assert c.L.ifhip_shim_fused_decode_resamples(c.p) == 1          # decode + resample ran as one call: no 960x540 bitmap in HBM
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
    # the canvas keeps parent.fmt (Bgra32): an opaque matte does not clear alpha_meaningful on this path (scale_render.rs:96-105)
    assert rc == 0 and (w2, h2, alpha2) == (40, 25, True) and np.array_equal(rows2, can)


@pytest.mark.parametrize("subsampling", ["4:2:0", "4:4:4", "4:2:2"])
@pytest.mark.parametrize("size", [(32, 32), (203, 131), (640, 427)])
def test_progressive_jpeg_decodes_to_libjpeg_turbos_pixels(size, subsampling):
    """What the reference's mozjpeg preset writes (codecs/mozjpeg.rs:121-123: progressive): the scans are decoded on the host
    (csrc/jpeg_read.cpp, jdphuff.c's four block decoders), the pixel stage is the GPU's -- the job's decode equals
    libjpeg-turbo's decode of the same file byte for byte, as for baseline files."""
    PIL = pytest.importorskip("PIL.Image")
    w, h = size
    rng = np.random.default_rng(w + h)
    y, x = np.mgrid[0:h, 0:w]
    img = np.clip(np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1) +
                  (30 * np.sin(x / 3.0) * np.cos(y / 5.0))[..., None] + rng.integers(-20, 21, (h, w, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    PIL.fromarray(img).save(b, "JPEG", quality=80, progressive=True, subsampling=subsampling)
    data = b.getvalue()
    assert b"\xff\xc2" in data
    with Context() as c:
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        status, info = c.send_json("v1/get_image_info", {"io_id": 0})
        assert status == 200 and (info["data"]["image_info"]["image_width"], info["data"]["image_info"]["image_height"]) == (w, h)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        rows, ow, oh, alpha = unpack_raw_bgra(c.get_output_buffer(1))
    ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
    assert (ow, oh, alpha) == (w, h, False)
    got = rows[:, :4 * w].reshape(h, w, 4)
    assert np.array_equal(got[:, :, 2::-1], ref)


def test_arithmetic_coded_jpeg_is_refused_not_mis_decoded():
    """SOF9 (arithmetic coding): ImageTypeNotSupported, category 5 -- never a wrong picture."""
    PIL = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    PIL.fromarray(np.zeros((32, 32, 3), np.uint8)).save(b, "JPEG")
    data = bytearray(b.getvalue())
    at = data.index(b"\xff\xc0")
    data[at + 1] = 0xC9
    with Context() as c:
        c.add_input_buffer(0, bytes(data))
        status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}]}})
        assert status == 400 and c.error_code() == 5 and "ImageTypeNotSupported" in r["message"]


@pytest.mark.parametrize("size,quality", [((1, 1), 75), ((17, 9), 90), ((203, 131), None), ((640, 427), 60)])
def test_libjpeg_turbo_preset_writes_the_file_libjpeg_turbo_writes(size, quality):
    """decode(raw frame) -> encode(libjpeg_turbo): matte, colour conversion, down-sampling, DCT and quantisation on the GPU,
    Huffman coding and markers on the GPU too (the preset's default: baseline, Annex K tables) -- byte for byte the file libjpeg-turbo (Pillow, same 4:2:0 sampling, standard
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
        assert c.L.ifhip_shim_device_coded_files(c.p) == 1          # the Huffman coder ran on the device: only the file was downloaded
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
        assert c.L.ifhip_shim_device_coded_files(c.p) == 0          # these options stay with the host writer
    buf = io.BytesIO()
    PIL.fromarray(rgb).save(buf, "JPEG", quality=88, subsampling="4:2:0", **pillow)
    assert got == buf.getvalue()


# ---- round 3: the compose nodes at the boundary the reference exposes, the node semantics ADVICE flagged ---------------
def _canvas_rows(w, h, fill=None):
    can = np.zeros((h, U.stride_for(w)), np.uint8)
    if fill is not None:
        can[:, :4 * w] = np.tile(np.array(fill, np.uint8), w)
    return can


def test_same_size_and_mixed_ratio_resamples_pick_the_filter_like_the_reference():
    """scale_render.rs:253-261: `upscaling = w > in.w || h > in.h` picks up_filter (Ginseng), everything else -- same
    size, or smaller in both -- down_filter (Robidoux).  Same-size + sharpen must blur/sharpen with Robidoux, and a
    wider-but-shorter target must use the UP filter."""
    src = U.random_frames(1, 120, 90, seed0=5, alpha=False)[0]

    def run(w, h, hints):
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 120, 90, alpha_meaningful=False))
            c.add_output_buffer(1)
            _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"resample_2d": {"w": w, "h": h, "hints": hints}},
                                                            {"encode": {"io_id": 1, "preset": "gif"}}]}})
            return unpack_raw_bgra(c.get_output_buffer(1))[0]
    same = run(120, 90, {"sharpen_percent": 30})
    assert np.array_equal(same, _oracle_resize(src, 120, 90, 120, 90, filter_id=2, sharpen=30.0))            # Robidoux, not Ginseng
    assert not np.array_equal(same, _oracle_resize(src, 120, 90, 120, 90, filter_id=4, sharpen=30.0))
    always = run(120, 90, {"resample_when": "always"})
    assert np.array_equal(always, _oracle_resize(src, 120, 90, 120, 90, filter_id=2))
    mixed = run(200, 45, {})                                                                                   # wider, shorter
    assert np.array_equal(mixed, _oracle_resize(src, 120, 90, 200, 45, filter_id=4))                          # Ginseng
    mixed2 = run(200, 45, {"up_filter": "hermite", "down_filter": "box"})
    assert np.array_equal(mixed2, _oracle_resize(src, 120, 90, 200, 45, filter_id=16))
    # sharpen_when / resample_when gates (scale_render.rs:55-78): sharpening asked for down-scaling only, size unchanged,
    # default resample_when -> the node disappears
    assert np.array_equal(run(120, 90, {"sharpen_percent": 30, "sharpen_when": "downscaling"})[:, :480], src[:, :480])
    assert np.array_equal(run(120, 90, {"sharpen_percent": 30, "resample_when": "size_differs"})[:, :480], src[:, :480])
    assert np.array_equal(run(60, 45, {"sharpen_percent": 30, "sharpen_when": "upscaling"}), _oracle_resize(src, 120, 90, 60, 45, filter_id=2))


def test_same_size_matte_job_on_a_bgra_parent():
    """scale_render.rs:44-47,80: a Bgra32 parent with a background_color is resampled even at the same size (matte)."""
    src = U.random_frames(1, 64, 48, seed0=6, alpha=True)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 64, 48, alpha_meaningful=True))
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}},
                                                        {"resample_2d": {"w": 64, "h": 48, "hints": {"background_color": {"srgb": {"hex": "336699"}}}}},
                                                        {"encode": {"io_id": 1, "preset": "gif"}}]}})
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
    can = _canvas_rows(64, 48, (0x99, 0x66, 0x33, 0xFF))
    rc, _ = O.scale_and_render(src, 64, 48, can, 64, 48, 0, 0, 64, 48, filter_id=2, compositing=O.BLEND_WITH_MATTE, matte_bgra=0xFF336699, alpha_meaningful=True)
    # the canvas keeps parent.fmt = Bgra32 (scale_render.rs:96-105); nothing on this path clears alpha_meaningful after
    # an opaque matte (only the encoder-side Bitmap::apply_matte does, bitmaps.rs:528-541)
    assert rc == 0 and alpha and np.array_equal(rows, can)


def _graph(nodes, edges):
    return {"framewise": {"graph": {"nodes": {str(k): v for k, v in nodes.items()},
                                    "edges": [{"from": a, "to": b, "kind": k} for a, b, k in edges]}}}


def test_graph_draw_image_exact_matches_the_oracle_chain():
    """visuals/composition.rs:175-230 (test_graph_draw_image_exact): overlay (alpha) drawn at (200,200) 100x100 with blend
    compose onto a background that was first resized to 400x400 -- the JSON the reference builds, checked against
    oracle(resize) -> oracle(scale_and_render BlendWithSelf at x,y)."""
    overlay = U.random_frames(1, 160, 120, seed0=31, alpha=True)[0]
    back = U.random_frames(1, 500, 430, seed0=32, alpha=False)[0]
    hints = {"down_filter": "robidoux", "up_filter": "robidoux"}                      # ResampleHints::with_bi_filter
    job = _graph({0: {"decode": {"io_id": 0}}, 1: {"decode": {"io_id": 1}}, 2: {"resample_2d": {"w": 400, "h": 400, "hints": hints}},
                  3: {"draw_image_exact": {"x": 200, "y": 200, "w": 100, "h": 100, "blend": "compose", "hints": hints}},
                  4: {"encode": {"io_id": 2, "preset": {"lodepng": {"maximum_deflate": None}}}}},
                 [(1, 2, "input"), (0, 3, "input"), (2, 3, "canvas"), (3, 4, "input")])
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(overlay, 160, 120, alpha_meaningful=True))
        c.add_input_buffer(1, pack_raw_bgra(back, 500, 430, alpha_meaningful=False))
        c.add_output_buffer(2)
        r = _run(c, "v1/execute", job)
        rows, w, h, alpha = unpack_raw_bgra(c.take_output_buffer(2))
        names = [n["name"] for n in r["data"]["job_result"]["performance"]["frames"][0]["nodes"]]
    assert sorted(names) == sorted(["primitive_decoder", "primitive_decoder", "create_canvas", "draw_image_to_canvas", "draw_image_to_canvas", "primitive_encoder"])
    can = _oracle_resize(back, 500, 430, 400, 400, filter_id=2)
    rc, _ = O.scale_and_render(overlay, 160, 120, can, 400, 400, 200, 200, 100, 100, filter_id=2, compositing=O.BLEND_WITH_SELF, alpha_meaningful=True)
    assert rc == 0 and (w, h) == (400, 400) and np.array_equal(rows, can)
    # blend overwrite on THIS canvas changes nothing: the resample that produced it left it BlendWithSelf
    # (scale_render.rs:314), and :284-292 only rewrites ReplaceSelf (compose) and BlendWithMatte (overwrite) canvases
    job["framewise"]["graph"]["nodes"]["3"]["draw_image_exact"]["blend"] = "overwrite"
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(overlay, 160, 120, alpha_meaningful=True))
        c.add_input_buffer(1, pack_raw_bgra(back, 500, 430, alpha_meaningful=False))
        c.add_output_buffer(2)
        _run(c, "v1/execute", job)
        rows2 = unpack_raw_bgra(c.get_output_buffer(2))[0]
    assert np.array_equal(rows2, rows)
    # blend overwrite on a matte canvas from create_canvas (Bgra32): :287-292 turns it into ReplaceSelf -- the rect is
    # replaced, alpha and all; with compose the same canvas blends with its matte
    for blend, mode in (("overwrite", O.REPLACE_SELF), ("compose", O.BLEND_WITH_MATTE)):
        job2 = _graph({0: {"decode": {"io_id": 0}}, 1: {"create_canvas": {"w": 400, "h": 400, "format": "bgra_32", "color": {"srgb": {"hex": "3366CCFF"}}}},
                       3: {"draw_image_exact": {"x": 200, "y": 200, "w": 100, "h": 100, "blend": blend, "hints": hints}},
                       4: {"encode": {"io_id": 2, "preset": {"lodepng": {"maximum_deflate": None}}}}},
                      [(0, 3, "input"), (1, 3, "canvas"), (3, 4, "input")])
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(overlay, 160, 120, alpha_meaningful=True))
            c.add_output_buffer(2)
            _run(c, "v1/execute", job2)
            rows3 = unpack_raw_bgra(c.get_output_buffer(2))[0]
        can3 = _canvas_rows(400, 400, (0xCC, 0x66, 0x33, 0xFF))
        rc, _ = O.scale_and_render(overlay, 160, 120, can3, 400, 400, 200, 200, 100, 100, filter_id=2, compositing=mode,
                                   matte_bgra=0xFF3366CC, alpha_meaningful=True)
        assert rc == 0 and np.array_equal(rows3[:, :1600], can3[:, :1600]), blend
    # a rect outside the canvas is InvalidNodeParams (scale_render.rs:237-240)
    job["framewise"]["graph"]["nodes"]["3"]["draw_image_exact"]["x"] = 350
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(overlay, 160, 120, alpha_meaningful=True))
        c.add_input_buffer(1, pack_raw_bgra(back, 500, 430, alpha_meaningful=False))
        c.add_output_buffer(2)
        status, r = c.send_json("v1/execute", job)
        assert status == 400 and c.error_code() == 2 and "does not fit canvas size 400x400" in r["message"]


def test_canvas_compositing_state_follows_the_reference_from_node_to_node():
    """What a frame's compositing mode is when a later node draws on it (found by tools/fuzz_shim_chains.py, round 6; expected
    pixels by the oracle):
    * copy_rectangle leaves its canvas BlendWithSelf (copy_rect.rs:37), so a draw_image_exact with blend overwrite onto the
      result of expand_canvas COMPOSES (scale_render.rs:284-292 rewrites only ReplaceSelf + compose and BlendWithMatte +
      overwrite);
    * create_canvas makes a ReplaceSelf canvas only for the enum value Transparent: an srgb colour whose alpha is 0 is a
      BlendWithMatte canvas (create_canvas.rs:79-82), and draw_image_exact with blend compose then blends with THAT matte;
    * crop is a window: the frame keeps the mode it had (bitmaps.rs:841-859)."""
    overlay = U.random_frames(1, 55, 104, seed0=71, alpha=True)[0]
    back = U.random_frames(1, 80, 89, seed0=72, alpha=True)[0]
    hints = {"down_filter": "robidoux", "up_filter": "ginseng"}

    def run(canvas_nodes, blend, canvas_input=True):
        nodes = {0: {"decode": {"io_id": 0}}}
        edges = []
        first = 1
        if canvas_input:
            nodes[1] = {"decode": {"io_id": 1}}
            first = 2
        for k, n in enumerate(canvas_nodes):
            nodes[first + k] = n
            if k or canvas_input:
                edges.append((first + k - 1, first + k, "input"))
        last = first + len(canvas_nodes) - 1
        d, e = last + 1, last + 2
        nodes[d] = {"draw_image_exact": {"x": 4, "y": 21, "w": 70, "h": 60, "blend": blend, "hints": hints}}
        nodes[e] = {"encode": {"io_id": 2, "preset": "gif"}}
        edges += [(0, d, "input"), (last, d, "canvas"), (d, e, "input")]
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(overlay, 55, 104, alpha_meaningful=True))
            c.add_input_buffer(1, pack_raw_bgra(back, 80, 89, alpha_meaningful=True))
            c.add_output_buffer(2)
            _run(c, "v1/execute", _graph(nodes, edges))
            return unpack_raw_bgra(c.get_output_buffer(2))

    def oracle_draw(can, cw, ch, mode, matte=0):
        rc, _ = O.scale_and_render(overlay, 55, 104, can, cw, ch, 4, 21, 70, 60, filter_id=4, compositing=mode, matte_bgra=matte, alpha_meaningful=True)
        assert rc == 0
        return can
    # expand_canvas, then overwrite: composes
    rows, w, h, alpha = run([{"expand_canvas": {"left": 13, "top": 9, "right": 20, "bottom": 19, "color": "transparent"}}], "overwrite")
    can = _canvas_rows(113, 117)
    inp = back.copy()
    rc, _ = O.copy_rect(inp, 80, 89, inp.shape[1], True, can, 113, 117, can.shape[1], True, 0, 0, 13, 9, 80, 89)
    assert rc == 0 and (w, h, alpha) == (113, 117, True)
    assert np.array_equal(rows[:, :4 * 113], oracle_draw(can.copy(), 113, 117, O.BLEND_WITH_SELF)[:, :4 * 113])
    assert not np.array_equal(rows[:, :4 * 113], oracle_draw(can.copy(), 113, 117, O.REPLACE_SELF)[:, :4 * 113])
    # a decoded frame, overwrite: replaces (the frame is ReplaceSelf)
    rows, w, h, alpha = run([], "overwrite")
    assert np.array_equal(rows[:, :320], oracle_draw(back.copy(), 80, 89, O.REPLACE_SELF)[:, :320])
    # crop of the expanded frame keeps BlendWithSelf
    rows, w, h, alpha = run([{"expand_canvas": {"left": 13, "top": 9, "right": 20, "bottom": 19, "color": "transparent"}},
                             {"crop": {"x1": 3, "y1": 2, "x2": 100, "y2": 110}}], "overwrite")
    win = np.zeros((108, U.stride_for(97)), np.uint8)
    win[:, :4 * 97] = can[2:110, 12:400]
    assert (w, h) == (97, 108) and np.array_equal(rows[:, :4 * 97], oracle_draw(win.copy(), 97, 108, O.BLEND_WITH_SELF)[:, :4 * 97])
    # ... and a crop of the decoded frame keeps ReplaceSelf
    rows, w, h, alpha = run([{"crop": {"x1": 2, "y1": 1, "x2": 80, "y2": 89}}], "overwrite")
    win = np.zeros((88, U.stride_for(78)), np.uint8)
    win[:, :4 * 78] = back[1:89, 8:320]
    assert (w, h) == (78, 88) and np.array_equal(rows[:, :4 * 78], oracle_draw(win.copy(), 78, 88, O.REPLACE_SELF)[:, :4 * 78])
    # create_canvas with an srgb colour of alpha 0: a matte canvas
    for blend, mode in (("compose", O.BLEND_WITH_MATTE), ("overwrite", O.REPLACE_SELF)):
        rows, w, h, alpha = run([{"create_canvas": {"w": 90, "h": 100, "format": "bgra_32", "color": {"srgb": {"hex": "19BD1200"}}}}], blend, canvas_input=False)
        exp = oracle_draw(_canvas_rows(90, 100), 90, 100, mode, matte=0x0019BD12)
        assert (w, h, alpha) == (90, 100, True) and np.array_equal(rows[:, :360], exp[:, :360]), blend
    rows, w, h, alpha = run([{"create_canvas": {"w": 90, "h": 100, "format": "bgra_32", "color": "transparent"}}], "compose", canvas_input=False)
    assert np.array_equal(rows[:, :360], oracle_draw(_canvas_rows(90, 100), 90, 100, O.BLEND_WITH_SELF)[:, :360])


def test_constrain_crop_and_pad_modes_equal_the_oracle_chain():
    """flow/nodes/constrain.rs:41-98: [Crop] -> Resample2D (canvas_color over hints.background_color) -> [ExpandCanvas], sizes by
    imageflow_riapi's process_constraint (csrc/layout.cpp; tests/test_constraint_layout.py pins the arithmetic).  Expected
    pixels: the oracle on the cropped window, copied into a canvas of the colour."""
    src = U.random_frames(1, 200, 100, seed0=81, alpha=False)[0]

    def run(c):
        with Context() as ctx:
            ctx.add_input_buffer(0, pack_raw_bgra(src, 200, 100, alpha_meaningful=False))
            ctx.add_output_buffer(1)
            _run(ctx, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"constrain": c}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
            return unpack_raw_bgra(ctx.get_output_buffer(1))
    # fit_crop to a square, gravity right: columns 100..200 of the source, scaled to 50x50
    rows, w, h, alpha = run({"mode": "fit_crop", "w": 50, "h": 50, "gravity": {"percentage": {"x": 100, "y": 50}}, "hints": {"down_filter": "lanczos"}})
    win = np.zeros((100, U.stride_for(100)), np.uint8)
    win[:, :400] = src[:, 400:800]
    assert (w, h, alpha) == (50, 50, False) and np.array_equal(rows, _oracle_resize(win, 100, 100, 50, 50, filter_id=6))
    # fit_pad into 120x120 on an opaque colour: the 120x60 image in the middle of the canvas
    rows, w, h, alpha = run({"mode": "fit_pad", "w": 120, "h": 120, "canvas_color": {"srgb": {"hex": "336699FF"}}})
    img = _oracle_resize(src, 200, 100, 120, 60, filter_id=2)
    can = _canvas_rows(120, 120, (0x99, 0x66, 0x33, 0xFF))
    rc, can_alpha = O.copy_rect(img, 120, 60, img.shape[1], False, can, 120, 120, can.shape[1], False, 0, 0, 0, 30, 120, 60)
    assert rc == 0 and (w, h, alpha) == (120, 120, False) and np.array_equal(rows[:, :480], can[:, :480])
    # within_pad without a colour pads with Transparent: a Bgra32 canvas (clone_crop_fill_expand.rs:236)
    rows, w, h, alpha = run({"mode": "within_pad", "w": 80, "h": 80})
    img = _oracle_resize(src, 200, 100, 80, 40, filter_id=2)
    can = _canvas_rows(80, 80)
    rc, can_alpha = O.copy_rect(img, 80, 40, img.shape[1], False, can, 80, 80, can.shape[1], True, 0, 0, 0, 20, 80, 40)
    assert rc == 0 and (w, h, alpha) == (80, 80, True) and np.array_equal(rows[:, :320], can[:, :320])
    # aspect_crop changes no pixel: the middle 100 columns
    rows, w, h, alpha = run({"mode": "aspect_crop", "w": 10, "h": 10})
    assert (w, h) == (100, 100) and np.array_equal(rows[:, :400], src[:, 200:600])


def test_a_frame_with_several_readers_is_not_changed_through_a_node_that_disappeared():
    """The reference deletes a node that has nothing to do and snaps its neighbours together (delete_node_and_snap_together), so
    the mutating node behind it sees the SHARED parent and MutProtect gives it a Clone (definitions.rs:320-341).  The interpreter
    hands the frame through such a node: whether a frame has other readers is a property of the frame, not of the edge it
    arrived on (round 6; before, the flip below ran in place on the decoded frame and the second output came out flipped).
    Likewise region: its Crop is a MutProtect node, and copy_rectangle normalises its INPUT's unused alpha in place."""
    src = U.random_frames(1, 60, 40, seed0=91, alpha=True)[0]
    job = _graph({0: {"decode": {"io_id": 0}}, 1: {"resample_2d": {"w": 60, "h": 40, "hints": {}}}, 2: "flip_h", 3: {"encode": {"io_id": 1, "preset": "gif"}},
                  4: {"encode": {"io_id": 2, "preset": "gif"}},
                  5: {"region": {"x1": 0, "y1": 0, "x2": 70, "y2": 40, "background_color": {"srgb": {"hex": "10203080"}}}}, 6: {"encode": {"io_id": 3, "preset": "gif"}}},
                 [(0, 1, "input"), (1, 2, "input"), (2, 3, "input"), (0, 4, "input"), (0, 5, "input"), (5, 6, "input")])
    for alpha in (True, False):
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 60, 40, alpha_meaningful=alpha))
            for o in (1, 2, 3):
                c.add_output_buffer(o)
            _run(c, "v1/execute", job)
            flipped = unpack_raw_bgra(c.get_output_buffer(1))[0]
            same = unpack_raw_bgra(c.get_output_buffer(2))[0]
            padded, pw, ph, palpha = unpack_raw_bgra(c.get_output_buffer(3))
        px = src[:, :240].reshape(40, 60, 4)
        assert np.array_equal(same[:, :240], src[:, :240]), alpha                      # every byte, the unused alpha too
        assert np.array_equal(flipped[:, :240].reshape(40, 60, 4)[:, :, :3], px[:, ::-1, :3]), alpha
        assert (pw, ph, palpha) == (70, 40, True) and np.array_equal(padded[:, :240].reshape(40, 60, 4)[:, :, :3], px[:, :, :3])


def test_fill_rect_refuses_an_empty_rectangle_like_the_node_does():
    """FillRectNodeDef::mutate checks `x2 <= x1 || y2 <= y1 || ... x2 > w || y2 > h` -> InvalidCoordinates before it fills
    (clone_crop_fill_expand.rs:114-127); only fill_rectangle itself lets a zero-width rectangle pass (bitmaps.rs:1523)."""
    src = U.random_frames(1, 20, 10, seed0=95, alpha=False)[0]
    for rect, ok in (((3, 2, 3, 8), False), ((3, 2, 9, 2), False), ((3, 2, 21, 8), False), ((3, 2, 9, 8), True), ((0, 0, 20, 10), True)):
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 20, 10, alpha_meaningful=False))
            c.add_output_buffer(1)
            status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"fill_rect": {"x1": rect[0], "y1": rect[1], "x2": rect[2], "y2": rect[3], "color": "black"}},
                                                                            {"encode": {"io_id": 1, "preset": "gif"}}]}})
            assert (status == 200) == ok, (rect, status, r)
            if not ok:
                assert status == 400 and "InvalidCoordinates" in r["message"] and c.error_code() == 2


def test_graph_copy_rect_to_canvas_matches_the_oracle():
    """visuals/composition.rs:111-160 (test_graph_copy_rect_to_canvas): 100x100 of the input copied to (50,50) of a red
    300x300 Bgra32 canvas."""
    src = U.random_frames(1, 220, 140, seed0=41, alpha=False)[0]
    job = _graph({0: {"decode": {"io_id": 0}}, 1: {"create_canvas": {"w": 300, "h": 300, "format": "bgra_32", "color": {"srgb": {"hex": "FF0000FF"}}}},
                  2: {"copy_rect_to_canvas": {"x": 50, "y": 50, "from_x": 0, "from_y": 0, "w": 100, "h": 100}},
                  3: {"encode": {"io_id": 1, "preset": {"lodepng": {"maximum_deflate": None}}}}},
                 [(0, 2, "input"), (1, 2, "canvas"), (2, 3, "input")])
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 220, 140, alpha_meaningful=False))
        c.add_output_buffer(1)
        _run(c, "v1/execute", job)
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
    can = _canvas_rows(300, 300, (0, 0, 255, 255))
    inp = src.copy()
    rc, can_alpha = O.copy_rect(inp, 220, 140, inp.shape[1], False, can, 300, 300, can.shape[1], True, 0, 0, 50, 50, 100, 100)
    assert rc == 0 and (w, h) == (300, 300) and alpha == can_alpha and np.array_equal(rows[:, :1200], can[:, :1200])
    job["framewise"]["graph"]["nodes"]["2"]["copy_rect_to_canvas"]["from_x"] = 200          # 200 + 100 > 220
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 220, 140, alpha_meaningful=False))
        c.add_output_buffer(1)
        status, r = c.send_json("v1/execute", job)
        assert status == 400 and "Invalid coordinates" in r["message"]


@pytest.mark.parametrize("wm,place", [
    ({"io_id": 1}, None),                                                                                   # defaults: whole canvas, within, centre
    ({"io_id": 1, "fit_box": {"image_percentage": {"x1": 60, "y1": 60, "x2": 95, "y2": 95}}, "fit_mode": "fit", "gravity": {"percentage": {"x": 100, "y": 100}}, "opacity": 0.6}, None),
    ({"io_id": 1, "fit_box": {"image_margins": {"left": 10, "top": 20, "right": 150, "bottom": 100}}, "fit_mode": "distort", "opacity": 0.25}, None),
    ({"io_id": 1, "min_canvas_width": 500}, "skipped"),
])
def test_watermark_node_through_the_job_interface(wm, place):
    """flow/nodes/watermark.rs:100-196 sent as JSON: bounding box, constraint, gravity, [alpha(opacity)], DrawImageExact
    (Compose).  Expected pixels: the same placement arithmetic restated here + the oracle chain."""
    import math
    cw, ch, mw, mh = 320, 200, 90, 40
    back = U.random_frames(1, cw, ch, seed0=51, alpha=False)[0]
    mark = U.random_frames(1, mw, mh, seed0=52, alpha=True)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(back, cw, ch, alpha_meaningful=False))
        c.add_input_buffer(1, pack_raw_bgra(mark, mw, mh, alpha_meaningful=True))
        c.add_output_buffer(2)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"watermark": wm}, {"encode": {"io_id": 2, "preset": "gif"}}]}})
        rows, w, h, _ = unpack_raw_bgra(c.get_output_buffer(2))
    if place == "skipped":
        assert np.array_equal(rows[:, :4 * cw], back[:, :4 * cw])
        return
    f32 = np.float32

    def rnd(v):                                                       # f32::round
        return int(math.copysign(math.floor(abs(float(v)) + 0.5), float(v)))
    box = (0, 0, cw, ch)
    fb = wm.get("fit_box")
    if fb and "image_percentage" in fb:
        p = fb["image_percentage"]
        box = tuple(rnd(f32(min(max(p[k], 0), 100)) / f32(100) * f32(s)) for k, s in (("x1", cw), ("y1", ch), ("x2", cw), ("y2", ch)))
    elif fb:
        m = fb["image_margins"]
        box = (m["left"], m["top"], cw - m["right"], ch - m["bottom"])
    bw, bh = box[2] - box[0], box[3] - box[1]
    mode = wm.get("fit_mode", "within")
    if mode == "distort":
        tw, th = bw, bh
    elif mode == "within" and mw <= bw and mh <= bh:
        tw, th = mw, mh
    elif mw / mh > bw / bh:
        tw, th = bw, max(1, rnd(bw / (mw / mh)))
    else:
        tw, th = max(1, rnd(bh * (mw / mh))), bh
    g = wm.get("gravity", {}).get("percentage", {"x": 50, "y": 50}) if isinstance(wm.get("gravity"), dict) else {"x": 50, "y": 50}
    x = rnd(f32(bw - tw) * (f32(min(max(g["x"], 0), 100)) / f32(100))) + box[0]
    y = rnd(f32(bh - th) * (f32(min(max(g["y"], 0), 100)) / f32(100))) + box[1]
    m2 = mark.copy()
    op = wm.get("opacity", 1.0)
    if op < 1.0:
        mat = np.eye(5, dtype=np.float32)
        mat[3, 3] = op
        O.apply_color_matrix(m2, mw, mh, m2.shape[1], mat)
    can = back.copy()
    rc, _ = O.scale_and_render(m2, mw, mh, can, cw, ch, x, y, tw, th, filter_id=4 if (tw > mw or th > mh) else 2,
                               compositing=O.BLEND_WITH_SELF, alpha_meaningful=True)
    assert rc == 0 and np.array_equal(rows[:, :4 * cw], can[:, :4 * cw]), (box, tw, th, x, y)
    assert not np.array_equal(rows[:, :4 * cw], back[:, :4 * cw])


def test_color_and_orientation_nodes_equal_the_oracle():
    src = U.random_frames(1, 50, 30, seed0=61, alpha=True)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 50, 30, alpha_meaningful=True))
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"color_filter_srgb": "sepia"}, {"color_filter_srgb": {"contrast": 0.3}},
                                                        {"apply_orientation": {"flag": 7}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        rows, w, h, _ = unpack_raw_bgra(c.get_output_buffer(1))
    from imageflow_amd.flow.nodes import color as CN
    exp = src.copy()
    O.apply_color_matrix(exp, 50, 30, exp.shape[1], CN.sepia())
    O.apply_color_matrix(exp, 50, 30, exp.shape[1], CN.contrast(0.3))
    O.flip_vertical(exp, 50, 30, exp.shape[1])                           # flag 7 = rotate_180, transpose
    O.flip_horizontal(exp, 50, 30, exp.shape[1])
    t = np.zeros((50, U.stride_for(30)), np.uint8)
    O.transpose(exp, 50, 30, exp.shape[1], t, 30, 50, t.shape[1])
    assert (w, h) == (30, 50) and np.array_equal(rows[:, :120], t[:, :120])


JOB_STEPS = [{"decode": {"io_id": 0}}, "flip_h", "rotate_90", {"resample_2d": {"w": 30, "h": 20, "hints": {"sharpen_percent": None}}},
             {"constrain": {"mode": "within", "w": 5, "h": 5}}, {"encode": {"io_id": 1, "preset": "gif"}}]


def test_job_with_cancellation_at_every_point():
    """imageflow_abi/src/lib.rs:1669-1743: cancel at the n-th poll for n = 1, 2, ... -- every run either ends in category
    21 with the countdown used up, or (first n beyond the job's poll count) completes without error."""
    src = U.random_frames(1, 80, 60, seed0=71, alpha=True)[0]
    n = 1
    while True:
        assert n < 200, "the job never ran out of cancellation points"
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 80, 60))
            c.add_output_buffer(1)
            c.L.ifhip_shim_request_cancellation_after_n_polls(c.p, n)
            status, r = c.send_json("v1/execute", {"framewise": {"steps": JOB_STEPS}})
            left = c.L.ifhip_shim_cancellation_polls_remaining(c.p)
            if left > 0:
                assert status == 200 and not c.has_error() and left < n
                assert len(c.get_output_buffer(1)) > 24
                break
            if left == 0 and status == 200:
                # the countdown ran out on the job's very last poll: `fetch_sub(1) < 1` (context.rs:63-71) cancels at poll
                # n + 1, which this job does not have -- the reference's own loop just moves on here (lib.rs:1728-1740)
                assert not c.has_error()
            else:
                assert left < 0 and status == 499 and c.error_code() == 21, (n, left, status, r)
        n += 1
    assert n > 8                                                                     # at least one poll per node


def test_cancellation_from_another_thread_while_a_job_runs():
    import threading
    src = U.random_frames(1, 2000, 1500, seed0=72, alpha=False)[0]
    steps = [{"decode": {"io_id": 0}}] + [{"resample_2d": {"w": 1000 + (i % 2), "h": 750, "hints": {"resample_when": "always"}}} for i in range(6000)]
    with Context() as c:                                                      # (a step takes ~60 us since plans and device memory are cached)
        c.add_input_buffer(0, pack_raw_bgra(src, 2000, 1500, alpha_meaningful=False))
        t = threading.Timer(0.03, c.request_cancellation)
        t.start()
        status, r = c.send_json("v1/execute", {"framewise": {"steps": steps}})
        t.join()
        assert status == 499 and c.error_code() == 21


def test_performance_block_and_take_output_buffer():
    src = U.random_frames(1, 640, 480, seed0=81, alpha=False)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 640, 480, alpha_meaningful=False))
        c.add_output_buffer(1)
        r = _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"resample_2d": {"w": 64, "h": 48}}, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        perf = r["data"]["job_result"]["performance"]["frames"]
        assert len(perf) == 1
        f = perf[0]
        walls = [n["wall_microseconds"] for n in f["nodes"]]
        assert walls == sorted(walls, reverse=True)                                   # execution_engine.rs:188-189
        assert {n["name"] for n in f["nodes"]} == {"primitive_decoder", "create_canvas", "draw_image_to_canvas", "primitive_encoder"}
        assert f["wall_microseconds"] >= sum(walls) + f["overhead_microseconds"] - len(walls)
        draw = [n for n in f["nodes"] if n["name"] == "draw_image_to_canvas"][0]
        assert 0 < draw["gpu_microseconds"] <= draw["wall_microseconds"] + 1
        taken = c.take_output_buffer(1)
        assert taken is not None and np.array_equal(unpack_raw_bgra(taken)[0], _oracle_resize(src, 640, 480, 64, 48))
        assert c.take_output_buffer(1) is None and c.get_output_buffer(1) is None


def test_matte_applying_encode_does_not_touch_a_frame_other_nodes_still_read():
    """graph mode: libjpeg_turbo's encode flattens in place (mozjpeg.rs:88-94) -- on a private copy when the frame has
    other consumers."""
    src = U.random_frames(1, 48, 32, seed0=91, alpha=True)[0]
    job = _graph({0: {"decode": {"io_id": 0}}, 1: {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": {"quality": 80}}}}, 2: {"encode": {"io_id": 2, "preset": "gif"}}},
                 [(0, 1, "input"), (0, 2, "input")])
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 48, 32, alpha_meaningful=True))
        c.add_output_buffer(1)
        c.add_output_buffer(2)
        _run(c, "v1/execute", job)
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(2))
        assert alpha and np.array_equal(rows[:, :4 * w], src[:, :4 * w])
        assert c.get_output_buffer(1)[:2] == b"\xff\xd8"


def test_tell_decoder_hints_apply_to_a_later_job():
    """v1/tell_decoder before v1/execute (the reference's two-call pattern, json/endpoints/v1.rs:365-371): the decoder runs
    at the told size with the told luma scaler, exactly as if the decode node carried the command itself."""
    data = _jpeg(1280, 720)
    hints = {"width": 600, "height": 330, "scale_luma_spatially": True, "gamma_correct_for_srgb_during_spatial_luma_scaling": True}
    steps_plain = [{"decode": {"io_id": 0}}, {"resample_2d": {"w": 320, "h": 180}}, {"encode": {"io_id": 1, "preset": "gif"}}]
    steps_cmd = [{"decode": {"io_id": 0, "commands": [{"jpeg_downscale_hints": hints}]}}] + steps_plain[1:]
    outs = []
    for told, steps in ((True, steps_plain), (False, steps_cmd), (False, steps_plain)):
        with Context() as c:
            c.add_input_buffer(0, data)
            c.add_output_buffer(1)
            if told:
                status, _ = c.send_json("v1/tell_decoder", {"io_id": 0, "command": {"jpeg_downscale_hints": hints}})
                assert status == 200
            _run(c, "v1/execute", {"framewise": {"steps": steps}})
            outs.append(unpack_raw_bgra(c.get_output_buffer(1))[0])
    assert np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[0], outs[2])
    j = O.jpeg_read_coefficients(data)
    small = O.jpeg_idct_color_scaled(j, 4, 2)                                          # 640x360 covers 600x330
    assert np.array_equal(outs[0], _oracle_resize(small, 640, 360, 320, 180, filter_id=2))


def test_build_with_byte_array_input_and_base64_output():
    """IoEnum::ByteArray in, IoEnum::OutputBase64 out (imageflow_types/src/lib.rs:1433-1456): the job result carries the
    encoded bytes as ResultBytes::Base64."""
    import base64
    src = U.random_frames(1, 24, 16, seed0=91, alpha=False)[0]
    raw = pack_raw_bgra(src, 24, 16, alpha_meaningful=False)
    with Context() as c:
        r = _run(c, "v1/build", {"io": [{"io_id": 0, "direction": "in", "io": {"byte_array": list(raw)}},
                                        {"io_id": 1, "direction": "out", "io": "output_base_64"}],
                                 "framewise": {"steps": [{"decode": {"io_id": 0}}, {"resample_2d": {"w": 12, "h": 8}},
                                                         {"encode": {"io_id": 1, "preset": "gif"}}]}})
        enc = r["data"]["build_result"]["encodes"][0]
        got = base64.b64decode(enc["bytes"]["base_64"])
        assert got == bytes(c.get_output_buffer(1)) and (enc["w"], enc["h"]) == (12, 8)
        rows = unpack_raw_bgra(got)[0]
    assert np.array_equal(rows, _oracle_resize(src, 24, 16, 12, 8, filter_id=2))


def _bgra(hex_rgba):
    r, g, b, a = (int(hex_rgba[i:i + 2], 16) for i in (0, 2, 4, 6))
    return np.array([b, g, r, a], np.uint8)


@pytest.mark.parametrize("node,corners,color,color_bgra", [
    ({"region": {"x1": -5, "y1": 4, "x2": 40, "y2": 36}}, (-5, 4, 40, 36), {"srgb": {"hex": "2233AAFF"}}, "2233AAFF"),       # crop + expand left / bottom
    ({"region": {"x1": 3, "y1": 2, "x2": 20, "y2": 11}}, (3, 2, 20, 11), "transparent", "00000000"),                        # inside: crop, expand by nothing
    ({"region": {"x1": 60, "y1": 0, "x2": 70, "y2": 10}}, (60, 0, 70, 10), {"srgb": {"hex": "FF8000"}}, "FF8000FF"),         # misses the frame: a canvas
    ({"region_percent": {"x1": 10, "y1": 20, "x2": 110, "y2": 60}}, (5, 6, 55, 18), "black", "000000FF"),                    # right edge beyond the frame
    ({"region_percent": {"x1": 33, "y1": 0, "x2": 33.5, "y2": 100}}, (17, 0, 17 + 0, 30), "black", "000000FF"),              # 16.5 rounds away from zero; see below
])
def test_region_nodes_crop_then_expand_like_the_reference(node, corners, color, color_bgra):
    """clone_crop_fill_expand.rs:263-452: RegionPercent -> Region (f32 percent arithmetic, half away from zero) -> Crop +
    ExpandCanvas, or a plain canvas of the colour when the rectangle misses the frame.  Expected pixels by numpy."""
    src = U.random_frames(1, 50, 30, seed0=83, alpha=True)[0]
    name = next(iter(node))
    node = {name: dict(node[name], background_color=color)}
    x1, y1, x2, y2 = corners
    if name == "region_percent" and x2 <= x1:
        # get_coords :279-284 only widens `x2 < x1`; 16.75 rounds to 17 as well, so Region refuses the empty rectangle
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 50, 30, alpha_meaningful=True))
            c.add_output_buffer(1)
            status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, node, {"encode": {"io_id": 1, "preset": "gif"}}]}})
            assert status == 400 and "Not a rectangle" in r["message"], r
        return
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 50, 30, alpha_meaningful=True))
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, node, {"encode": {"io_id": 1, "preset": "gif"}}]}})
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
    assert (w, h) == (x2 - x1, y2 - y1)
    exp = np.tile(_bgra(color_bgra), (h, w, 1))
    sx1, sy1, sx2, sy2 = max(0, x1), max(0, y1), min(50, x2), min(30, y2)
    if sx2 > sx1 and sy2 > sy1:
        exp[sy1 - y1:sy2 - y1, sx1 - x1:sx2 - x1] = src[:, :200].reshape(30, 50, 4)[sy1:sy2, sx1:sx2]
    assert np.array_equal(rows[:, :4 * w].reshape(h, w, 4), exp)
    assert alpha                                                    # Bgra32 parent: the format survives an opaque colour (:237)


def test_region_refuses_an_empty_rectangle():
    src = U.random_frames(1, 16, 16, seed0=84, alpha=False)[0]
    for node in ({"region": {"x1": 4, "y1": 4, "x2": 4, "y2": 9, "background_color": "black"}},
                 {"region_percent": {"x1": 50, "y1": 10, "x2": 50, "y2": 90, "background_color": "black"}}):
        with Context() as c:
            c.add_input_buffer(0, pack_raw_bgra(src, 16, 16, alpha_meaningful=False))
            c.add_output_buffer(1)
            status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, node, {"encode": {"io_id": 1, "preset": "gif"}}]}})
            assert status == 400 and c.error_code() != 0 and "Not a rectangle" in r["message"], r


# ---- EXIF orientation behind decode (flow/nodes/codecs_and_pointer.rs:94-107) ---------------------------------------------
def _exif_segment(value, little=True):
    import struct
    e = "<" if little else ">"
    tiff = (b"II\x2a\x00" if little else b"MM\x00\x2a") + struct.pack(e + "I", 8) + struct.pack(e + "H", 1)
    tiff += struct.pack(e + "HHI", 0x0112, 3, 1) + struct.pack(e + "H", value) + b"\0\0" + struct.pack(e + "I", 0)
    data = b"Exif\0\0" + tiff + b"\0" * 8
    return b"\xff\xe1" + struct.pack(">H", len(data) + 2) + data


def _oracle_orient(rows, w, h, flag):
    """ApplyOrientationDef::expand (rotate_flip_transpose.rs:44-66) with the primitives of :100-215 on the oracle's bitmaps:
    rotate_90 = transpose + flip_h, rotate_180 = flip_v + flip_h, rotate_270 = transpose + flip_v."""
    rows = np.ascontiguousarray(rows).copy()

    def tr(a, aw, ah):
        t = np.zeros((aw, U.stride_for(ah)), np.uint8)
        O.transpose(a, aw, ah, a.shape[1], t, ah, aw, t.shape[1])
        return t, ah, aw
    if flag == 2: O.flip_horizontal(rows, w, h, rows.shape[1])
    elif flag == 3: O.flip_vertical(rows, w, h, rows.shape[1]); O.flip_horizontal(rows, w, h, rows.shape[1])
    elif flag == 4: O.flip_vertical(rows, w, h, rows.shape[1])
    elif flag == 5: rows, w, h = tr(rows, w, h)
    elif flag == 6: rows, w, h = tr(rows, w, h); O.flip_horizontal(rows, w, h, rows.shape[1])
    elif flag == 7:
        O.flip_vertical(rows, w, h, rows.shape[1]); O.flip_horizontal(rows, w, h, rows.shape[1])
        rows, w, h = tr(rows, w, h)
    elif flag == 8: rows, w, h = tr(rows, w, h); O.flip_vertical(rows, w, h, rows.shape[1])
    return rows, w, h


@pytest.mark.parametrize("subsampling", ["4:2:0", "4:4:4"])
@pytest.mark.parametrize("flag", list(range(0, 9)) + [9])
def test_exif_orientation_is_applied_behind_every_decode(flag, subsampling):
    """Eight 4:2:0 files that differ only in their EXIF orientation tag (+ tag 0 and an out-of-range 9): decode ->
    resample_2d -> encode equals oracle decode + oracle orientation + oracle resize; the job's decode record and
    v1/get_image_info report the rotated size (context.rs:486-538); flags 0, 1 and 9 keep decode + resample one call
    where the decode can be fused at all (4:4:4 at full size; full-size 4:2:0 needs the fancy up-sampler's bitmap)."""
    base = _jpeg(176, 112, seed=flag, subsampling=subsampling)
    data = base[:2] + _exif_segment(flag, little=flag % 2 == 0) + base[2:]
    swap = 5 <= flag <= 8
    ow, oh = (40, 66) if swap else (66, 40)
    with Context() as c:
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        status, info = c.send_json("v1/get_image_info", {"io_id": 0})
        assert status == 200
        ii = info["data"]["image_info"]
        assert (ii["image_width"], ii["image_height"]) == ((112, 176) if swap else (176, 112))
        r = _run(c, "v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}},
                                                           {"resample_2d": {"w": ow, "h": oh, "hints": {"down_filter": "robidoux"}}},
                                                           {"encode": {"io_id": 1, "preset": "gif"}}]}})
        d = r["data"]["job_result"]["decodes"][0]
        assert (d["w"], d["h"]) == ((112, 176) if swap else (176, 112))
        rows, w, h, alpha = unpack_raw_bgra(c.get_output_buffer(1))
        assert c.L.ifhip_shim_fused_decode_resamples(c.p) == (1 if flag in (0, 1, 9) and subsampling == "4:4:4" else 0)
        names = [n["name"] for f in r["data"]["job_result"]["performance"]["frames"] for n in f["nodes"]]
        assert ("apply_orientation" in names) == (2 <= flag <= 8)
    j = O.jpeg_read_coefficients(data)                                     # (the oracle's parser skips APP1 like any APPn)
    full = O.jpeg_idct_color(j)
    rot, rw, rh = _oracle_orient(full, 176, 112, flag)
    exp = _oracle_resize(rot, rw, rh, ow, oh, filter_id=2)
    assert (w, h, alpha) == (ow, oh, False) and np.array_equal(rows, exp)


def test_exif_orientation_in_a_command_string_job():
    """command_string sizes its layout on the ROTATED frame and hands hints worked out from the rotated sides to the
    decoder as they are (command_string.rs:20-58, ir4/mod.rs:167-176): a 1600x600 file tagged 6 is 600x1600 to the
    querystring; width=100 -> 100x267; pre-shrink min(600/100, 1600/100) = 6 -> hints 210x560 against the decoder's
    UNROTATED 1600x600 -> 8/8 (7/8 is skipped, 6/8 = 450 rows < 560), so the job decodes at full size."""
    base = _jpeg(1600, 600, seed=5)
    data = base[:2] + _exif_segment(6) + base[2:]
    with Context() as c:
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=100", "decode": 0, "encode": 1}}]}})
        rows, w, h, _ = unpack_raw_bgra(c.get_output_buffer(1))
    assert (w, h) == (100, 267)
    j = O.jpeg_read_coefficients(data)
    rot, rw, rh = _oracle_orient(O.jpeg_idct_color(j), 1600, 600, 6)
    assert (rw, rh) == (600, 1600)
    assert np.array_equal(rows, _oracle_resize(rot, 600, 1600, 100, 267, filter_id=2))


# ---- decodes of different threads' jobs coalesced into one device call -------------------------------------------------
def test_concurrent_jobs_share_one_entropy_decode_and_keep_their_own_bytes(debug_switch):
    """Six threads, one context each (lib.rs:20-27), each a different 640x400 file through `command_string width=100`, started
    together; the coalescer is told to wait for all six (a development switch: the product waits ~0.1 ms).  Every job's
    output equals ITS file's oracle chain, and at least one decode shared its device call with another thread's."""
    import threading
    n = 6
    debug_switch("coalesce_window_us", "2000000")
    debug_switch("coalesce_wait_for", str(n))
    files = [_jpeg(640, 400, seed=300 + i, quality=60 + 5 * i) for i in range(n)]          # same geometry, own tables and content
    outs, shared, errs = [None] * n, [0] * n, []
    gate = threading.Barrier(n)

    def run(i):
        try:
            with Context() as c:
                c.add_input_buffer(0, files[i])
                c.add_output_buffer(1)
                gate.wait()
                _run(c, "v1/execute", {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=100", "decode": 0, "encode": 1}}]}})
                outs[i] = unpack_raw_bgra(c.get_output_buffer(1))
                shared[i] = c.L.ifhip_shim_coalesced_decodes(c.p)
        except Exception as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert sum(shared) >= 2, shared                                       # a batch has at least two members
    for i in range(n):
        rows, w, h, _ = outs[i]
        j = O.jpeg_read_coefficients(files[i])
        # width=100 on 640x400: target 100x63 (62.5 rounds up); pre-shrink min(6.4, 4.0) = 4 -> 2.1 / 4 -> hints 336x210 -> 5/8 = 400x250
        small = O.jpeg_idct_color_scaled(j, 5, 2, general=True)
        assert (w, h) == (100, 63)
        assert np.array_equal(rows[:, :400], _oracle_resize(small, 400, 250, 100, 63, filter_id=2)[:, :400]), i


def test_a_damaged_file_in_a_coalesced_batch_fails_alone(debug_switch):
    """One of three concurrent jobs carries a truncated scan: its job reports the malformed image, the other two finish with
    their own (correct) outputs."""
    import threading
    n = 3
    debug_switch("coalesce_window_us", "2000000")
    debug_switch("coalesce_wait_for", str(n))
    files = [_jpeg(320, 240, seed=400 + i) for i in range(n)]
    files[1] = files[1][: len(files[1]) // 2]                              # scan ends early
    res = [None] * n
    gate = threading.Barrier(n)

    def run(i):
        with Context() as c:
            c.add_input_buffer(0, files[i])
            c.add_output_buffer(1)
            gate.wait()
            status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"resample_2d": {"w": 80, "h": 60}},
                                                                             {"encode": {"io_id": 1, "preset": "gif"}}]}})
            res[i] = (status, r["message"] if status != 200 else unpack_raw_bgra(c.get_output_buffer(1)))
    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert res[1][0] != 200 and "ImageMalformed" in res[1][1], res[1]
    for i in (0, 2):
        assert res[i][0] == 200
        rows, w, h, _ = res[i][1]
        full = O.jpeg_idct_color(O.jpeg_read_coefficients(files[i]))
        assert np.array_equal(rows[:, :320], _oracle_resize(full, 320, 240, 80, 60, filter_id=2)[:, :320]), i


# ---- contexts bound to devices (include/imageflow_abi_subset.h: ifhip_shim_spread_contexts / _context_set_device) --------
def test_contexts_bound_to_devices():
    """With the spreading policy on, new contexts take the usable devices round-robin (one GPU here: all ordinal 0) and their
    jobs run there whichever thread calls; an ordinal past the usable devices is refused on the context; the outputs are the
    unbound context's bytes.  One context per thread, jobs started together (lib.rs:20-27)."""
    import threading
    from imageflow_amd import _native, abi
    L = abi._bind()
    n_dev = _native.lib().ifhip_device_count()
    assert n_dev >= 1
    files = [_jpeg(640, 400, seed=500 + i) for i in range(4)]
    job = {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=100", "decode": 0, "encode": 1}}]}}

    def run_one(c, data):
        c.add_input_buffer(0, data)
        c.add_output_buffer(1)
        _run(c, "v1/execute", job)
        return c.get_output_buffer(1)
    with Context() as c:
        assert c.device == -1
        want = [None] * 4
        want[0] = run_one(c, files[0])
    for i in range(1, 4):
        with Context() as c:
            want[i] = run_one(c, files[i])
    L.ifhip_shim_spread_contexts(1)
    try:
        ctxs = [Context() for _ in range(4)]
        devs = [c.device for c in ctxs]
        assert all(0 <= d < n_dev for d in devs), devs
        assert len(set(devs)) == min(4, n_dev), devs                  # round-robin over the usable devices
        got, errs = [None] * 4, []

        def work(i):
            try:
                got[i] = run_one(ctxs[i], files[i])
            except Exception as e:  # noqa: BLE001
                errs.append((i, repr(e)))
        th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert got == want
        assert not ctxs[0].set_device(n_dev) and ctxs[0].has_error()
        for c in ctxs:
            c.close()
    finally:
        L.ifhip_shim_spread_contexts(0)
    with Context() as c:                                              # explicit binding, then back to "the caller's device"
        assert c.set_device(0) and c.device == 0
        assert run_one(c, files[1]) == want[1]
        assert c.set_device(-1) and c.device == -1


# ---- where the drop-in is not equal it says so: embedded ICC profiles, querystring keys -----------------------------------------
def _tagged(jpeg, profile):
    from tests.test_jpeg_headers import icc_app2
    return jpeg[:2] + icc_app2(profile) + jpeg[2:]


def test_a_file_with_a_non_srgb_profile_is_refused_unless_the_decoder_is_told_to_discard_it():
    """MzDec::read_frame converts a frame to sRGB whenever the file carries an ICC profile (mozjpeg_decoder.rs:370-420) unless
    told discard_color_profile (:88-91).  No CMS here: a Display-P3-tagged file is refused with ActionNotSupported; told to
    discard the profile (decode command or v1/tell_decoder) it decodes to the untagged file's bytes; a profile that IS sRGB
    passes (the reference's transform is the identity up to its rounding)."""
    from tests.test_jpeg_headers import P3_XYZ, make_icc
    base = _jpeg(320, 200, seed=11)
    p3, srgb = _tagged(base, make_icc(xyz=P3_XYZ)), _tagged(base, make_icc())
    steps = [{"decode": {"io_id": 0}}, {"resample_2d": {"w": 80, "h": 50}}, {"encode": {"io_id": 1, "preset": "gif"}}]

    def run(data, steps, tell=False, expect=200):
        with Context() as c:
            c.add_input_buffer(0, data)
            c.add_output_buffer(1)
            if tell:
                assert c.send_json("v1/tell_decoder", {"io_id": 0, "command": "discard_color_profile"})[0] == 200
            status, r = c.send_json("v1/execute", {"framewise": {"steps": steps}})
            assert status == expect, (status, r)
            if expect != 200:
                assert c.error_code() == 8 and "ICC profile" in r["message"] and "discard_color_profile" in r["message"], r
                return None
            return unpack_raw_bgra(c.get_output_buffer(1))[0]
    plain = run(base, steps)
    assert run(p3, steps, expect=400) is None
    assert np.array_equal(run(p3, steps, tell=True), plain)
    with_cmd = [{"decode": {"io_id": 0, "commands": ["discard_color_profile"]}}] + steps[1:]
    assert np.array_equal(run(p3, with_cmd), plain)
    assert np.array_equal(run(srgb, steps), plain)
    # what the reference itself treats as "no profile": a chunk set its reassembly drops (mozjpeg_decoder_helpers.rs:42-83) and
    # a GRAY profile on a colour frame (mozjpeg_decoder.rs:391-395 -> SourceProfile::Srgb) -- decoded, not refused
    from tests.test_jpeg_headers import icc_app2
    broken = base[:2] + icc_app2(make_icc(xyz=P3_XYZ), pieces=3, drop=1) + base[2:]
    assert np.array_equal(run(broken, steps), plain)
    assert np.array_equal(run(_tagged(base, make_icc(space=b"GRAY")), steps), plain)
    # the querystring path decodes through the same gate
    qs = [{"command_string": {"kind": "ir4", "value": "width=80", "decode": 0, "encode": 1}}]
    assert run(p3, qs, expect=400) is None
    assert run(p3, qs, tell=True) is not None


def test_querystring_keys_are_honoured_or_refused_never_dropped():
    """`down.filter` reaches the resample (any spelling FilterStrings knows, ir4/parsing.rs:159-193), `quality` / `jpeg.quality`
    and `format=jpg` reach the JPEG writer (ir4/encoder.rs:74), any other format and unknown filter names are refused."""
    PIL = pytest.importorskip("PIL.Image")
    data = _jpeg(640, 400, seed=21)
    j = O.jpeg_read_coefficients(data)

    def run(value, expect=200):
        with Context() as c:
            c.add_input_buffer(0, data)
            c.add_output_buffer(1)
            status, r = c.send_json("v1/execute", {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": value, "decode": 0, "encode": 1}}]}})
            assert status == expect, (status, r)
            return bytes(c.get_output_buffer(1)) if expect == 200 else r
    # width=160: pre-shrink 2.1 / 4 -> hints 336x210 -> the 5/8 decode (400x250) is the first that covers them ... the oracle
    # chain needs no decode guess: compare the two filters' outputs with each other and the default with robidoux
    default = unpack_raw_bgra(run("width=160"))[0]
    assert np.array_equal(unpack_raw_bgra(run("width=160&down.filter=robidoux"))[0], default)
    lanczos = unpack_raw_bgra(run("width=160&down.filter=lanczos"))[0]
    assert not np.array_equal(lanczos, default)
    assert np.array_equal(unpack_raw_bgra(run("width=160&down.filter=Lanczos"))[0], lanczos)
    assert np.array_equal(unpack_raw_bgra(run("width=160&down.filter=catmullrom"))[0], unpack_raw_bgra(run("width=160&down.filter=catmull_rom"))[0])
    # against the oracle with filter 6 where the decode is known: half size is below the 2.1 pre-shrink threshold -> full decode
    rows, w, h, _ = unpack_raw_bgra(run("width=320&down.filter=lanczos"))
    assert (w, h) == (320, 200) and np.array_equal(rows, _oracle_resize(O.jpeg_idct_color(j), 640, 400, 320, 200, filter_id=6))
    assert "InvalidNodeParams" in run("width=160&down.filter=sharpest", expect=400)["message"]
    assert "ActionNotSupported" in run("width=160&format=png", expect=400)["message"]
    assert "ActionNotSupported" in run("width=160&format=webp", expect=400)["message"]
    # quality: a real JPEG whose tables are libjpeg's for that quality; jpeg.quality wins over quality; format=jpg alone -> 90
    def tables(buf):
        im = PIL.open(io.BytesIO(buf))
        assert im.format == "JPEG" and im.size == (160, 100)
        return im.quantization
    def libjpeg_tables(q):
        b = io.BytesIO()
        PIL.new("RGB", (16, 16)).save(b, "JPEG", quality=q, subsampling="4:2:0")
        return PIL.open(io.BytesIO(b.getvalue())).quantization
    assert tables(run("width=160&quality=50")) == libjpeg_tables(50)
    assert tables(run("width=160&format=jpg")) == libjpeg_tables(90)
    assert tables(run("width=160&format=jpeg&quality=30&jpeg.quality=77")) == libjpeg_tables(77)
    assert tables(run("width=160&quality=high")) == libjpeg_tables(90)       # not an integer: ignored as the reference's parse_i32 does -> the default
    # an explicit format=jpg makes the querystring layer resample onto WHITE (ir4/layout.rs:492-503, :530): a Bgra32 source is
    # flattened in the working space by the resampler, not by the encoder afterwards
    src = U.random_frames(1, 120, 80, seed0=33, alpha=True)[0]
    with Context() as c:
        c.add_input_buffer(0, pack_raw_bgra(src, 120, 80, alpha_meaningful=True))
        c.add_output_buffer(1)
        _run(c, "v1/execute", {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=60&format=jpg&quality=85", "decode": 0, "encode": 1}}]}})
        got = bytes(c.get_output_buffer(1))
    can = _canvas_rows(60, 40, (255, 255, 255, 255))
    rc, _ = O.scale_and_render(src, 120, 80, can, 60, 40, 0, 0, 60, 40, filter_id=2, compositing=O.BLEND_WITH_MATTE, matte_bgra=0xFFFFFFFF, alpha_meaningful=True)
    b = io.BytesIO()
    PIL.fromarray(np.ascontiguousarray(can[:, :240].reshape(40, 60, 4)[:, :, 2::-1])).save(b, "JPEG", quality=85, subsampling="4:2:0")
    assert rc == 0 and got == b.getvalue()
