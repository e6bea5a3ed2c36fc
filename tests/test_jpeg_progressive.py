"""Host-side entropy decoder for progressive / multi-scan JPEGs (csrc/jpeg_read.cpp) without a GPU.

Pinned two ways: (1) this library's own file WRITER -- byte-identical to libjpeg-turbo for progressive files
(tests/test_jpeg_writer.py) -- codes known coefficient planes as a progressive file, and the reader must return exactly
those planes; (2) a progressive file Pillow (libjpeg-turbo) wrote holds the same coefficients as the baseline file Pillow
writes from the same pixels at the same quality, which the oracle's serial baseline decoder reads."""
import ctypes as C
import io

import numpy as np
import pytest

from imageflow_amd import _native
from oracle import oracle as O


def _lib():
    L = _native.lib()
    L.ifhip_jpeg_frame_info.argtypes = [C.c_char_p, C.c_size_t] + [C.c_void_p] * 9
    L.ifhip_jpeg_read_coefficients_host.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def read_host(data):
    L = _lib()
    w, h, n, prog = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()
    hs, vs = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
    bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
    qt = np.zeros((3, 64), np.uint16)
    _native.check(L.ifhip_jpeg_frame_info(data, len(data), C.addressof(w), C.addressof(h), C.addressof(n), hs.ctypes.data, vs.ctypes.data,
                                          bw.ctypes.data, bh.ctypes.data, qt.ctypes.data, C.addressof(prog)))
    coef = [np.full((max(int(bh[c]), 1), max(int(bw[c]), 1), 64), 77, np.int16) for c in range(3)]      # (the call clears them)
    _native.check(L.ifhip_jpeg_read_coefficients_host(data, len(data), coef[0].ctypes.data, coef[1].ctypes.data, coef[2].ctypes.data,
                                                      qt.ctypes.data))
    return {"width": w.value, "height": h.value, "ncomp": n.value, "progressive": bool(prog.value), "hs": [int(v) for v in hs], "vs": [int(v) for v in vs],
            "bw": [int(v) for v in bw], "bh": [int(v) for v in bh], "qt": qt, "coef": coef}


def _image(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    return np.clip(np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1) +
                   (35 * np.sin(x / 2.5) * np.cos(y / 4.0))[..., None] + rng.integers(-25, 26, (h, w, 3)), 0, 255).astype(np.uint8)


def _save(img, **kw):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


@pytest.mark.parametrize("subsampling", ["4:4:4", "4:2:2", "4:2:0"])
@pytest.mark.parametrize("size", [(1, 1), (8, 8), (17, 9), (100, 75), (321, 203)])
@pytest.mark.parametrize("quality", [30, 90])
def test_progressive_files_hold_the_coefficients_of_their_baseline_twins(size, subsampling, quality):
    pytest.importorskip("PIL.Image")
    img = _image(size[0], size[1], size[0] * 3 + quality)
    prog = _save(img, quality=quality, subsampling=subsampling, progressive=True)
    base = _save(img, quality=quality, subsampling=subsampling, progressive=False, optimize=False)
    assert b"\xff\xc2" in prog
    got = read_host(prog)
    j = O.jpeg_read_coefficients(base)
    assert got["progressive"] and (got["width"], got["height"], got["ncomp"]) == (j["width"], j["height"], j["ncomp"])
    assert got["hs"][:3] == [int(v) for v in j["hs"][:3]] and got["bw"][:3] == [int(v) for v in j["bw"][:3]]
    assert np.array_equal(got["qt"], j["qt"][:3])
    for c in range(3):
        # the visible blocks carry the image; the padding blocks of a non-interleaved AC scan are not coded (jdinput.c
        # per_scan_setup: width_in_blocks), a baseline file codes them as dummy blocks: compare what both define
        wb = -(-size[0] * got["hs"][c] // (8 * max(got["hs"])))
        hb = -(-size[1] * got["vs"][c] // (8 * max(got["vs"])))
        assert np.array_equal(got["coef"][c][:hb, :wb], j["coef"][c][:hb, :wb]), c
        assert np.array_equal(got["coef"][c][:, :, 0], j["coef"][c][:, :, 0]), c       # DC scans are interleaved: every block


def test_grayscale_and_restart_intervals():
    from PIL import Image
    img = _image(150, 90, 5)[..., 0]
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=75, progressive=True)
    got = read_host(b.getvalue())
    b2 = io.BytesIO()
    Image.fromarray(img).save(b2, "JPEG", quality=75, progressive=False, optimize=False)
    j = O.jpeg_read_coefficients(b2.getvalue())
    assert got["ncomp"] == 1 and np.array_equal(got["coef"][0][:12, :19], j["coef"][0][:12, :19])


@pytest.mark.parametrize("flags", [2, 3])
def test_the_writers_progressive_files_read_back_exactly(flags):
    """Known coefficient planes (from the oracle's decode of a baseline file, dummy blocks included) -> ifhip_jpeg_write with
    the progressive flag (byte-identical to libjpeg-turbo's jcphuff.c, tests/test_jpeg_writer.py) -> the reader."""
    pytest.importorskip("PIL.Image")
    L = _lib()
    for (w, h), sub in (((97, 61), "4:2:0"), ((64, 48), "4:4:4"), ((200, 33), "4:2:2")):
        base = _save(_image(w, h, w), quality=85, subsampling=sub, optimize=False)
        j = O.jpeg_read_coefficients(base)
        bw, bh = np.array(j["bw"][:3], np.uint32), np.array(j["bh"][:3], np.uint32)
        hs, vs = np.array(j["hs"][:3], np.uint8), np.array(j["vs"][:3], np.uint8)
        out = np.zeros(1 << 20, np.uint8)
        n = C.c_size_t()
        L.ifhip_jpeg_write.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                                                           C.c_size_t, C.POINTER(C.c_size_t)]
        coef = [np.ascontiguousarray(j["coef"][c]) for c in range(3)]
        _native.check(L.ifhip_jpeg_write(coef[0].ctypes.data, coef[1].ctypes.data, coef[2].ctypes.data, bw.ctypes.data, bh.ctypes.data, 3,
                                         hs.ctypes.data, vs.ctypes.data, w, h, 85, flags, out.ctypes.data, out.size, C.byref(n)))
        got = read_host(out[:n.value].tobytes())
        for c in range(3):
            wb, hb = -(-w * int(hs[c]) // (8 * int(hs.max()))), -(-h * int(vs[c]) // (8 * int(vs.max())))
            assert np.array_equal(got["coef"][c][:hb, :wb], coef[c][:hb, :wb]), (sub, c)
            assert np.array_equal(got["coef"][c][:, :, 0], coef[c][:, :, 0])


def test_damaged_progressive_files_are_errors_not_crashes():
    pytest.importorskip("PIL.Image")
    data = _save(_image(120, 80, 9), quality=70, progressive=True)
    rng = np.random.default_rng(3)
    for trial in range(300):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            d[int(rng.integers(2, len(d)))] = int(rng.integers(0, 256))
        if trial % 5 == 0:
            d = d[: int(rng.integers(4, len(d)))]
        L = _lib()
        coef = [np.zeros((64, 64, 64), np.int16) for _ in range(3)]                    # far larger than 120x80 needs
        w, h, n, prog = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()
        hs, vs = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
        bw, bh = np.zeros(3, np.uint32), np.zeros(3, np.uint32)
        if L.ifhip_jpeg_frame_info(bytes(d), len(d), C.addressof(w), C.addressof(h), C.addressof(n), hs.ctypes.data, vs.ctypes.data,
                                   bw.ctypes.data, bh.ctypes.data, None, C.addressof(prog)) != 0:
            continue
        if int(bw.max()) > 64 or int(bh.max()) > 64:
            continue                                                                    # (a mutated size: planes would be too small)
        L.ifhip_jpeg_read_coefficients_host(bytes(d), len(d), coef[0].ctypes.data, coef[1].ctypes.data, coef[2].ctypes.data, None)
