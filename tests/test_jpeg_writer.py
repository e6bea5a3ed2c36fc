"""Host half of the classic JPEG encoder (csrc/jpeg_write.cpp, host-only code of libimageflow_hip.so -- no GPU needed):
a baseline file written by libjpeg-turbo (through Pillow: standard tables, no optimisation) is entropy-decoded by the
oracle and written again by ifhip_jpeg_write_baseline; the bytes must be identical -- markers, tables, Huffman codes,
stuffing and padding.  Also jpeg_set_quality's tables against the ones libjpeg-turbo put in the file."""
import ctypes as C
import io

import numpy as np
import pytest
from PIL import Image

from imageflow_amd import _native
from oracle import oracle as O

ZIGZAG_SAMPLINGS = {"4:2:0": ([2, 1, 1], [2, 1, 1]), "4:2:2": ([2, 1, 1], [1, 1, 1]), "4:4:4": ([1, 1, 1], [1, 1, 1])}


def _photo(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 3 % 256)], -1).astype(np.int32)
    return np.clip(base + rng.integers(-40, 40, (h, w, 3)), 0, 255).astype(np.uint8)


def _write(j, quality):
    L = _native.lib()
    L.ifhip_jpeg_write_baseline.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int,
                                            C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    bw, bh = np.array(j["bw"], np.uint32), np.array(j["bh"], np.uint32)
    hs, vs = np.array(j["hs"], np.uint8), np.array(j["vs"], np.uint8)
    n = C.c_size_t(0)
    args = [j["coef"][0].ctypes.data, j["coef"][1].ctypes.data, j["coef"][2].ctypes.data, bw.ctypes.data, bh.ctypes.data, j["ncomp"],
            hs.ctypes.data, vs.ctypes.data, j["width"], j["height"], quality]
    assert L.ifhip_jpeg_write_baseline(*args, None, 0, C.byref(n)) == 0
    out = np.zeros(n.value, np.uint8)
    assert L.ifhip_jpeg_write_baseline(*args, out.ctypes.data, n.value, C.byref(n)) == 0
    short = np.zeros(16, np.uint8)
    assert L.ifhip_jpeg_write_baseline(*args, short.ctypes.data, 16, C.byref(n)) != 0
    return out.tobytes()


@pytest.mark.parametrize("sampling", ["4:2:0", "4:2:2", "4:4:4"])
@pytest.mark.parametrize("size", [(1, 1), (17, 9), (64, 48), (203, 131)])
@pytest.mark.parametrize("quality", [5, 75, 90, 100])
def test_rewritten_file_is_byte_identical(sampling, size, quality):
    w, h = size
    buf = io.BytesIO()
    Image.fromarray(_photo(w, h, w * 31 + h + quality)).save(buf, "JPEG", quality=quality, subsampling=sampling, optimize=False)
    data = buf.getvalue()
    j = O.jpeg_read_coefficients(data)
    assert j["ncomp"] == 3 and (j["hs"], j["vs"]) == ZIGZAG_SAMPLINGS[sampling]
    assert _write(j, quality) == data


def test_grayscale_file_is_byte_identical():
    buf = io.BytesIO()
    Image.fromarray(_photo(50, 37, 3)[:, :, 0]).save(buf, "JPEG", quality=80, optimize=False)
    data = buf.getvalue()
    j = O.jpeg_read_coefficients(data)
    assert j["ncomp"] == 1
    assert _write(j, 80) == data


@pytest.mark.parametrize("quality", [1, 10, 49, 50, 51, 75, 95, 100])
def test_quality_tables_match_libjpeg(quality):
    L = _native.lib()
    L.ifhip_jpeg_quality_tables.argtypes = [C.c_int, C.c_void_p]
    qt = np.zeros((2, 64), np.uint16)
    assert L.ifhip_jpeg_quality_tables(quality, qt.ctypes.data) == 0
    buf = io.BytesIO()
    Image.fromarray(_photo(16, 16, 0)).save(buf, "JPEG", quality=quality, subsampling="4:2:0", optimize=False)
    j = O.jpeg_read_coefficients(buf.getvalue())
    assert np.array_equal(j["qt"][0], qt[0]) and np.array_equal(j["qt"][1], qt[1]) and np.array_equal(j["qt"][2], qt[1])
