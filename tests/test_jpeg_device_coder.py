"""The device entropy coder's algorithm without a device (csrc/jpeg_encode_core.hpp: the block routine, scan order and the
bit / byte placement rules the gfx950 kernels of csrc/jpeg_encode.hip are built from).  tests/enc_emulate.cpp runs the
passes lane by lane on the CPU -- count, scan, write into the shared word stream in a scrambled order, 0xFF count, scan,
stuffed bytes -- and the file must equal what libjpeg-turbo wrote (through Pillow; the coefficients come back through the
oracle's entropy decoder).  The GPU tests (tests/test_gpu_jpeg_device_coder.py) compare the kernels with the same files."""
import ctypes as C
import io
import os
import subprocess
import tempfile

import numpy as np
import pytest
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24

from imageflow_amd import _native
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLINGS = {"4:2:0": ([2, 1, 1], [2, 1, 1]), "4:2:2": ([2, 1, 1], [1, 1, 1]), "4:4:4": ([1, 1, 1], [1, 1, 1])}
_EMU = {}


def emulator():
    if "lib" not in _EMU:
        d = tempfile.mkdtemp(prefix="enc_emulate_")
        so = os.path.join(d, "libenc_emulate.so")
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", os.path.join(HERE, "enc_emulate.cpp"), "-o", so], check=True)
        lib = C.CDLL(so)
        lib.enc_emulate.argtypes = [C.c_void_p] * 3 + [C.c_uint32, C.c_uint32, C.c_int] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _EMU["lib"] = lib
    return _EMU["lib"]


def tables_and_header(ncomp, hs, vs, width, height, quality):
    L = _native.lib()
    L.ifhip_jpeg_debug_encode_tables.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                                 C.c_size_t, C.POINTER(C.c_size_t)]
    tabs, header, n = np.zeros((4, 256), np.uint32), np.zeros(1024, np.uint8), C.c_size_t(0)
    h, v = np.array(hs, np.uint8), np.array(vs, np.uint8)
    assert L.ifhip_jpeg_debug_encode_tables(tabs.ctypes.data, ncomp, h.ctypes.data, v.ctypes.data, width, height, quality, header.ctypes.data,
                                            header.size, C.byref(n)) == 0
    return tabs, header[:n.value].copy()


def emulate(j, quality, capacity=None):
    """j: the oracle's jpeg_read_coefficients dict.  Returns (file bytes or None, status, ownership violations)."""
    lib = emulator()
    ncomp = j["ncomp"]
    hs, vs = (list(j["hs"]) + [1, 1, 1])[:3], (list(j["vs"]) + [1, 1, 1])[:3]
    tabs, header = tables_and_header(ncomp, hs, vs, j["width"], j["height"], quality)
    bw, bh = np.array((list(j["bw"]) + [0, 0, 0])[:3], np.uint32), np.array((list(j["bh"]) + [0, 0, 0])[:3], np.uint32)
    h, v = np.array(hs, np.uint8), np.array(vs, np.uint8)
    planes = [np.ascontiguousarray(j["coef"][c], np.int16) if c < ncomp else None for c in range(3)]
    cap = capacity if capacity is not None else 1024 + 4 * sum(p.size for p in planes if p is not None) + 8192
    out, n, st, viol, win = np.zeros(cap, np.uint8), C.c_size_t(0), C.c_uint32(0), C.c_int(0), C.c_int(0)
    rc = lib.enc_emulate(*[p.ctypes.data if p is not None else None for p in planes], j["width"], j["height"], ncomp, h.ctypes.data,
                         v.ctypes.data, bw.ctypes.data, bh.ctypes.data, header.ctypes.data, header.size, tabs.ctypes.data, out.ctypes.data,
                         out.size, C.byref(n), C.byref(st), C.byref(viol), C.byref(win))
    assert rc == 0
    emulate.window_paths = win.value
    return (out[:n.value].tobytes() if n.value else None), st.value, viol.value


def photo(w, h, seed, noise=40):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 3 % 256)], -1).astype(np.int32)
    return np.clip(base + rng.integers(-noise, noise + 1, (h, w, 3)), 0, 255).astype(np.uint8)


def pillow_file(img, quality, sampling=None):
    buf = io.BytesIO()
    kw = {} if sampling is None else {"subsampling": sampling}
    Image.fromarray(img).save(buf, "JPEG", quality=quality, optimize=False, **kw)
    return buf.getvalue()


@pytest.mark.parametrize("sampling", ["4:2:0", "4:2:2", "4:4:4"])
@pytest.mark.parametrize("size", [(1, 1), (17, 9), (64, 48), (203, 131), (640, 360)])
@pytest.mark.parametrize("quality", [5, 75, 90, 100])
def test_emulated_passes_write_libjpeg_turbos_file(sampling, size, quality):
    w, h = size
    data = pillow_file(photo(w, h, w * 31 + h + quality), quality, sampling)
    j = O.jpeg_read_coefficients(data)
    assert (j["hs"], j["vs"]) == SAMPLINGS[sampling]
    out, status, violations = emulate(j, quality)
    assert status == 0 and violations == 0
    assert out == data


def test_grayscale():
    data = pillow_file(photo(150, 97, 3)[:, :, 0], 80)
    j = O.jpeg_read_coefficients(data)
    assert j["ncomp"] == 1
    out, status, violations = emulate(j, 80)
    assert (status, violations) == (0, 0) and out == data


def test_noise_at_q100_stuffs_bytes_across_chunks_and_workgroups():
    """Dense streams: many 0xFF bytes (stuffing in every 4 KiB chunk), blocks of a thousand bits, several workgroups of 256 blocks."""
    rng = np.random.default_rng(11)
    data = pillow_file(rng.integers(0, 256, (256, 384, 3), dtype=np.uint8), 100, "4:2:0")
    assert data.count(b"\xff\x00") > 100 and len(data) > 5 * 4096
    j = O.jpeg_read_coefficients(data)
    assert j["bw"][0] * j["bh"][0] > 4 * 256
    out, status, violations = emulate(j, 100)
    assert (status, violations) == (0, 0) and out == data
    assert emulate.window_paths <= 1                      # too dense for the LDS window: written to the stream word by word


def test_flat_image_has_blocks_of_a_few_bits():
    """Eight blocks per stream word: every word of the stream is shared."""
    data = pillow_file(np.full((128, 128, 3), 77, np.uint8), 90, "4:2:0")
    j = O.jpeg_read_coefficients(data)
    out, status, violations = emulate(j, 90)
    assert (status, violations) == (0, 0) and out == data
    assert emulate.window_paths == (16 * 16 * 6 // 4 + 255) // 256


def test_out_of_range_coefficient_drops_the_file():
    data = pillow_file(photo(40, 40, 5), 90, "4:4:4")
    j = O.jpeg_read_coefficients(data)
    j["coef"][1] = j["coef"][1].copy()
    j["coef"][1].reshape(-1)[64 * 3 + 5] = 1024                 # 11 magnitude bits in an AC coefficient
    out, status, _ = emulate(j, 90)
    assert out is None and status == 1
    j["coef"][1].reshape(-1)[64 * 3 + 5] = 0
    j["coef"][0] = j["coef"][0].copy()
    j["coef"][0].reshape(-1)[0] = -2048                          # DC difference of 12 bits against the next block
    j["coef"][0].reshape(-1)[64] = 2047
    out, status, _ = emulate(j, 90)
    assert out is None and status == 1


def test_file_capacity():
    data = pillow_file(photo(64, 64, 9), 90, "4:2:0")
    j = O.jpeg_read_coefficients(data)
    out, status, _ = emulate(j, 90, capacity=len(data))
    assert out == data and status == 0
    out, status, _ = emulate(j, 90, capacity=len(data) - 1)
    assert out is None and status == 4


def test_library_exports_the_device_coder():
    L = _native.lib()
    for name in ("ifhip_jpeg_enc_stage_create", "ifhip_jpeg_enc_stage_destroy", "ifhip_jpeg_enc_stage_max_file_bytes",
                 "ifhip_jpeg_encode_batch_device", "ifhip_jpeg_debug_encode_tables"):
        assert hasattr(L, name)
