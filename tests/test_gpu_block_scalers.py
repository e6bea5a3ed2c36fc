"""flow_scale_spatial[_srgb]_NxN on the GPU against the reference's OWN code: the committed outputs of its compiled
functions (tests/golden/ref_block_scalers.npz, incl. the KAT 188 of c_components/tests/test_idct_scaling.rs:5-19) and,
when oracle/_ref/libref_idct.so travelled along, the compiled functions themselves on fresh random blocks."""
import ctypes
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs import block_scalers as BS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_kat_188():
    blk = np.tile(np.array([0, 255] * 4, np.uint8), 8)[None]
    assert int(BS.flow_scale_spatial(blk, 1, True)[0, 0, 0]) == 188


def test_all_14_functions_match_committed_reference_outputs(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_block_scalers.npz"))
    blocks = z["blocks"]
    for srgb in (0, 1):
        for n in range(1, 8):
            name = f"flow_scale_spatial_{'srgb_' if srgb else ''}{n}x{n}"
            got = BS.flow_scale_spatial(blocks, n, srgb)
            assert np.array_equal(got, z[name]), name


def _ref_lib():
    so = os.path.join(ROOT, "oracle", "_ref", "libref_idct.so")
    return ctypes.CDLL(so) if os.path.exists(so) else None


def test_against_the_compiled_reference_on_fresh_blocks():
    lib = _ref_lib()
    if lib is None:
        pytest.skip("oracle/_ref/libref_idct.so not present")
    rng = np.random.default_rng(99)
    blocks = rng.integers(0, 256, size=(500, 64), dtype=np.uint8)
    blocks[:8] = np.array([0, 1, 2, 127, 128, 253, 254, 255], np.uint8)[:, None]
    for srgb in (0, 1):
        for n in range(1, 8):
            fn = getattr(lib, f"flow_scale_spatial_{'srgb_' if srgb else ''}{n}x{n}")
            exp = np.zeros((len(blocks), n, n), np.uint8)
            for b, blk in enumerate(blocks):
                rows = [np.zeros(8, np.uint8) for _ in range(n)]
                ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(*[r.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) for r in rows])
                inp = np.ascontiguousarray(blk)
                fn(inp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ptrs, ctypes.c_uint32(0))
                for r in range(n):
                    exp[b, r] = rows[r][:n]
            assert np.array_equal(BS.flow_scale_spatial(blocks, n, srgb), exp), (n, srgb)


def test_plane_form_equals_block_form():
    rng = np.random.default_rng(3)
    bw, bh = 30, 17
    plane = rng.integers(0, 256, size=(8 * bh, 8 * bw), dtype=np.uint8)
    blocks = plane.reshape(bh, 8, bw, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
    for n, srgb in ((1, 1), (4, 1), (7, 0), (3, 0)):
        out = BS.flow_scale_spatial_plane(torch.from_numpy(plane).to("cuda:0"), n, srgb)
        torch.cuda.synchronize()
        exp = BS.flow_scale_spatial(blocks, n, srgb).reshape(bh, bw, n, n).transpose(0, 2, 1, 3).reshape(bh * n, bw * n)
        assert np.array_equal(out.cpu().numpy(), exp)
