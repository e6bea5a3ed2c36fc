"""Pin oracle/jpeg_oracle.c (dequant + islow IDCT + fancy up-sampling + YCbCr->BGRA, plus the test-only baseline
entropy decoder) against an independent implementation of the same algorithm family: Pillow/libjpeg-turbo's decode of
the committed files (tests/golden/jpeg_cases.npz, made by tests/golden/make_jpeg_golden.py).  Byte equality."""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    return z, [str(n) for n in z["names"]]


def test_oracle_decode_equals_libjpeg_turbo_on_all_committed_files(cases):
    z, names = cases
    assert len(names) == 36
    for i, name in enumerate(names):
        data = z[f"jpg_{i}"].tobytes()
        ref = z[f"rgb_{i}"]
        j = O.jpeg_read_coefficients(data)
        h, w = ref.shape[:2]
        assert (j["width"], j["height"]) == (w, h), name
        out = O.jpeg_idct_color(j)
        px = out[:, :4 * w].reshape(h, w, 4)
        assert np.array_equal(px[..., 2], ref[..., 0]) and np.array_equal(px[..., 1], ref[..., 1]) \
            and np.array_equal(px[..., 0], ref[..., 2]), name
        assert np.all(px[..., 3] == 255), name                   # JCS_EXT_BGRA writes opaque alpha
        assert np.all(out[:, 4 * w:] == 0)                       # row padding untouched


def test_oracle_decode_equals_live_pillow_when_available():
    PIL = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(5)
    for (w, h, sub) in ((97, 61, "4:2:0"), (33, 9, "4:2:2"), (24, 24, "4:4:4")):
        a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        buf = io.BytesIO()
        PIL.fromarray(a).save(buf, "JPEG", quality=70, subsampling=sub, optimize=False)
        ref = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGB"))
        out = O.jpeg_idct_color(O.jpeg_read_coefficients(buf.getvalue()))[:, :4 * w].reshape(h, w, 4)
        assert np.array_equal(out[..., [2, 1, 0]], ref)


def test_idct_known_answers():
    q = np.ones(64, np.uint16)
    blk = np.zeros(64, np.int16)
    assert np.all(O.idct_islow_block(blk, q) == 128)             # all-zero block -> mid grey
    blk[0] = 8 * 100                                             # DC only: +100 everywhere
    assert np.all(O.idct_islow_block(blk, q) == 228)
    blk[0] = 8 * 200
    assert np.all(O.idct_islow_block(blk, q) == 255)             # range limit
    blk[0] = -8 * 200
    assert np.all(O.idct_islow_block(blk, q) == 0)
    # against a float DCT-III on random in-range blocks: the fixed-point IDCT is within 1 of the exact transform
    rng = np.random.default_rng(0)
    k = np.arange(8)
    basis = np.cos((2 * k[:, None] + 1) * k[None, :] * np.pi / 16) * np.where(k == 0, np.sqrt(0.5), 1.0)[None, :] * 0.5
    for _ in range(50):
        co = rng.integers(-60, 60, size=(8, 8)).astype(np.int16)
        exact = basis @ co.astype(np.float64) @ basis.T + 128
        got = O.idct_islow_block(co.reshape(64), q).astype(np.float64)
        assert np.max(np.abs(got - np.clip(np.round(exact), 0, 255))) <= 1


def test_reference_block_scalers_golden_is_self_consistent(golden_dir):
    """oracle/_ref: the reference's own compiled c_components/lib/codecs_jpeg_idct_fast.c.  The fixture holds its outputs
    on seeded blocks; the reference KAT (c_components/tests/test_idct_scaling.rs:5-19) must be in it."""
    z = np.load(os.path.join(golden_dir, "ref_block_scalers.npz"))
    assert int(z["flow_scale_spatial_srgb_1x1"][0, 0, 0]) == 188
    assert np.all(z["flow_scale_spatial_1x1"][1] == 0) and np.all(z["flow_scale_spatial_7x7"][2] == 255)
    so = os.path.join(os.path.dirname(golden_dir), "..", "oracle", "_ref", "libref_idct.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built here")
    import ctypes
    lib = ctypes.CDLL(so)
    blocks = z["blocks"]
    for n in (1, 4, 7):
        for srgb in (0, 1):
            name = f"flow_scale_spatial_{'srgb_' if srgb else ''}{n}x{n}"
            fn = getattr(lib, name)
            for b in (0, 5, 33):
                rows = [np.zeros(8, np.uint8) for _ in range(n)]
                ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(*[r.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) for r in rows])
                blk = np.ascontiguousarray(blocks[b])
                fn(blk.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ptrs, ctypes.c_uint32(0))
                got = np.stack([r[:n] for r in rows])
                assert np.array_equal(got, z[name][b])


def test_reduced_size_decode_equals_libjpeg_turbo_draft_mode(cases):
    """scale_num 4, 2, 1 with libjpeg's own reduced IDCTs (luma_mode 0): jidctred.c's 4x4 / 2x2 / 1x1 and jdmaster.c's
    'sub-sampled chroma takes the twice-larger IDCT' rule, pinned byte for byte by Pillow's draft-mode decode."""
    z, names = cases
    checked = 0
    for i, name in enumerate(names):
        j = None
        for num in (4, 2, 1):
            key = f"rgb_{i}_s{num}"
            if key not in z.files:
                continue
            j = j or O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
            ref = z[key]
            oh, ow = ref.shape[:2]
            out = O.jpeg_idct_color_scaled(j, num, 0)
            assert np.array_equal(out[:, : 4 * ow].reshape(oh, ow, 4)[..., [2, 1, 0]], ref), (name, num)
            checked += 1
    assert checked >= 60


def test_spatial_block_scaler_restatement_equals_reference_outputs(golden_dir):
    """jo_scale_spatial_block over the reference's tables == the reference's compiled functions (committed outputs)."""
    z = np.load(os.path.join(golden_dir, "ref_block_scalers.npz"))
    for srgb in (0, 1):
        for n in range(1, 8):
            name = f"flow_scale_spatial_{'srgb_' if srgb else ''}{n}x{n}"
            assert np.array_equal(O.scale_spatial_blocks(z["blocks"], n, srgb), z[name]), name
