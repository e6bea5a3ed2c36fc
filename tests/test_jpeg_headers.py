"""Host-only part of the entropy stage: ifhip_jpeg_parse_headers (csrc/jpeg_entropy.hip) against the oracle's parser on
every committed file, and the oracle's serial Huffman decoder against libjpeg-turbo (Pillow) on the restart-interval /
optimised-table fixtures that the GPU decoder is checked with."""
import io
import os

import numpy as np
import pytest

from oracle import oracle as O
from imageflow_amd.codecs import mozjpeg_decoder as D
from imageflow_amd.errors import FlowError


def all_files(golden_dir):
    for name in ("jpeg_cases.npz", "jpeg_encode_cases.npz", "jpeg_entropy_cases.npz"):
        z = np.load(os.path.join(golden_dir, name))
        for i, n in enumerate(z["names"]):
            yield f"{name}:{n}", z[f"jpg_{i}"].tobytes()


def test_headers_match_the_oracle_parser(golden_dir):
    count = 0
    for name, data in all_files(golden_dir):
        info = D.get_image_info(data)
        j = O.jpeg_read_coefficients(data)
        n = j["ncomp"]
        assert (info["width"], info["height"], info["ncomp"]) == (j["width"], j["height"], n), name
        assert info["hs"] == j["hs"][:n] and info["vs"] == j["vs"][:n], name
        assert info["bw"] == j["bw"][:n] and info["bh"] == j["bh"][:n], name
        assert np.array_equal(info["qt"][:n], j["qt"][:n]), name
        count += 1
    assert count == 36 + 108 + 62


def test_rejections(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    with pytest.raises(FlowError) as e:
        D.get_image_info(z["progressive"].tobytes())
    assert "MethodNotImplemented" in str(e.value)
    with pytest.raises(FlowError):
        D.get_image_info(b"\\x89PNG\\r\\n\\x1a\\n" + bytes(64))
    data = z["jpg_0"].tobytes()
    with pytest.raises(FlowError):
        D.get_image_info(data[:200])                      # truncated inside the tables


def test_oracle_decoder_equals_libjpeg_turbo_on_the_entropy_fixtures(golden_dir):
    PIL = pytest.importorskip("PIL.Image")
    z = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    for i, name in enumerate(z["names"]):
        data = z[f"jpg_{i}"].tobytes()
        ref = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
        j = O.jpeg_read_coefficients(data)
        h, w = ref.shape[:2]
        px = O.jpeg_idct_color(j)[:, :4 * w].reshape(h, w, 4)
        assert np.array_equal(px[..., [2, 1, 0]], ref), name


def test_rgb_coded_files_are_left_to_libjpeg():
    PIL = pytest.importorskip("PIL.Image")
    a = np.random.default_rng(0).integers(0, 256, size=(24, 32, 3), dtype=np.uint8)
    buf = io.BytesIO()
    try:
        PIL.fromarray(a).save(buf, "JPEG", quality=90, keep_rgb=True)           # Adobe marker, transform 0
    except TypeError:
        pytest.skip("this Pillow cannot write RGB-coded JPEG")
    with pytest.raises(FlowError) as e:
        D.get_image_info(buf.getvalue())
    assert "RGB-coded" in str(e.value)


def _patch_sampling(data, factors):
    """Rewrite the sampling bytes of the SOF0 segment: factors = [(h, v)] * 3."""
    b = bytearray(data)
    i = 2
    while i + 4 <= len(b):
        assert b[i] == 0xFF
        m, seg = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xC0:
            for c, (h, v) in enumerate(factors):
                b[i + 4 + 6 + 3 * c + 1] = (h << 4) | v
            return bytes(b)
        i += 2 + seg
    raise AssertionError("no SOF0")


def test_oversized_mcu_is_rejected_by_the_parser(golden_dir):
    """A crafted 2x2 / 2x2 / 2x2 header is 12 blocks per MCU: libjpeg refuses it (D_MAX_BLOCKS_IN_MCU = 10) and so must
    the parser, BEFORE the entropy stage sizes its per-MCU block tables from it (round-1 advisor finding: device
    out-of-bounds write driven by file bytes)."""
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    data = next(z[f"jpg_{i}"].tobytes() for i, n in enumerate(z["names"]) if "420" in str(n) or True)
    info = D.get_image_info(data)
    if info["ncomp"] != 3:
        pytest.skip("first fixture is grayscale")
    for factors in ([(2, 2)] * 3, [(2, 2), (2, 1), (1, 1)], [(1, 1), (2, 2), (2, 2)], [(2, 1), (1, 2), (1, 1)]):
        with pytest.raises(FlowError) as e:
            D.get_image_info(_patch_sampling(data, factors))
        assert "MethodNotImplemented" in str(e.value) or "ImageMalformed" in str(e.value), str(e.value)


def test_parser_survives_mutated_headers(golden_dir):
    """Untrusted bytes: every single-byte mutation of the marker segments (and truncations) of a few files either parses or
    is refused with a FlowError -- never a crash, never geometry the decoder's tables cannot hold."""
    z = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    rng = np.random.default_rng(5)
    tried = refused = 0
    for i in (0, 7, 20, 33):
        data = bytearray(z[f"jpg_{i}"].tobytes())
        sos = bytes(data).find(b"\xff\xda")
        assert sos > 0
        for _ in range(150):
            m = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(2, sos + 14))] = int(rng.integers(0, 256))
            if rng.integers(0, 5) == 0:
                m = m[: int(rng.integers(4, len(m)))]
            tried += 1
            try:
                info = D.get_image_info(bytes(m))
            except FlowError:
                refused += 1
                continue
            assert 1 <= info["ncomp"] <= 3 and 0 < info["width"] <= 65535 and 0 < info["height"] <= 65535
            assert sum(h * v for h, v in zip(info["hs"][:info["ncomp"]], info["vs"][:info["ncomp"]])) <= 10
    assert tried == 600 and 0 < refused < tried


import ctypes as _C


class _ScanReport(_C.Structure):
    _fields_ = ([(n, _C.c_uint32) for n in ("segments", "sub_sequences", "scan_complete", "pool_entries", "pool_entries_used",
                                            "prefixes_left_to_search", "pair_entries", "segments_with_wrong_block_count",
                                            "segments_with_invalid_codes", "pair_walk_mismatches", "count_walk_mismatches")]
                + [("_pad", _C.c_uint32)] + [(n, _C.c_uint64) for n in ("symbols", "table_reads_with_pairs", "blocks")]
                + [("dc_sum", _C.c_int32 * 3), ("dc_last_segment", _C.c_int32 * 3)])


def _scan_report(data):
    import ctypes as C
    from imageflow_amd import _native
    L = _native.lib()
    L.ifhip_jpeg_debug_scan_report.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_ScanReport)]
    r = _ScanReport()
    _native.check(L.ifhip_jpeg_debug_scan_report(data, len(data), C.byref(r)))
    return r


@pytest.mark.parametrize("pool", [None, "0", "40"])
def test_decode_tables_and_scan_layout_without_a_device(golden_dir, monkeypatch, pool):
    """What the entropy kernels rely on, checked on the host for every committed file (ifhip_jpeg_debug_scan_report):
    a serial walk with the two-level tables starts exactly the blocks the geometry asks for and ends on the DC values of
    the oracle's decode; walks with the pair tables (synchronisation rounds) and with the count tables pass through the
    same state at every sub-sequence boundary as the plain walk.  Also with a second-level pool of 0 / 40 entries, which
    leaves long prefixes to the serial code search."""
    if pool is not None:
        monkeypatch.setenv("IFHIP_ENT_TEST_POOL", pool)
    files = searched = paired = 0
    reads = symbols = 0
    for name, data in all_files(golden_dir):
        r = _scan_report(data)
        j = O.jpeg_read_coefficients(data)
        n = j["ncomp"]
        assert r.scan_complete == 1 and r.segments >= 1, name
        assert (r.segments_with_wrong_block_count, r.segments_with_invalid_codes) == (0, 0), name
        assert (r.pair_walk_mismatches, r.count_walk_mismatches) == (0, 0), name
        assert r.blocks == sum(j["bw"][c] * j["bh"][c] for c in range(n)), name
        assert list(r.dc_last_segment)[:n] == [int(j["coef"][c][-1, -1, 0]) for c in range(n)], name
        assert r.pool_entries_used <= r.pool_entries == 768 and r.table_reads_with_pairs <= r.symbols, name
        files += 1
        searched += r.prefixes_left_to_search > 0
        paired += r.pair_entries > 0
        reads += r.table_reads_with_pairs
        symbols += r.symbols
    assert files == 206 and paired == files
    assert (searched > 0) == (pool is not None)          # the committed files fit the pool; the shrunken pools do not hold them
    assert reads < 0.8 * symbols                          # one read covers two symbols often enough
