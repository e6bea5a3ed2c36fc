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
def test_decode_tables_and_scan_layout_without_a_device(golden_dir, debug_switch, pool):
    """What the entropy kernels rely on, checked on the host for every committed file (ifhip_jpeg_debug_scan_report):
    a serial walk with the two-level tables starts exactly the blocks the geometry asks for and ends on the DC values of
    the oracle's decode; walks with the pair tables (synchronisation rounds) and with the count tables pass through the
    same state at every sub-sequence boundary as the plain walk.  Also with a second-level pool of 0 / 40 entries, which
    leaves long prefixes to the serial code search."""
    if pool is not None:
        debug_switch("ent_test_pool", pool)
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


# ---- EXIF orientation (ifhip_jpeg_exif_orientation; codecs/mozjpeg_decoder_helpers.rs:107-202) ----------------------------
def _exif_app1(value=6, little=True, tag_type=3, count=1, lead=0, ifd_offset=8, pad_to=40, extra_tags=2):
    """An APP1 "Exif" segment: header, `lead` junk bytes, TIFF header, IFD0 with `extra_tags` other tags and 0x0112."""
    import struct
    e = "<" if little else ">"
    tiff = (b"II\x2a\x00" if little else b"MM\x00\x2a") + struct.pack(e + "I", ifd_offset)
    tiff += b"\0" * (ifd_offset - 8)
    tiff += struct.pack(e + "H", extra_tags + 1)
    for k in range(extra_tags):
        tiff += struct.pack(e + "HHI", 0x010F + k, 2, 4) + b"abc\0"
    tiff += struct.pack(e + "HHI", 0x0112, tag_type, count) + struct.pack(e + "H", value) + b"\0\0"
    tiff += struct.pack(e + "I", 0)
    data = b"Exif\0\0" + b"\x01" * lead + tiff
    data += b"\0" * max(0, pad_to - len(data))
    return b"\xff\xe1" + struct.pack(">H", len(data) + 2) + data


def _exif_flag(jpeg):
    import ctypes as C
    L = _native.lib()
    L.ifhip_jpeg_exif_orientation.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    flag = C.c_int(-7)
    assert L.ifhip_jpeg_exif_orientation(jpeg, len(jpeg), C.byref(flag)) == 0
    return flag.value


def _with_segments(jpeg, *segs):
    return jpeg[:2] + b"".join(segs) + jpeg[2:]


def test_exif_orientation_follows_the_reference_parser(golden_dir):
    from imageflow_amd import _native
    globals()["_native"] = _native
    base = next(all_files(golden_dir))[1]
    assert _exif_flag(base) == -1                                                    # no EXIF: None
    for little in (True, False):
        for v in range(0, 9):
            assert _exif_flag(_with_segments(base, _exif_app1(v, little))) == v      # 0..8 are Some(v), 0 included (:191)
        assert _exif_flag(_with_segments(base, _exif_app1(9, little))) == -1         # above 8: None
        assert _exif_flag(_with_segments(base, _exif_app1(300, little))) == -1
    # `tag_type != 3 && count != 1` is the only rejection (:187): one of the two may be off
    assert _exif_flag(_with_segments(base, _exif_app1(6, tag_type=4, count=1))) == 6
    assert _exif_flag(_with_segments(base, _exif_app1(6, tag_type=3, count=2))) == 6
    assert _exif_flag(_with_segments(base, _exif_app1(6, tag_type=4, count=2))) == -1
    # the TIFF header is searched at offsets 0..15 of the marker data (:136-145); "Exif\0\0" occupies 0..5
    assert _exif_flag(_with_segments(base, _exif_app1(5, lead=9))) == 5               # header at offset 15
    assert _exif_flag(_with_segments(base, _exif_app1(5, lead=10))) == -1             # at 16: not found
    # shorter than 32 bytes: "EXIF too short" (:116-121) -- and the FIRST Exif marker decides
    def cut(seg, n):                                                                  # the segment without its last n bytes
        return seg[:2] + (len(seg) - 2 - n).to_bytes(2, "big") + seg[4:len(seg) - n]
    full = _exif_app1(3, extra_tags=0, pad_to=0)
    short = cut(full, 4)                                                              # (without the next-IFD pointer: 28 bytes)
    assert len(full) - 4 == 32 and _exif_flag(_with_segments(base, full)) == 3 and _exif_flag(_with_segments(base, short)) == -1
    assert _exif_flag(_with_segments(base, short, _exif_app1(6))) == -1
    assert _exif_flag(_with_segments(base, _exif_app1(8), _exif_app1(6))) == 8
    # other APP1 / APP2 content in front is skipped; an IFD offset below 4 reads from the start (max(4, offset) - 4, :173)
    xmp = b"\xff\xe1\x00\x20" + b"http://ns.adobe.com/xap/1.0/\0x"
    icc = b"\xff\xe2\x00\x12" + b"ICC_PROFILE\0\x01\x01ab"
    assert _exif_flag(_with_segments(base, xmp, icc, _exif_app1(7))) == 7
    # IFD that points behind the data, and a tag list that runs off the end: io errors are None (:150 unwrap_or(None))
    assert _exif_flag(_with_segments(base, _exif_app1(6, ifd_offset=4000))) == 6     # IFD0 far behind the header
    seg = bytearray(_exif_app1(6))
    seg[14:18] = (4000).to_bytes(4, "little")                                          # ... and one that is not there
    assert _exif_flag(_with_segments(base, bytes(seg))) == -1
    seg[14:18] = (2).to_bytes(4, "little")                                             # max(4, offset) - 4: read as offset 4, i.e. the
    assert _exif_flag(_with_segments(base, bytes(seg))) == -1                          # offset field's second half as the tag count
    assert _exif_flag(_with_segments(base, cut(_exif_app1(6, extra_tags=1, pad_to=0), 8))) == -1   # ends inside the tag's value
    # an Exif marker behind SOS is never seen (jpeg_read_header stops at the scan)
    sos = base.index(b"\xff\xda")
    assert _exif_flag(base[:sos] + base[sos:-2] + _exif_app1(6) + b"\xff\xd9") == -1



# ---- embedded ICC profile (ifhip_jpeg_icc_profile_kind; codecs/mozjpeg_decoder.rs:370-420) -----------------------------------
SRGB_XYZ = ((0.4360, 0.2225, 0.0139), (0.3851, 0.7169, 0.0971), (0.1431, 0.0606, 0.7141))
P3_XYZ = ((0.5151, 0.2412, -0.0011), (0.2920, 0.6922, 0.0419), (0.1571, 0.0666, 0.7841))


def make_icc(xyz=SRGB_XYZ, trc="para", space=b"RGB ", pcs=b"XYZ "):
    """A minimal matrix/TRC ICC profile: header, tag table, rXYZ gXYZ bXYZ, one tone-curve element shared by r/g/bTRC."""
    import struct

    def s15(v):
        return struct.pack(">i", int(round(v * 65536)))
    if trc == "para":
        curve = b"para" + b"\0" * 4 + struct.pack(">HH", 3, 0) + b"".join(s15(v) for v in (2.4, 1 / 1.055, 0.055 / 1.055, 1 / 12.92, 0.04045))
    elif trc == "curv1024":
        n = 1024
        xs = [k / (n - 1) for k in range(n)]
        ys = [x / 12.92 if x <= 0.04045 else ((x + 0.055) / 1.055) ** 2.4 for x in xs]
        curve = b"curv" + b"\0" * 4 + struct.pack(">I", n) + b"".join(struct.pack(">H", int(round(y * 65535))) for y in ys)
    elif trc == "gamma22":
        curve = b"curv" + b"\0" * 4 + struct.pack(">I", 1) + struct.pack(">H", int(2.2 * 256)) + b"\0\0"
    else:
        raise ValueError(trc)
    elems = [b"XYZ " + b"\0" * 4 + b"".join(s15(v) for v in c) for c in xyz] + [curve]
    sigs = [b"rXYZ", b"gXYZ", b"bXYZ", b"rTRC", b"gTRC", b"bTRC"]
    which = [0, 1, 2, 3, 3, 3]
    table_end = 128 + 4 + 12 * len(sigs)
    offs, body = [], b""
    for e in elems:
        offs.append(table_end + len(body))
        body += e + b"\0" * (-len(e) % 4)
    table = struct.pack(">I", len(sigs)) + b"".join(sig + struct.pack(">II", offs[w], len(elems[w])) for sig, w in zip(sigs, which))
    size = table_end + len(body)
    header = struct.pack(">I", size) + b"test" + bytes([4, 0x30, 0, 0]) + b"mntr" + space + pcs + b"\0" * 12 + b"acsp" + b"\0" * (128 - 40)
    assert len(header) == 128
    return header + table + body


def icc_app2(profile, pieces=1, drop=None, dup=False):
    import struct
    step = -(-len(profile) // pieces)
    segs = []
    for k in range(pieces):
        part = profile[k * step:(k + 1) * step]
        data = b"ICC_PROFILE\0" + bytes([k + 1, pieces]) + part
        segs.append(b"\xff\xe2" + struct.pack(">H", len(data) + 2) + data)
    if drop is not None:
        del segs[drop]
    if dup:
        segs.append(segs[0])
    return b"".join(segs)


def icc_kind(jpeg):
    import ctypes as C
    from imageflow_amd import _native
    L = _native.lib()
    L.ifhip_jpeg_icc_profile_kind.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    kind = C.c_int(-7)
    assert L.ifhip_jpeg_icc_profile_kind(jpeg, len(jpeg), C.byref(kind)) == 0
    return kind.value


def test_icc_profile_kind(golden_dir):
    """0 = no profile (the only case the reference does not run its CMS for, mozjpeg_decoder.rs:409) -- which includes chunk
    sets its reassembly drops (mozjpeg_decoder_helpers.rs:42-83) and a GRAY profile on a colour frame (mozjpeg_decoder.rs:391-395);
    1 = a profile that IS sRGB (matrix profile, sRGB primaries and tone curve); 2 = anything else."""
    base = next(all_files(golden_dir))[1]
    assert icc_kind(base) == 0
    assert icc_kind(_with_segments(base, icc_app2(make_icc()))) == 1
    assert icc_kind(_with_segments(base, icc_app2(make_icc(trc="curv1024")))) == 1
    assert icc_kind(_with_segments(base, _exif_app1(6), icc_app2(make_icc(), pieces=3))) == 1          # reassembled from three markers
    assert icc_kind(_with_segments(base, icc_app2(make_icc(xyz=P3_XYZ)))) == 2                          # Display P3 primaries
    assert icc_kind(_with_segments(base, icc_app2(make_icc(trc="gamma22")))) == 2                       # sRGB primaries, plain gamma
    assert icc_kind(_with_segments(base, icc_app2(make_icc(space=b"CMYK")))) == 2
    assert icc_kind(_with_segments(base, icc_app2(make_icc(space=b"GRAY")))) == 0                       # grey profile, colour frame: Srgb
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.new("L", (16, 16), 128).save(buf, "JPEG")
    assert icc_kind(_with_segments(buf.getvalue(), icc_app2(make_icc(space=b"GRAY")))) == 2               # ... on a grey frame: IccProfileGray
    assert icc_kind(_with_segments(base, icc_app2(make_icc(xyz=P3_XYZ), pieces=3, drop=1))) == 0        # a chunk is missing: None
    assert icc_kind(_with_segments(base, icc_app2(make_icc(xyz=P3_XYZ), pieces=2, dup=True))) == 0      # a chunk twice: None
    assert icc_kind(_with_segments(base, b"\xff\xe2\x00\x10ICC_PROFILE\0\x01\x01")) == 0                # only an empty marker: None
    assert icc_kind(_with_segments(base, b"\xff\xe2\x00\x12ICC_PROFILE\0\x02\x01ab")) == 0              # sequence number past the count
    assert icc_kind(_with_segments(base, icc_app2(make_icc()[:100]))) == 2                              # shorter than a header
    nearly = make_icc(xyz=((0.4360 + 0.004, 0.2225, 0.0139),) + SRGB_XYZ[1:])                           # a primary off by 0.004
    assert icc_kind(_with_segments(base, icc_app2(nearly))) == 2
    sos = base.index(b"\xff\xda")                                                                       # behind SOS: never seen
    assert icc_kind(base[:sos] + base[sos:-2] + icc_app2(make_icc(xyz=P3_XYZ)) + b"\xff\xd9") == 0
