"""SeaHash (the `seahash` crate's default-seed `hash()`), restated from its published algorithm, and the bitmap checksum
of the reference's visual tests: `checksum_bitmap_window` (imageflow_core/tests/integration/common/mod.rs:307-324) =
seahash(w_le32 || h_le32 || rows without stride padding).  A `.checksums` id such as `novel-box-967914e71e:sea` carries
the FIRST ten hex digits of that 64-bit hash (established by test_oracle_reference_checksums.py on three canvases
whose pixels are known without any resampling).  Test infrastructure only.
"""
import struct

import numpy as np

_M = (1 << 64) - 1
_K = 0x6EED0E9DA4D94A4F


def _diffuse(x):
    x = (x * _K) & _M
    x ^= (x >> 32) >> (x >> 60)
    return (x * _K) & _M


def seahash(buf: bytes) -> int:
    st = [0x16F11FE89B0D677C, 0xB480A793D8E6C86C, 0x6FE2E5AAF078EBC9, 0x14F994A4C5259381]
    n = len(buf)
    full = n // 32
    if full:
        words = np.frombuffer(buf, dtype="<u8", count=full * 4).tolist()
        a, b, c, d = st
        for i in range(0, full * 4, 4):
            a = _diffuse(a ^ words[i])
            b = _diffuse(b ^ words[i + 1])
            c = _diffuse(c ^ words[i + 2])
            d = _diffuse(d ^ words[i + 3])
        st = [a, b, c, d]
    tail = buf[full * 32:]
    for i in range(0, len(tail), 8):
        st[i // 8] = _diffuse(st[i // 8] ^ int.from_bytes(tail[i:i + 8], "little"))
    return _diffuse(st[0] ^ st[1] ^ st[2] ^ st[3] ^ n)


def bitmap_checksum(img) -> str:
    """img: uint8 [h][w][4] BGRA without padding -> 16 hex digits."""
    h, w, _ = img.shape
    return "%016x" % seahash(struct.pack("<II", w, h) + np.ascontiguousarray(img).tobytes())


def checksum_id_digits(img) -> str:
    return bitmap_checksum(img)[:10]
