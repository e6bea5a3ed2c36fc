#!/usr/bin/env python3
"""Records what libjpeg decodes at scale_num/8 for scale_num in {1..6, 8} -- the sizes MzDec::apply_downscaling can ask
for (imageflow_core/src/codecs/mozjpeg_decoder.rs:603-617) -- into tests/golden/jpeg_scaled_cases.npz.

Reference implementation: the SYSTEM libjpeg (/lib/x86_64-linux-gnu/libjpeg.so.8 = libjpeg-turbo 2.1.2; mozjpeg-sys 2.2.3,
the reference's dependency, derives from the same jdmaster.c / jidctint.c / jdsample.c), driven through
tests/golden/libjpeg_driver.c in a separate process (no headers are installed; Pillow's bundled libjpeg-turbo 3.1.4 only
exposes 1/1, 1/2, 1/4, 1/8 -- those four agree with the driver, checked below).  Files are written with Pillow;
4:4:0 (h1v2) files, which Pillow cannot write, are 4:2:2 files of square size with the SOF sampling bytes swapped (same
block count per MCU and same MCU count, so the scan stays decodable; the picture is scrambled, the arithmetic is not).

Run here (needs gcc + the system libjpeg):  python tests/golden/make_jpeg_scaled_golden.py
"""
import io
import os
import struct
import subprocess
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SCALES = (1, 2, 3, 4, 5, 6, 8)


def build_driver(tmp):
    exe = os.path.join(tmp, "ljd")
    subprocess.run(["gcc", "-O1", "-DLJD_MAIN", "-o", exe, os.path.join(HERE, "libjpeg_driver.c"),
                    "/lib/x86_64-linux-gnu/libjpeg.so.8"], check=True)
    return exe


def decode(exe, tmp, data, scale):
    src, dst = os.path.join(tmp, "a.jpg"), os.path.join(tmp, "o.bin")
    open(src, "wb").write(data)
    subprocess.run([exe, src, str(scale), "1", dst], check=True)
    raw = open(dst, "rb").read()
    w, h = struct.unpack("<II", raw[:8])
    return np.frombuffer(raw, np.uint8, w * h * 3, 8).reshape(h, w, 3).copy()


def swap_sampling_to_440(data):
    b = bytearray(data)
    i = 2
    while i + 4 <= len(b):
        m, seg = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xC0:
            assert b[i + 4 + 6 + 1] == 0x21, "expected a 4:2:2 file"
            b[i + 4 + 6 + 1] = 0x12
            return bytes(b)
        i += 2 + seg
    raise AssertionError("no SOF0")


def main():
    rng = np.random.default_rng(20260921)
    names, files, refs = [], [], {}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_driver(tmp)
        cases = []
        for (w, h) in ((83, 61), (64, 48), (17, 9), (33, 16), (3, 3), (5, 2), (1, 1), (130, 70)):
            y, x = np.mgrid[0:h, 0:w]
            grad = np.stack([(x * 7) % 256, (y * 5) % 256, ((x + y) * 3) % 256], -1).astype(np.uint8)
            noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            for content, img in (("grad", grad), ("noise", noise)):
                for ss in ("4:2:0", "4:2:2", "4:4:4", "gray"):
                    if content == "noise" and ss == "4:4:4" and w > 64:
                        continue
                    b = io.BytesIO()
                    if ss == "gray":
                        Image.fromarray(img[..., 0]).save(b, "JPEG", quality=88)
                    else:
                        Image.fromarray(img).save(b, "JPEG", quality=88, subsampling=ss)
                    cases.append((f"{w}x{h}_{content}_{ss.replace(':', '')}", b.getvalue()))
        for side in (48, 83, 9, 2):
            img = rng.integers(0, 256, (side, side, 3), dtype=np.uint8)
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=90, subsampling="4:2:2")
            cases.append((f"{side}x{side}_noise_440", swap_sampling_to_440(b.getvalue())))
        for i, (name, data) in enumerate(cases):
            names.append(name)
            files.append(np.frombuffer(data, np.uint8))
            for s in SCALES:
                refs[f"ref_{i}_{s}"] = decode(exe, tmp, data, s)
            if not name.endswith("_440"):                      # cross-check of the driver against Pillow's own decoder
                full = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
                assert np.array_equal(full, refs[f"ref_{i}_8"]), name
    out = {"names": np.array(names), "scales": np.array(SCALES)}
    for i, f in enumerate(files):
        out[f"jpg_{i}"] = f
    out.update(refs)
    path = os.path.join(HERE, "jpeg_scaled_cases.npz")
    np.savez_compressed(path, **out)
    print(len(names), "files,", os.path.getsize(path), "bytes ->", path)


if __name__ == "__main__":
    main()
