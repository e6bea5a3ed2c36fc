#!/usr/bin/env python3
"""Generate tests/golden/jpeg_entropy_cases.npz: baseline JPEG files that exercise the entropy stage beyond
jpeg_cases.npz -- restart intervals (per MCU row and every few MCUs), optimised (per-file) Huffman tables, high quality
noise (16-bit codes, long blocks), grayscale.  Written by Pillow / libjpeg-turbo; the expected coefficients come from
the oracle's serial decoder at test time.  Run in the build container."""
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260923)


def picture(w, h, kind):
    y, x = np.mgrid[0:h, 0:w]
    if kind == "noise":
        a = rng.integers(0, 256, size=(h, w, 3))
    elif kind == "gradient":
        a = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1)
    else:
        a = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 100 * np.cos(y / 5.0), 128 + 80 * np.sin((x + y) / 11.0)], -1)
        a = a + rng.integers(-20, 21, size=(h, w, 3))
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    out, names = {}, []
    i = 0
    for (w, h) in [(320, 200), (97, 61), (16, 16), (640, 33)]:
        for sub in ("4:2:0", "4:4:4", "4:2:2"):
            for kind, q, kw in (("noise", 95, dict(optimize=False)),
                                ("waves", 85, dict(optimize=True)),
                                ("gradient", 75, dict(optimize=False, restart_marker_rows=1)),
                                ("waves", 90, dict(optimize=True, restart_marker_blocks=3)),
                                ("noise", 50, dict(optimize=False, restart_marker_blocks=1))):
                buf = io.BytesIO()
                Image.fromarray(picture(w, h, kind)).save(buf, "JPEG", quality=q, subsampling=sub, **kw)
                out[f"jpg_{i}"] = np.frombuffer(buf.getvalue(), np.uint8)
                names.append(f"{w}x{h}_{kind}_{sub}_q{q}_" + "_".join(f"{k}={v}" for k, v in kw.items()))
                i += 1
    for kw in (dict(optimize=True), dict(restart_marker_rows=2)):
        buf = io.BytesIO()
        Image.fromarray(picture(200, 120, "waves")[..., 0]).save(buf, "JPEG", quality=80, **kw)
        out[f"jpg_{i}"] = np.frombuffer(buf.getvalue(), np.uint8)
        names.append("200x120_gray_q80_" + "_".join(f"{k}={v}" for k, v in kw.items()))
        i += 1
    buf = io.BytesIO()
    Image.fromarray(picture(48, 32, "waves")).save(buf, "JPEG", quality=80, progressive=True)
    out["progressive"] = np.frombuffer(buf.getvalue(), np.uint8)          # must be rejected (MethodNotImplemented)
    out["names"] = np.array(names)
    out["encoder"] = np.array(f"Pillow {Image.__version__} / libjpeg-turbo {features.version('jpg')}")
    path = os.path.join(HERE, "jpeg_entropy_cases.npz")
    np.savez_compressed(path, **out)
    print(len(names), "cases;", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
