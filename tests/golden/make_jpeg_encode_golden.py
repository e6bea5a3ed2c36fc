#!/usr/bin/env python3
"""Generate tests/golden/jpeg_encode_cases.npz: RGB source pictures and the baseline JPEG files Pillow / libjpeg-turbo
writes for them (optimize=False, islow DCT: the same pixel pipeline mozjpeg runs under set_fastest_defaults).  The
quantised coefficients inside each file are the independent pin for oracle/jpeg_oracle.c jo_jpeg_forward.
Run in the build container."""
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260922)


def picture(w, h, kind):
    y, x = np.mgrid[0:h, 0:w]
    if kind == "gradient":
        a = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1)
    elif kind == "noise":
        a = rng.integers(0, 256, size=(h, w, 3))
    elif kind == "extremes":                                  # saturated checker: exercises the rounding of every path
        a = np.where(((x // 3 + y // 2) % 2)[..., None] == 0, np.array([255, 0, 255]), np.array([0, 255, 0]))
    else:
        a = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 100 * np.cos(y / 5.0), 128 + 80 * np.sin((x + y) / 11.0)], -1)
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    out, names = {}, []
    i = 0
    sizes = [(64, 48), (37, 29), (16, 16), (8, 8), (1, 1), (9, 17), (23, 8), (31, 33), (100, 75), (7, 3)]
    settings = (("gradient", 90), ("noise", 75), ("waves", 30), ("extremes", 100))
    for (w, h) in sizes:
        for sub in ("4:4:4", "4:2:2", "4:2:0"):
            for kind, q in settings[(i % 2)::2] if (w * h > 2000) else settings:
                a = picture(w, h, kind)
                buf = io.BytesIO()
                Image.fromarray(a).save(buf, "JPEG", quality=q, subsampling=sub, optimize=False)
                out[f"src_{i}"] = a
                out[f"jpg_{i}"] = np.frombuffer(buf.getvalue(), np.uint8)
                names.append(f"{w}x{h}_{kind}_{sub}_q{q}")
                i += 1
    out["names"] = np.array(names)
    out["encoder"] = np.array(f"Pillow {Image.__version__} / libjpeg-turbo {features.version('jpg')}")
    np.savez_compressed(os.path.join(HERE, "jpeg_encode_cases.npz"), **out)
    print(len(names), "cases;", out["encoder"], os.path.getsize(os.path.join(HERE, "jpeg_encode_cases.npz")), "bytes")


if __name__ == "__main__":
    main()
