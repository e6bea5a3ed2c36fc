#!/usr/bin/env python3
"""Generate tests/golden/jpeg_cases.npz: small baseline JPEG files (Pillow / libjpeg-turbo encoder, optimize=False) and
Pillow's own decode of each (the independent pin for oracle/jpeg_oracle.c).  Run in the build container."""
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260921)


def picture(w, h, kind):
    y, x = np.mgrid[0:h, 0:w]
    if kind == "gradient":     # bench_codecs.rs:24-41 style gradient
        a = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1)
    elif kind == "noise":
        a = rng.integers(0, 256, size=(h, w, 3))
    else:
        a = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 100 * np.cos(y / 5.0), 128 + 80 * np.sin((x + y) / 11.0)], -1)
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    out = {}
    names = []
    i = 0
    for (w, h) in [(64, 48), (37, 29), (16, 16), (8, 8), (1, 1), (250, 130), (17, 33)]:
        for kind, sub, q in (("gradient", "4:2:0", 85), ("noise", "4:2:0", 85), ("waves", "4:4:4", 90),
                             ("noise", "4:2:2", 75), ("waves", "4:2:0", 30)):
            buf = io.BytesIO()
            Image.fromarray(picture(w, h, kind)).save(buf, "JPEG", quality=q, subsampling=sub, optimize=False)
            data = buf.getvalue()
            ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
            out[f"jpg_{i}"] = np.frombuffer(data, np.uint8)
            out[f"rgb_{i}"] = ref
            if sub in ("4:2:0", "4:4:4"):          # libjpeg's reduced-size decode (draft mode): scale 4/8, 2/8, 1/8
                for denom, num in ((2, 4), (4, 2), (8, 1)):
                    im = Image.open(io.BytesIO(data))
                    im.draft("RGB", (max(1, w // denom), max(1, h // denom)))
                    d = np.asarray(im.convert("RGB"))
                    if d.shape[:2] == ((h * num + 7) // 8, (w * num + 7) // 8):
                        out[f"rgb_{i}_s{num}"] = d
            names.append(f"{w}x{h}_{kind}_{sub}_q{q}")
            i += 1
    buf = io.BytesIO()
    Image.fromarray(picture(40, 30, "waves")[..., 0]).save(buf, "JPEG", quality=80)
    data = buf.getvalue()
    out[f"jpg_{i}"] = np.frombuffer(data, np.uint8)
    out[f"rgb_{i}"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    names.append("40x30_gray_q80")
    out["names"] = np.array(names)
    out["decoder"] = np.array(f"Pillow {Image.__version__} / libjpeg-turbo {features.version('jpg')}")
    np.savez_compressed(os.path.join(HERE, "jpeg_cases.npz"), **out)
    print(len(names), "cases;", out["decoder"])


if __name__ == "__main__":
    main()
