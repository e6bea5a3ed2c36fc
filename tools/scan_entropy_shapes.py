#!/usr/bin/env python3
"""Research tool (GPU): the entropy stage over file kinds (not only BASELINE config 4's 4K 4:2:0 q85 files): sizes x samplings x
restart intervals x optimised tables x content, one batch of files of one geometry per case; source gigapixels per second of
read_coefficients -- finds kinds that fall onto a slow path."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from imageflow_amd.codecs import mozjpeg_decoder as D  # noqa: E402


def make(w, h, k, content, **kw):
    y, x = np.mgrid[0:h, 0:w]
    if content == "gradient":
        a = np.stack([(x + y + k) & 255, (y + k) & 255, (x + k) & 255], -1).astype(np.uint8)
    elif content == "noise":
        a = np.random.default_rng(k).integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:                                    # photo-like: smooth field + mild noise
        rng = np.random.default_rng(k)
        base = 128 + 90 * np.sin(x / 97.0 + k) * np.cos(y / 61.0) + 20 * np.sin(x / 7.0) * np.sin(y / 5.0)
        a = np.clip(base[..., None] + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
    im = Image.fromarray(a, "RGB")
    if kw.pop("gray", False):
        im = im.convert("L")
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue()


def main():
    dev = "cuda:0"
    for (w, h, n) in ((640, 480, 256), (1920, 1080, 64), (3840, 2160, 32), (200, 150, 512)):
        for content in ("gradient", "photo", "noise"):
            for sname, sub, gray in (("420", "4:2:0", False), ("444", "4:4:4", False), ("gray", "4:2:0", True)):
                for extra_name, extra in (("plain", {}), ("optimize", {"optimize": True}), ("rst_row", {"restart_marker_rows": 1}),
                                          ("rst_4_blocks", {"restart_marker_blocks": 4}), ("q100", {"quality": 100})):
                    kw = {"quality": 85, "subsampling": sub}
                    kw.update(extra)
                    if gray:
                        kw.pop("subsampling")
                        kw["gray"] = True
                    try:
                        distinct = [make(w, h, k, content, **dict(kw)) for k in range(4)]
                        files = [distinct[i % 4] for i in range(n)]
                        ent = D.JpegEntropyBatch(files, device=dev)
                        coef = ent.read_coefficients()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        reps = 4
                        e0.record()
                        for _ in range(reps):
                            ent.read_coefficients(coef)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / reps
                        rec = {"size": [w, h], "files": n, "content": content, "sampling": sname, "kind": extra_name,
                               "bytes_per_file": int(np.mean([len(f) for f in distinct])), "ms": round(ms, 4),
                               "source_GPps": round(n * w * h / ms / 1e6, 1), "compressed_GBps": round(sum(len(f) for f in files) / ms / 1e6, 2),
                               "files_per_s": int(n / ms * 1e3)}
                    except Exception as e:  # noqa: BLE001
                        rec = {"size": [w, h], "content": content, "sampling": sname, "kind": extra_name, "error": str(e)[:160]}
                    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
