#!/usr/bin/env python3
"""Throughput of the whole-bitmap primitives (csrc/bitmap_ops.hip) and the flatten kernel on n 3840x2160 BGRA frames
resident in HBM: ms per batch and GB/s of bytes moved (read + written)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd.flow.nodes import color as CN  # noqa: E402
from imageflow_amd.graphics import bitmap_ops as G  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.blend import apply_matte  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = "cuda:0"
    w, h = 3840, 2160
    a = Bitmap.create_u8(n, w, h, dev, alpha_meaningful=True)
    a.data.copy_(torch.randint(0, 256, a.data.shape, dtype=torch.uint8, device=dev))
    b = Bitmap.create_u8(n, w, h, dev, alpha_meaningful=True)
    t = Bitmap.create_u8(n, h, w, dev, alpha_meaningful=True)
    px = n * w * h
    res = {}

    def rec(name, seconds, bytes_moved):
        res[name] = {"ms": round(seconds * 1e3, 3), "GBps": round(bytes_moved / seconds / 1e9, 1), "GPps": round(px / seconds / 1e9, 1)}

    rec("copy_rect (full frame, aligned)", timed(lambda: G.copy_rectangle(a, b, 0, 0, 0, 0, w, h)), 8 * px)
    rec("copy_rect (crop 3001x2001 at 13,7 -> 5,3)", timed(lambda: G.copy_rectangle(a, b, 13, 7, 5, 3, 3001, 2001)), 8 * n * 3001 * 2001)
    rec("fill_rect (full frame)", timed(lambda: G.fill_rectangle(b, 0xFF336699, 0, 0, w, h)), 4 * px)
    rec("flip_vertical", timed(lambda: G.flow_bitmap_bgra_flip_vertical_safe(b)), 8 * px)
    rec("flip_horizontal", timed(lambda: G.flow_bitmap_bgra_flip_horizontal_safe(b)), 8 * px)
    rec("transpose", timed(lambda: G.bitmap_window_transpose(a, t)), 8 * px)
    rec("color_matrix (saturation)", timed(lambda: G.window_bgra32_apply_color_matrix(b, CN.saturation(0.3))), 8 * px)

    def matte_once():                                   # the flatten makes every pixel opaque: restore the input each time
        b.data.copy_(a.data)
        b.alpha_meaningful = True
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        apply_matte(b, 0xFFFFFFFF)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    matte_once()
    rec("apply_matte (random alpha, white)", min(matte_once() for _ in range(5)), 8 * px)
    b.alpha_meaningful = True
    rec("apply_matte (already opaque: read only)", timed(lambda: (setattr(b, "alpha_meaningful", True), apply_matte(b, 0xFFFFFFFF))), 4 * px)
    print(json.dumps({"frames": n, "size": [w, h], **res}, indent=1))


if __name__ == "__main__":
    main()
