#!/usr/bin/env python3
"""Research tool (GPU): time scale_and_render on a grid of REALISTIC shapes (not only BASELINE's) and print the fraction of 8 TB/s
each reaches on its algorithmic bytes -- finds shapes that fall onto a slow path (round 6: HD up-scales ran at 0.01 - 0.02).
    python tools/scan_shapes.py [--alpha]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402

SHAPES = [
    (3840, 2160, 1920, 1080), (1920, 1080, 1280, 720), (4000, 3000, 1024, 768), (4000, 3000, 2000, 1500), (6000, 4000, 1200, 800),
    (1001, 667, 500, 333), (1280, 720, 640, 360), (1920, 1080, 1900, 1069), (1920, 1080, 1920, 1080), (800, 600, 799, 599),
    (640, 480, 1280, 960), (1920, 1080, 3840, 2160), (500, 333, 2000, 1332), (64, 64, 1024, 1024), (256, 256, 300, 300),
    (1920, 1080, 2560, 1440), (3840, 2160, 4096, 2304), (1200, 1800, 400, 600), (3000, 4000, 150, 200), (8000, 6000, 800, 600),
    (12000, 9000, 400, 300), (400, 300, 100, 75), (150, 150, 48, 48), (5000, 300, 500, 30), (300, 5000, 30, 500),
]
FILTERS = [Filter.Robidoux, Filter.Lanczos, Filter.Ginseng, Filter.Box, Filter.Triangle]


def main():
    alpha = "--alpha" in sys.argv
    dev = torch.device("cuda:0")
    for (iw, ih, ow, oh) in SHAPES:
        for filt in FILTERS:
            per = iw * ih * 4 + ow * oh * 4
            n = max(1, min(256, int(1.5e9 // per)))
            st = (iw * 4 + 63) // 64 * 64
            src = torch.randint(0, 256, (n, ih * st), dtype=torch.uint8, device=dev)
            inp = Bitmap(src, iw, ih, st, alpha)
            can = Bitmap.create_u8(n, ow, oh, dev)
            info = ScaleAndRenderParams(0, 0, ow, oh, 0.0, filt)
            try:
                plan = scale_and_render(inp, can, info)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 3
                e0.record()
                for _ in range(reps):
                    scale_and_render(inp, can, info)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                frac = n * per / (ms * 1e-3) / 8e12
                rec = {"shape": [iw, ih, ow, oh], "filter": filt.name, "frames": n, "ms": round(ms, 3), "of_8TBps": round(frac, 4),
                       "kernel_kind": int(plan.kernel_kind(alpha))}
            except Exception as e:  # noqa: BLE001
                rec = {"shape": [iw, ih, ow, oh], "filter": filt.name, "error": str(e)[:120]}
            print(json.dumps(rec), flush=True)
            del src, inp, can


if __name__ == "__main__":
    main()
