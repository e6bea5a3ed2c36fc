#!/usr/bin/env python3
"""BASELINE config 3 as a whole job on one GPU: the export_4_sizes pyramid (imageflow_tool/src/self_test.rs:185-198)
src 3840x2160 -> 1600x900 -> {1200x675 -> 400x225, 800x450}, n frames device resident, all four outputs kept.
Reports ms per batch, source megapixels/s and algorithmic GB/s (58 737 600 B per image)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, plan_for, scale_and_render  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    w, h = 3840, 2160
    src = Bitmap.create_u8(n, w, h, dev)
    src.data.copy_(torch.randint(0, 256, src.data.shape, dtype=torch.uint8, device=dev))
    sizes = {"1600": (1600, 900), "1200": (1200, 675), "800": (800, 450), "400": (400, 225)}
    out = {k: Bitmap.create_u8(n, *v, dev) for k, v in sizes.items()}
    edges = [(src, "1600"), ("1600", "1200"), ("1600", "800"), ("1200", "400")]
    plans = {}
    for a, b in edges:
        s = src if a is src else out[a]
        plans[(id(a) if a is src else a, b)] = plan_for(s.w, s.h, *sizes[b], Filter.Robidoux, 0.0, dev)

    def job():
        for a, b in edges:
            s = src if a is src else out[a]
            scale_and_render(s, out[b], ScaleAndRenderParams(0, 0, *sizes[b]), plan=plans[(id(a) if a is src else a, b)])

    for _ in range(3):
        job()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    algo = n * 58_737_600
    print(json.dumps({"images": n, "ms_per_batch": round(t * 1e3, 3), "images_per_s": round(n / t, 1),
                      "source_MPps": round(n * w * h / 1e6 / t, 1), "algorithmic_GBps": round(algo / t / 1e9, 1),
                      "frac_of_8TBps": round(algo / t / 8e12, 3)}))


if __name__ == "__main__":
    main()
