#!/usr/bin/env python3
"""BASELINE config 3 as a whole job on one GPU: the export_4_sizes pyramid (imageflow_tool/src/self_test.rs:185-198)
src 3840x2160 -> 1600x900 -> {1200x675 -> 400x225, 800x450}, n frames device resident, all four outputs kept.
Reports ms per batch, source megapixels/s and algorithmic GB/s (58 737 600 B per image).
With --encode the job ends where the reference's does: every output through the classic JPEG encoder at quality 90
(self_test.rs:190-197 `libjpeg_turbo`), pixel stage and entropy coder on the device, 4 n files left in HBM."""
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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    encode = "--encode" in sys.argv
    n = int(args[0]) if args else 128
    dev = torch.device("cuda:0")
    w, h = 3840, 2160
    src = Bitmap.create_u8(n, w, h, dev)
    if encode:                                    # photo-like content: file sizes of uniform noise say nothing about real ones
        from bench_jpeg_encode import smooth_frames
        src.data.copy_(smooth_frames(n, w, h, src.stride, dev))
    else:
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

    resize_only = job
    if encode:
        import numpy as np
        from imageflow_amd.codecs import mozjpeg as M
        hs, vs = M.sampling_factors((2, 2), (2, 2))
        qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(90)] * n).view(np.int16)).to(dev)
        enc = {}
        for k, (ow, oh) in sizes.items():
            fwd = M.JpegForwardStage(ow, oh, hs, vs, n, dev)
            coef = fwd.write_frames(out[k], qt)
            coder = M.JpegEntropyStage(ow, oh, hs, vs, fwd.blocks_w, fwd.blocks_h, n, dev)
            pitch = (2 * ow * oh + 4095) // 4096 * 4096 + 4096                   # two bytes per pixel: several times a q90 file
            enc[k] = (fwd, coef, coder, torch.empty((n, pitch), dtype=torch.uint8, device=dev))

        def job():                                                              # noqa: F811
            resize_only()
            res = {}
            for k, (fwd, coef, coder, files) in enc.items():
                fwd.write_frames(out[k], qt, coef)                              # (outputs have no meaningful alpha: no matte pass)
                res[k] = coder.encode_device(coef, 90, files=files)
            return res

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
    extra = {}
    if encode:
        res = job()
        torch.cuda.synchronize()
        lengths = {k: v[1].cpu().numpy() for k, v in res.items()}
        assert all(int(v[2].abs().sum()) == 0 for v in res.values()) and all((l > 0).all() for l in lengths.values())
        t1 = time.perf_counter()
        for _ in range(reps):
            resize_only()
        torch.cuda.synchronize()
        extra = {"encode": "libjpeg_turbo q90 4:2:0, pixel stage + entropy coder on the device", "files": 4 * n,
                 "file_bytes_per_image": int(sum(int(l.sum()) for l in lengths.values()) // n),
                 "resize_only_ms": round((time.perf_counter() - t1) / reps * 1e3, 3)}
    print(json.dumps({**extra, "images": n, "ms_per_batch": round(t * 1e3, 3), "images_per_s": round(n / t, 1),
                      "source_MPps": round(n * w * h / 1e6 / t, 1), "algorithmic_GBps": round(algo / t / 1e9, 1),
                      "frac_of_8TBps": round(algo / t / 8e12, 3)}))


if __name__ == "__main__":
    main()
