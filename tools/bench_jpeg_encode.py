#!/usr/bin/env python3
"""Encode side end to end on the device: n BGRA frames in HBM -> forward pixel stage -> device entropy coder -> files in HBM
(tools/bench_jpeg_encode.py [n] [w h] [quality]).  Prints one JSON line: ms per batch of the pixel stage, of the coder (per
kernel with IFHIP_ENC_TIMING-free hipEvents around the whole call) and of the host writer on the same planes (download +
ifhip_jpeg_write_batch on every core) for scale; the files of the two coders are compared byte for byte."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageflow_amd.codecs import mozjpeg as M
from imageflow_amd.graphics.bitmaps import Bitmap


def smooth_frames(n, w, h, stride, dev):
    """photo-like content: gradients + mild noise (uniform noise would make q90 files ten times the size of real ones)"""
    g = torch.Generator(device=dev).manual_seed(7)
    y = torch.arange(h, device=dev).view(1, h, 1).float()
    x = torch.arange(w, device=dev).view(1, 1, w).float()
    k = torch.arange(n, device=dev).view(n, 1, 1).float()
    frames = torch.zeros((n, h, stride), dtype=torch.uint8, device=dev)
    px = frames[:, :, :4 * w].view(n, h, w, 4)
    noise = lambda: torch.randint(-12, 13, (n, h, w), device=dev, generator=g).float()
    px[..., 0] = (128 + 100 * torch.sin((x + 13 * k) / 97) + noise()).clamp(0, 255).to(torch.uint8)
    px[..., 1] = (128 + 100 * torch.cos((y + 7 * k) / 61) + noise()).clamp(0, 255).to(torch.uint8)
    px[..., 2] = ((x + y + 31 * k) / 5 % 256 * 0.8 + noise()).clamp(0, 255).to(torch.uint8)
    px[..., 3] = 255
    return frames.view(n, -1)


def main():
    a = [int(v) for v in sys.argv[1:]]
    n = a[0] if a else 32
    w, h = (a[1], a[2]) if len(a) >= 3 else (3840, 2160)
    q = a[3] if len(a) >= 4 else 90
    dev = "cuda:0"
    stride = (w * 4 + 63) // 64 * 64
    bm = Bitmap(smooth_frames(n, w, h, stride, dev), w, h, stride)
    res = {"frames": n, "w": w, "h": h, "quality": q, "device": torch.cuda.get_device_name(0)}
    for name, hs, vs in (("420", (2, 1, 1), (2, 1, 1)), ("444", (1, 1, 1), (1, 1, 1))):
        if os.environ.get("ENC_ONLY", name) != name:       # (counter passes: one sampling per run)
            continue
        fwd = M.JpegForwardStage(w, h, hs, vs, n, dev)
        qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(q)] * n).view(np.int16)).to(dev)
        coef = fwd.write_frames(bm, qt)
        coder = M.JpegEntropyStage(w, h, hs, vs, fwd.blocks_w, fwd.blocks_h, n, dev)
        # a pitch that holds these files (not the worst case: 2 x 208 bytes per block)
        files, lengths, status = coder.encode_device(coef, q)
        torch.cuda.synchronize()
        assert int(status.abs().sum()) == 0
        pitch = (int(lengths.max()) * 5 // 4 + 4095) // 4096 * 4096
        files = torch.empty((n, pitch), dtype=torch.uint8, device=dev)

        def timed(fn, reps=20):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        t_fwd = timed(lambda: fwd.write_frames(bm, qt, coef))
        t_enc = timed(lambda: coder.encode_device(coef, q, files=files))
        t_both = timed(lambda: (fwd.write_frames(bm, qt, coef), coder.encode_device(coef, q, files=files)))
        _, lengths, status = coder.encode_device(coef, q, files=files)
        torch.cuda.synchronize()
        lengths = lengths.cpu().numpy()
        t0 = time.perf_counter()
        host_planes = [c.cpu().numpy() for c in coef]
        t1 = time.perf_counter()
        host = M.write_jpeg_batch(host_planes, w, h, hs, vs, q)
        t2 = time.perf_counter()
        got = files.cpu().numpy()
        same = all(got[i, :int(lengths[i])].tobytes() == host[i] for i in range(n))
        coef_bytes = sum(c.numel() * 2 for c in coef)
        file_bytes = int(lengths.sum())
        res[name] = {
            "forward_ms": round(t_fwd, 4), "entropy_ms": round(t_enc, 4), "both_ms": round(t_both, 4),
            "MPps_entropy": round(n * w * h / 1e6 / (t_enc * 1e-3), 1), "MPps_both": round(n * w * h / 1e6 / (t_both * 1e-3), 1),
            "file_bytes": file_bytes, "bytes_per_px": round(file_bytes / (n * w * h), 4),
            # the coder reads every coefficient twice and writes the stream twice (words, then stuffed bytes)
            "entropy_TBps_algorithmic": round((2 * coef_bytes + 3 * file_bytes) / (t_enc * 1e-3) / 1e12, 3),
            "host_download_ms": round((t1 - t0) * 1e3, 2), "host_write_ms": round((t2 - t1) * 1e3, 2), "host_threads": os.cpu_count(),
            "files_equal_host_writer": bool(same),
        }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
