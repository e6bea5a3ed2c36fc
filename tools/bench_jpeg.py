#!/usr/bin/env python3
"""Measurement of the JPEG pixel stage (BASELINE config 4 shape): n x 3840x2160 4:2:0 coefficient planes resident in HBM
-> BGRA (full size, or reduced with spatial sRGB luma) -> resample to 800x450.  Not the headline bench; numbers go to
DESIGN.md.  Synthetic coefficients (sparse, in-range), per-image quantisation tables."""
import json
import sys
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd.codecs.mozjpeg_decoder import JpegPixelStage  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    # --chain N: only the cfg4 chain (4/8 decode with the spatial sRGB luma scaler -> 800x450), N calls and nothing else --
    # the command the counter passes profile; one_call = 0 runs the two-call form (BGRA bitmap in HBM) instead
    chain_calls = int(sys.argv[sys.argv.index("--chain") + 1]) if "--chain" in sys.argv else 0
    two_call = "--two-call" in sys.argv
    dev = "cuda:0"
    w, h = 3840, 2160
    res = {}
    for scale_num, spatial in ((8, False), (4, True), (1, True)):
        st = JpegPixelStage(w, h, 3, (2, 1, 1), (2, 1, 1), n, dev, scale_num=scale_num, luma_spatial=spatial, luma_srgb=spatial)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        coef = []
        for c in range(3):
            shape = (n, st.blocks_h[c], st.blocks_w[c], 64)
            t = torch.randint(-30, 31, shape, dtype=torch.int16, device=dev, generator=g)
            mask = torch.rand(shape, device=dev, generator=g) < 0.15
            mask[..., 0] = True
            coef.append(t * mask)
        qt = torch.randint(1, 40, (n, 3, 64), dtype=torch.int16, device=dev, generator=g)
        out = Bitmap.create_u8(n, st.out_w, st.out_h, dev)
        small = Bitmap.create_u8(n, 800, 450, dev)
        info = ScaleAndRenderParams(0, 0, 800, 450)
        if chain_calls:
            if scale_num != 4:
                continue
            def chain():
                if two_call:
                    st.read_frames(coef, qt, out)
                    scale_and_render(out, small, info)
                else:
                    st.read_frames_into(coef, qt, small, info)
            chain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(chain_calls):
                chain()
            torch.cuda.synchronize()
            print(json.dumps({"chain_calls": chain_calls, "two_call": two_call, "frames": n,
                              "ms_per_call": round((time.perf_counter() - t0) / chain_calls * 1e3, 4)}))
            return
        for _ in range(3):
            st.read_frames(coef, qt, out)
            if (st.out_w, st.out_h) != (800, 450):
                scale_and_render(out, small, info)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            st.read_frames(coef, qt, out)
        torch.cuda.synchronize()
        t_dec = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            st.read_frames(coef, qt, out)
            scale_and_render(out, small, info)
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / reps
        # the same chain as ONE call (ifhip_jpeg_decode_resample_batch_device): at reduced scales the resampler reads the
        # component planes, no decoded BGRA frame goes through HBM
        fused = False
        t_one = None
        if (st.out_w, st.out_h) != (800, 450):
            for _ in range(3):
                fused = st.read_frames_into(coef, qt, small, info)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                st.read_frames_into(coef, qt, small, info)
            torch.cuda.synchronize()
            t_one = (time.perf_counter() - t0) / reps
        coef_bytes = sum(int(np.prod(c.shape)) * 2 for c in coef)
        res[f"scale_{scale_num}_8{'_spatial_srgb' if spatial else ''}"] = {
            "one_call_ms": None if t_one is None else round(t_one * 1e3, 3), "one_call_fused": fused,
            "one_call_algorithmic_GBps": None if t_one is None else round((coef_bytes + n * 800 * 450 * 4) / t_one / 1e9, 1),
            "frames": n, "decoded_size": [st.out_w, st.out_h],
            "decode_ms": round(t_dec * 1e3, 3), "decode_source_MPps": round(n * w * h / 1e6 / t_dec, 1),
            "decode_algorithmic_GBps": round((coef_bytes + n * st.out_w * st.out_h * 4) / t_dec / 1e9, 1),
            "decode_plus_resize_800_ms": round(t_all * 1e3, 3), "chain_source_MPps": round(n * w * h / 1e6 / t_all, 1)}
    # encode side: n x 4K BGRA frames in HBM -> quantised coefficient planes (4:2:0 and 4:4:4)
    from imageflow_amd.codecs.mozjpeg import JpegForwardStage, quant_tables_for_quality
    frames = Bitmap.create_u8(n, w, h, dev)
    frames.data.copy_(torch.randint(0, 256, frames.data.shape, dtype=torch.uint8, device=dev))
    qt = torch.from_numpy(np.stack([quant_tables_for_quality(90)] * n).view(np.int16)).to(dev)
    for name, hs, vs in (("420", (2, 1, 1), (2, 1, 1)), ("444", (1, 1, 1), (1, 1, 1))):
        st = JpegForwardStage(w, h, hs, vs, n, dev)
        coef = st.write_frames(frames, qt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            st.write_frames(frames, qt, coef)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        coef_bytes = sum(int(np.prod(c.shape)) * 2 for c in coef)
        res[f"forward_{name}"] = {"frames": n, "ms": round(t * 1e3, 3), "MPps": round(n * w * h / 1e6 / t, 1),
                                  "algorithmic_GBps": round((n * w * h * 4 + coef_bytes) / t / 1e9, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
