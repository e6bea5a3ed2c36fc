#!/usr/bin/env python3
"""Research tool (GPU): shapes the fused kernel takes (<= 8 live rows) timed on the fused kernel and on the banded kernel
(force_kernel 2), with and without alpha -- where should auto mode prefer the banded one?"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402

SHAPES = [(1920, 1080, 3840, 2160), (1920, 1080, 2560, 1440), (640, 480, 1280, 960), (200, 200, 400, 400), (1920, 1080, 1900, 1069),
          (1920, 1080, 1920, 1080), (800, 600, 1000, 750), (256, 256, 300, 300), (1280, 720, 1920, 1080), (1920, 1080, 1280, 720),
          (1600, 900, 1200, 675), (500, 333, 900, 600)]


def main():
    dev = torch.device("cuda:0")
    global SHAPES
    if len(sys.argv) > 1:                                    # shapes as iw,ih,ow,oh arguments
        SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (iw, ih, ow, oh) in SHAPES:
        for filt in (Filter.Robidoux, Filter.Box, Filter.Triangle, Filter.Hermite, Filter.Ginseng):
            for alpha in (False, True):
                per = iw * ih * 4 + ow * oh * 4
                n = max(1, min(256, int(1.5e9 // per)))
                st = (iw * 4 + 63) // 64 * 64
                src = torch.randint(0, 256, (n, ih * st), dtype=torch.uint8, device=dev)
                inp = Bitmap(src, iw, ih, st, alpha)
                can = Bitmap.create_u8(n, ow, oh, dev)
                info = ScaleAndRenderParams(0, 0, ow, oh, 0.0, filt)
                rec = {"shape": [iw, ih, ow, oh], "filter": filt.name, "alpha": alpha, "frames": n}
                for name, force in (("auto", -1), ("banded", 2)):
                    try:
                        plan = scale_and_render(inp, can, info, force_kernel=force)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(3):
                            scale_and_render(inp, can, info, force_kernel=force)
                        e1.record()
                        torch.cuda.synchronize()
                        rec[name + "_ms"] = round(e0.elapsed_time(e1) / 3, 3)
                        if force == -1:
                            rec["auto_kind"] = int(plan.kernel_kind(alpha))
                    except Exception as e:  # noqa: BLE001
                        rec[name + "_ms"] = None
                if rec.get("auto_ms") and rec.get("banded_ms"):
                    rec["banded_over_auto"] = round(rec["banded_ms"] / rec["auto_ms"], 2)
                print(json.dumps(rec), flush=True)
                del src, inp, can


if __name__ == "__main__":
    main()
