#!/usr/bin/env python3
"""Research tool (GPU): the banded kernel's column strips on wide up-scales -- what the planner picks and what other strip widths
cost (`banded_strip` hook; the band height follows from the planner's cost rule for that width).
    python tools/exp_banded_strips.py [in_w in_h out_w out_h filter frames alpha]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from imageflow_amd import _native  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402


def main():
    a = sys.argv[1:]
    iw, ih, ow, oh = (int(v) for v in a[:4]) if len(a) >= 4 else (960, 540, 1920, 1080)
    filt = getattr(Filter, a[4]) if len(a) > 4 else Filter.Ginseng
    n = int(a[5]) if len(a) > 5 else 256
    alpha = bool(int(a[6])) if len(a) > 6 else False
    dev = torch.device("cuda:0")
    st = (iw * 4 + 63) // 64 * 64
    src = torch.randint(0, 256, (n, ih * st), dtype=torch.uint8, device=dev)
    inp = Bitmap(src, iw, ih, st, alpha)
    can = Bitmap.create_u8(n, ow, oh, dev)
    info = ScaleAndRenderParams(0, 0, ow, oh, 0.0, filt)
    gb = n * (iw * ih * 4 + ow * oh * 4) / 1e9

    def timed(reps=5):
        scale_and_render(inp, can, info)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            scale_and_render(inp, can, info)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    _native.debug_set("trace_launch", "1")
    scale_and_render(inp, can, info)
    torch.cuda.synchronize()
    _native.debug_set("trace_launch", None)
    ms = timed()
    print(json.dumps({"shape": [iw, ih, ow, oh], "filter": filt.name, "frames": n, "alpha": alpha, "strip": "planner", "ms": round(ms, 4),
                      "of_8TBps": round(gb / ms / 8e3 * 1e3 / 1e3, 4)}), flush=True)
    ref = can.data.clone()
    for s in (32, 48, 64, 96, 112, 128, 192, 256, 384, 512):
        if s >= ow:
            continue
        _native.debug_set("banded_strip", str(s))
        _native.debug_set("trace_launch", "1")
        scale_and_render(inp, can, info)
        torch.cuda.synchronize()
        _native.debug_set("trace_launch", None)
        ms = timed()
        same = bool(torch.equal(ref, can.data))
        print(json.dumps({"strip": s, "ms": round(ms, 4), "of_8TBps": round(gb / ms / 8e3, 4), "same_bytes": same}), flush=True)
    _native.debug_set("banded_strip", None)


if __name__ == "__main__":
    main()
