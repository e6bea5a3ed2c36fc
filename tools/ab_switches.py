#!/usr/bin/env python3
"""A/B of development-switch settings on resample workloads in ONE process on one box: inputs are made once per workload,
the settings are timed interleaved (hipEvents around back-to-back launches, as bench.py's kernel probe), and every
setting's canvas is check-summed -- settings that claim the same pixels must print the same sum.

    tools/ab_switches.py [--reps 3] [--launches 30] --workloads cfg3-l0,cfg3-l1 \
        --settings "base:" "one_wg:banded_wgs=1" ...

One JSON line per (workload, setting, repetition): {"workload", "setting", "rep", "kernel_ms", "frac", "checksum"} and a
closing summary line per workload with the median of each setting."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", required=True)
    ap.add_argument("--settings", nargs="+", required=True, help="name:key=value,key=value (empty list: name:)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--launches", type=int, default=30)
    ap.add_argument("--frames", type=int, default=None)
    args = ap.parse_args()
    import torch
    from imageflow_amd import _native
    from imageflow_amd.graphics.bitmaps import Bitmap, BitmapCompositing
    from imageflow_amd.graphics.scaling import ResamplePlan, ScaleAndRenderParams, scale_and_render, time_scale_and_render
    from imageflow_amd.graphics.weights import Filter
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    settings = []
    for s in args.settings:
        name, _, kv = s.partition(":")
        settings.append((name, [tuple(x.split("=", 1)) for x in kv.split(",") if x]))
    all_keys = sorted({k for _, kvs in settings for k, _ in kvs})
    for wl_name in args.workloads.split(","):
        wl = bench.WORKLOADS[wl_name]
        in_w, in_h, out_w, out_h = wl[:4]
        n = args.frames or wl[9]
        inp = bench.make_frames(torch, n, 0, 0, dev, "mixed", in_w, in_h)
        inp.alpha_meaningful = wl[6]
        if wl[6]:
            inp.data.view(n, in_h, -1)[:, :, 3::4] = torch.randint(0, 256, (n, in_h, inp.stride // 4), dtype=torch.uint8, device=dev)
        can = Bitmap.create_u8(n, out_w, out_h, dev, compose=BitmapCompositing[wl[7]], matte=wl[8])
        info = ScaleAndRenderParams(0, 0, out_w, out_h, wl[5], Filter[wl[4]])
        algo = n * (in_w * in_h * 4 + out_w * out_h * 4)
        times = {name: [] for name, _ in settings}
        for rep in range(args.reps):
            for name, kvs in settings:
                for k in all_keys:
                    _native.debug_set(k, None)
                for k, v in kvs:
                    _native.debug_set(k, v)
                try:
                    plan = ResamplePlan(in_w, in_h, out_w, out_h, info.interpolation_filter, wl[5])     # (not plan_for: plan-time switches must apply)
                    can.data.zero_()
                    scale_and_render(inp, can, info, plan=plan)
                    torch.cuda.synchronize()
                    checksum = int(can.data.to(torch.int64).sum().item())
                    ms = time_scale_and_render(inp, can, info, launches=args.launches, plan=plan)
                    times[name].append(ms)
                    print(json.dumps({"workload": wl_name, "setting": name, "rep": rep, "kernel_ms": round(ms, 4),
                                      "frac": round(algo / (ms * 1e-3) / bench.HBM_PEAK, 4), "checksum": checksum}), flush=True)
                except Exception as e:  # noqa: BLE001
                    print(json.dumps({"workload": wl_name, "setting": name, "rep": rep, "error": f"{type(e).__name__}: {e}"}), flush=True)
        for k in all_keys:
            _native.debug_set(k, None)
        print(json.dumps({"workload": wl_name, "median_ms": {k: round(statistics.median(v), 4) for k, v in times.items() if v},
                          "frac": {k: round(algo / (statistics.median(v) * 1e-3) / bench.HBM_PEAK, 4) for k, v in times.items() if v}}), flush=True)
        del inp, can
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
