#!/usr/bin/env python3
"""What an overlapped gather costs the resample kernel, measured on ONE GPU: tools/probes/cu_hog_probe.hip stands in for
RCCL's kernel (same LDS per workgroup, one workgroup per channel: it cannot share a CU with a resample workgroup) and is
launched on a second stream behind every batch, exactly where `bench.py --gather every` issues batch k's gather -- beside
the kernel of batch k + 1.  For each (frames per GPU, hog workgroups, hog duration) the batch time with the launch geometry
planned for all 256 CUs and for 256 - reserve (ifhip_set_cu_budget), and the same without any hog.

    tools/exp_gather_overlap.py [--frames 128,256] [--channels 8] [--micros 150,300,600] [--steps 100]"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imageflow_amd import _native  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, plan_for, scale_and_render  # noqa: E402
from imageflow_amd.graphics.weights import Filter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="128,256")
    ap.add_argument("--channels", default="8")
    ap.add_argument("--micros", default="150,300,600")
    ap.add_argument("--reserve", default="0,8,16")
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    so = os.path.join(ROOT, "tools", "probes", "libcu_hog.so")
    if not os.path.exists(so):                              # (built in-tree, git-ignored like every binary)
        import subprocess
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                        os.path.join(ROOT, "tools", "probes", "cu_hog_probe.hip")], check=True)
    hog = ctypes.CDLL(so)
    hog.cu_hog_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sink = torch.zeros(16, dtype=torch.int32, device=dev)
    in_w, in_h, out_w, out_h = 3840, 2160, 200, 200
    side = torch.cuda.Stream()
    main_stream = torch.cuda.current_stream()
    for n in [int(v) for v in args.frames.split(",")]:
        inp = bench.make_frames(torch, n, 0, 0, dev, "mixed", in_w, in_h)
        canv = [Bitmap.create_u8(n, out_w, out_h, dev) for _ in range(2)]
        info = ScaleAndRenderParams(0, 0, out_w, out_h, 0.0, Filter.Robidoux)
        plan = plan_for(in_w, in_h, out_w, out_h, Filter.Robidoux, 0.0, dev)

        def run(steps, wgs, micros, reserve):
            _native.set_cu_budget(256 - reserve if reserve else 0)
            done = [None, None]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                if done[i & 1] is not None:
                    main_stream.wait_event(done[i & 1])          # the canvas of step i - 2 has been "gathered"
                scale_and_render(inp, canv[i & 1], info, plan=plan)
                if wgs:
                    ev = torch.cuda.Event()
                    ev.record(main_stream)
                    side.wait_event(ev)
                    rc = hog.cu_hog_launch(ctypes.c_void_p(side.cuda_stream), wgs, micros, ctypes.c_void_p(sink.data_ptr()))
                    assert rc == 0, rc
                    done[i & 1] = torch.cuda.Event()
                    done[i & 1].record(side)
            torch.cuda.synchronize()
            _native.set_cu_budget(0)
            return (time.perf_counter() - t0) / steps * 1e3

        run(10, 0, 0, 0)
        for reserve in [int(v) for v in args.reserve.split(",")]:
            base = min(run(args.steps, 0, 0, reserve) for _ in range(3))
            print(json.dumps({"frames": n, "hog_workgroups": 0, "reserve_cus": reserve, "ms_per_batch": round(base, 4)}), flush=True)
        for wgs in [int(v) for v in args.channels.split(",")]:
            for micros in [int(v) for v in args.micros.split(",")]:
                for reserve in [int(v) for v in args.reserve.split(",")]:
                    t = min(run(args.steps, wgs, micros, reserve) for _ in range(3))
                    print(json.dumps({"frames": n, "hog_workgroups": wgs, "hog_micros": micros, "reserve_cus": reserve, "ms_per_batch": round(t, 4)}), flush=True)
        del inp, canv


if __name__ == "__main__":
    main()
