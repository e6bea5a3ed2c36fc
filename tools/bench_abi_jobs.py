#!/usr/bin/env python3
"""Throughput through the boundary the repository exports (libimageflow C ABI + v1/execute JSON): file in, file out.

    python tools/bench_abi_jobs.py [--threads 1,8,64] [--seconds 4] [--jobs cfg1,cfg4]

  cfg1  BASELINE config 1: `command_string width=200` on a 3840x2160 4:2:0 q85 JPEG (GPU Huffman decode -> 2/8 IDCT with the
        spatial sRGB luma scaler -> Robidoux to 200x113; the shim's command_string writes the raw BGRA container)
  cfg4  BASELINE config 4: decode -> constrain within 800 -> encode libjpeg_turbo q85 (a real JPEG out; full-size decode)
  cfg4h the same with the reference's querystring decoder hints (4/8 IDCT + spatial luma scaler; decode + resample fused)

For each job kind and thread count: tools/bench_abi_jobs.cpp (g++, std::thread, one imageflow_context per job) -> jobs/s,
source megapixels/s, the per-node wall / GPU microseconds of the jobs' `performance` blocks.  Beside them, on ONE host core:
libjpeg-turbo (Pillow: draft-mode decode + resize + save, what a CPU service does for the same request) and the oracle's
chain (tests' checker; decode + resize only).  One JSON document on stdout."""
import argparse
import io
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

JOBS = {
    "cfg1": {"framewise": {"steps": [{"command_string": {"kind": "ir4", "value": "width=200", "decode": 0, "encode": 1}}]}},
    "cfg4": {"framewise": {"steps": [{"decode": {"io_id": 0}}, {"constrain": {"mode": "within", "w": 800}},
                                     {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": {"quality": 85}}}}]}},
    # the same with the decoder hints the reference's querystring path sends for an 800 px target (ir4/mod.rs:167-198:
    # 2.1 x the target, spatial luma scaling in linear light): 4/8 IDCT, decode + resample as one device call
    "cfg4h": {"framewise": {"steps": [{"decode": {"io_id": 0, "commands": [{"jpeg_downscale_hints": {
        "width": 1680, "height": 945, "scale_luma_spatially": True, "gamma_correct_for_srgb_during_spatial_luma_scaling": True}}]}},
                                      {"constrain": {"mode": "within", "w": 800}},
                                      {"encode": {"io_id": 1, "preset": {"libjpeg_turbo": {"quality": 85}}}}]}},
}


def make_file(w=3840, h=2160, k=0):
    from PIL import Image
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(1)
    base = np.stack([(x + 3 * k) * 255 // (w + 60), (y + 5 * k) * 255 // (h + 90), (x + y) * 255 // (w + h)], -1).astype(np.int16)
    tex = (40 * np.sin(x / 3.0) * np.cos(y / 4.0))[..., None] + rng.integers(-12, 13, size=(h, w, 3))
    buf = io.BytesIO()
    Image.fromarray(np.clip(base + tex, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
    return buf.getvalue()


def build_harness(tmp):
    exe = os.path.join(tmp, "bench_abi_jobs")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "bench_abi_jobs.cpp"), "-o", exe, "-ldl", "-lpthread"], check=True)
    return exe


def cpu_one_core(data, kind, seconds=3.0):
    from PIL import Image
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        im = Image.open(io.BytesIO(data))
        if kind == "cfg1":
            im.draft("RGB", (480, 270))                              # libjpeg's scaled IDCT, as the reference's decoder hints ask
            im = im.convert("RGB").resize((200, 113), Image.BICUBIC)
            im.tobytes()
        else:
            if kind == "cfg4h":
                im.draft("RGB", (1920, 1080))
            im = im.convert("RGB").resize((800, 450), Image.BICUBIC)
            im.save(io.BytesIO(), "JPEG", quality=85)
        n += 1
    return n / (time.perf_counter() - t0)


def oracle_one_core(data, kind):
    from oracle import oracle as O
    from tests import util as U
    t0 = time.perf_counter()
    j = O.jpeg_read_coefficients(data)
    if kind == "cfg1":
        small, (sw, sh, tw, th) = O.jpeg_idct_color_scaled(j, 2, 2), (960, 540, 200, 113)
    elif kind == "cfg4":
        small, (sw, sh, tw, th) = O.jpeg_idct_color(j), (3840, 2160, 800, 450)
    else:
        small, (sw, sh, tw, th) = O.jpeg_idct_color_scaled(j, 4, 2), (1920, 1080, 800, 450)
    can = np.zeros((th, U.stride_for(tw)), np.uint8)
    O.scale_and_render(np.ascontiguousarray(small), sw, sh, can, tw, th, 0, 0, tw, th, filter_id=2)
    return 1.0 / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,16,64")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--jobs", default="cfg1,cfg4,cfg4h")
    ap.add_argument("--lib", default=os.environ.get("IFHIP_LIB") or os.path.join(ROOT, "imageflow_amd", "lib", "libimageflow_hip.so"))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--spread", action="store_true",
                    help="ifhip_shim_spread_contexts(1): contexts take the node's GPUs round-robin; every run reports jobs per device "
                         "and fails if a device stays idle (on a one-GPU box: every context on ordinal 0)")
    ap.add_argument("--hw-queues", default="", help="GPU_MAX_HW_QUEUES for the harness process (the HIP runtime's own switch: how many "
                                                      "hardware queues its streams are spread over; default 4)")
    args = ap.parse_args()
    data = make_file()
    out = {"file": {"w": 3840, "h": 2160, "bytes": len(data), "what": "4:2:0 q85 baseline, gradient + texture + noise (Pillow)"},
           "host_cores": os.cpu_count(), "lib": os.path.basename(args.lib), "GPU_MAX_HW_QUEUES": args.hw_queues or "default", "runs": []}
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_harness(tmp)
        fpath = os.path.join(tmp, "in.jpg")
        open(fpath, "wb").write(data)
        for kind in args.jobs.split(","):
            jpath = os.path.join(tmp, kind + ".json")
            open(jpath, "w").write(json.dumps(JOBS[kind]))
            for t in [int(v) for v in args.threads.split(",")]:
                env = dict(os.environ)
                if args.hw_queues:
                    env["GPU_MAX_HW_QUEUES"] = args.hw_queues
                r = subprocess.run([exe, args.lib, fpath, jpath, str(t), str(args.seconds), "3"] + (["--spread"] if args.spread else []),
                                   capture_output=True, text=True, timeout=600, env=env)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                rec = json.loads(line[-1]) if line else {"error": (r.stderr or r.stdout)[-500:]}
                rec["job"] = kind
                if "jobs_per_s" in rec:
                    rec["source_MPps"] = round(rec["jobs_per_s"] * 3840 * 2160 / 1e6, 1)
                out["runs"].append(rec)
            if not args.no_cpu:
                out.setdefault("cpu_one_core", {})[kind] = {
                    "libjpeg_turbo_pillow_jobs_per_s": round(cpu_one_core(data, kind), 2),
                    "oracle_chain_jobs_per_s": round(oracle_one_core(data, kind), 3),
                    "what": "one host core: Pillow (libjpeg-turbo draft-mode decode + bicubic resize" + (" + JPEG q85 save" if kind == "cfg4" else "") +
                            "); the oracle's decode + resize chain (tests' checker, scalar C)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
