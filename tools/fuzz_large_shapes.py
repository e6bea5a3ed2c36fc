#!/usr/bin/env python3
"""Randomised parity sweep of the resample / render boundary in the LARGE-shape regime (GPU; round 6).

tests/test_gpu_random_shapes.py draws 480 configurations with sources up to 900 pixels wide; the launch heuristics that only
wide or many frames reach -- column strips (sources wider than a workgroup's 4 096 / 2 048 columns), bands of output rows,
several frames per workgroup, the fast horizontal pass with 2 / 3 / 4 groups and its two-column form, the 1 024- / 512-lane
shapes by ring size, the banded kernel at large bands -- are covered there only by hand-picked BASELINE shapes.  This sweep
draws from that regime, compares every configuration bit for bit with the CPU oracle (BGRA8 canvas incl. padding and the
f32 working buffer, `tests.test_gpu_resample.run_case`) and keeps going after a mismatch.

    python tools/fuzz_large_shapes.py [--seconds 480] [--seed 1] [--out gpurun_out/fuzz_large.jsonl]

One JSON line per configuration (geometry, the kernel the plan chose, ok / the mismatch) and a summary line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def draw(rng, Filter, WorkingFloatspace, BitmapCompositing):
    common = [Filter.Robidoux, Filter.Robidoux, Filter.Lanczos, Filter.Ginseng, Filter.Hermite, Filter.CatmullRom, Filter.Mitchell,
              Filter.Triangle, Filter.Box, Filter.RobidouxSharp, Filter.LanczosSharp, Filter.Lanczos2]
    kind = int(rng.integers(0, 6))
    n = int(rng.integers(1, 4))
    if kind == 0:      # wide thumbnails: strips x bands, big rings
        in_w, in_h = int(rng.integers(2000, 9001)), int(rng.integers(60, 700))
        out_w, out_h = int(rng.integers(20, 520)), int(rng.integers(4, max(5, in_h // 3)))
    elif kind == 1:    # moderate ratios on wide rows (fast horizontal pass, its two-column form)
        in_w, in_h = int(rng.integers(900, 5200)), int(rng.integers(40, 420))
        out_w = max(1, int(in_w * rng.uniform(0.2, 0.98)))
        out_h = max(1, int(in_h * rng.uniform(0.2, 0.98)))
    elif kind == 2:    # many narrow frames: several frames per workgroup
        in_w, in_h = int(rng.integers(60, 1100)), int(rng.integers(30, 300))
        out_w = max(1, int(in_w * rng.uniform(0.05, 0.95)))
        out_h = max(1, int(in_h * rng.uniform(0.05, 0.95)))
        n = int(rng.integers(3, 24))
    elif kind == 3:    # up-scales the fused kernel keeps (<= 2.67x) and the banded kernel beyond, on wide rows
        in_w, in_h = int(rng.integers(300, 2200)), int(rng.integers(20, 200))
        out_w = int(in_w * rng.uniform(1.0, 3.6))
        out_h = int(in_h * rng.uniform(1.0, 3.6))
    elif kind == 4:    # one axis up, one down, wide
        in_w, in_h = int(rng.integers(1500, 8000)), int(rng.integers(8, 120))
        out_w, out_h = int(in_w * rng.uniform(0.03, 0.6)), int(in_h * rng.uniform(1.0, 3.0))
    else:              # exact BASELINE ratios at other sizes (19.2 x 10.8, 2.4, 1.333, 2, 3)
        r = [(19.2, 10.8), (2.4, 2.4), (4 / 3, 4 / 3), (2.0, 2.0), (3.0, 3.0), (19.2, 19.115)][int(rng.integers(0, 6))]
        out_w, out_h = int(rng.integers(50, 1700)), int(rng.integers(10, 200))
        in_w, in_h = int(round(out_w * r[0])), int(round(out_h * r[1]))
    while in_w * in_h * n > 14_000_000:                     # the oracle is a scalar port: keep a case under a second or two
        if n > 1:
            n -= 1
        else:
            in_h = max(8, in_h // 2)
            out_h = max(1, out_h // 2)
    out_w, out_h = max(1, out_w), max(1, out_h)
    x, y = (int(rng.integers(0, 40)), int(rng.integers(0, 9))) if rng.random() < 0.4 else (0, 0)
    ew, eh = (int(rng.integers(0, 30)), int(rng.integers(0, 5))) if rng.random() < 0.4 else (0, 0)
    filt = common[int(rng.integers(0, len(common)))] if rng.random() < 0.8 else list(Filter)[int(rng.integers(0, len(list(Filter))))]
    return dict(in_w=in_w, in_h=in_h, out_w=out_w, out_h=out_h, n=n, filt=filt,
                sharpen=float(rng.choice([0.0, 0.0, 0.0, 15.0, 60.0])),
                space=WorkingFloatspace(int(rng.random() < 0.8)), compose=BitmapCompositing(int(rng.integers(0, 3))),
                matte=int(rng.choice([0xFFFFFFFF, 0x80FF2010, 0x00000000])), alpha=bool(rng.integers(0, 2)),
                x=x, y=y, cw=out_w + x + ew, ch=out_h + y + eh), kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=480.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_large.jsonl"))
    args = ap.parse_args()

    import torch
    from imageflow_amd.graphics.bitmaps import BitmapCompositing
    from imageflow_amd.graphics.color import WorkingFloatspace
    from imageflow_amd.graphics.weights import Filter
    from oracle import oracle as O
    from tests.test_gpu_resample import run_case

    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    done = bad = rejected = 0
    kinds, kernels = {}, {}
    with open(args.out, "w") as f:
        while time.time() < t_end and done < args.max_cases:
            c, kind = draw(rng, Filter, WorkingFloatspace, BitmapCompositing)
            iw, ih, ow, oh = c.pop("in_w"), c.pop("in_h"), c.pop("out_w"), c.pop("out_h")
            probe_in = np.zeros((ih, O.stride_for_width(iw)), np.uint8)
            probe_cv = np.zeros((c["ch"], O.stride_for_width(c["cw"])), np.uint8)
            rc, _ = O.scale_and_render(probe_in, iw, ih, probe_cv, c["cw"], c["ch"], c["x"], c["y"], ow, oh, filter_id=int(c["filt"]), sharpen=c["sharpen"])
            if rc != 0:                                     # the reference's populate_weights errors on this one
                rejected += 1
                continue
            rec = {"case": done, "kind": kind, "in": [iw, ih], "out": [ow, oh], "n": c["n"], "filter": c["filt"].name, "sharpen": c["sharpen"],
                   "space": c["space"].name, "compose": c["compose"].name, "alpha": c["alpha"], "rect": [c["x"], c["y"], c["cw"], c["ch"]]}
            try:
                plan = run_case(iw, ih, ow, oh, seed=done, **c)
                rec["kernel_kind"] = int(plan.kernel_kind(c["alpha"]))
                rec["ok"] = True
            except AssertionError as e:
                rec["ok"] = False
                rec["error"] = str(e)[:300]
                bad += 1
            except Exception as e:  # noqa: BLE001
                rec["ok"] = False
                rec["error"] = f"{type(e).__name__}: {str(e)[:300]}"
                bad += 1
            kinds[kind] = kinds.get(kind, 0) + 1
            kk = str(rec.get("kernel_kind"))
            kernels[kk] = kernels.get(kk, 0) + 1
            f.write(json.dumps(rec) + "\n")
            f.flush()
            done += 1
        summary = {"summary": True, "seed": args.seed, "cases": done, "mismatches": bad, "rejected_by_the_reference_rule": rejected,
                   "by_kind": kinds, "by_kernel_kind (0 fused, 1 the banded kernel where a band fits the LDS, else the generic pair)": kernels}
        f.write(json.dumps(summary) + "\n")
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
