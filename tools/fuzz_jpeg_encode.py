#!/usr/bin/env python3
"""Randomised parity sweep of the JPEG ENCODE side on large frames (GPU; round 6): BGRA frames in HBM -> forward pixel stage
(colour conversion, down-sampling, islow FDCT, quantiser: csrc/jpeg_forward.hip) -> device entropy coder (csrc/jpeg_encode.hip)
-> complete baseline files, compared BYTE FOR BYTE with the file libjpeg-turbo itself writes for the same pixels and quality
(Pillow, optimize=False).  The seeded suite stops at 1 600 x 900 (tests/test_gpu_jpeg_device_coder.py) and 500 x 300
(test_gpu_jpeg_random.py's forward sweep); this one draws 16 ... 4 000 x 16 ... 2 400, 1 ... 3 frames per call, seven content
families (noise, gradients, flat regions, checkers ...), quality 1 ... 100, 4:2:0 / 4:2:2 / 4:4:4.

    python tools/fuzz_jpeg_encode.py [--seconds 300] [--seed 1] [--out gpurun_out/fuzz_jpeg_encode.jsonl]"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402

SAMPLINGS = {"4:2:0": ([2, 1, 1], [2, 1, 1]), "4:2:2": ([2, 1, 1], [1, 1, 1]), "4:4:4": ([1, 1, 1], [1, 1, 1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_jpeg_encode.jsonl"))
    args = ap.parse_args()
    import torch
    from PIL import Image
    from fuzz_jpeg_files import picture
    from imageflow_amd.codecs import mozjpeg as M
    from imageflow_amd.graphics.bitmaps import Bitmap
    from oracle import oracle as O
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    done = bad = files_total = 0
    with open(args.out, "w") as f:
        while time.time() < t_end:
            if rng.random() < 0.25:
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 200))
            else:
                w, h = int(rng.integers(16, 4001)), int(rng.integers(16, 2401))
            n = int(rng.integers(1, 4))
            q = int(rng.choice([int(rng.integers(1, 101)), 90, 85, 100]))
            sampling = ["4:2:0", "4:2:0", "4:4:4", "4:2:2"][int(rng.integers(0, 4))]
            hs, vs = SAMPLINGS[sampling]
            stride = O.stride_for_width(w) + 64 * int(rng.integers(0, 2))
            frames = np.zeros((n, h, stride), np.uint8)
            refs, kinds = [], []
            for k in range(n):
                rgb, kind = picture(rng, w, h, False)
                kinds.append(kind)
                px = frames[k, :, :4 * w].reshape(h, w, 4)
                px[..., 0], px[..., 1], px[..., 2], px[..., 3] = rgb[..., 2], rgb[..., 1], rgb[..., 0], 255
                buf = io.BytesIO()
                Image.fromarray(rgb).save(buf, "JPEG", quality=q, optimize=False, subsampling=sampling)
                refs.append(buf.getvalue())
            rec = {"case": done, "size": [w, h], "n": n, "quality": q, "sampling": sampling, "kinds": kinds, "bytes": [len(r) for r in refs]}
            try:
                stage = M.JpegForwardStage(w, h, hs, vs, n)
                qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(q)] * n).view(np.int16)).cuda()
                coef = stage.write_frames(Bitmap.from_numpy(frames, w, h, stride, "cuda:0"), qt)
                coder = M.JpegEntropyStage(w, h, hs, vs, stage.blocks_w, stage.blocks_h, n)
                files, status = coder.encode(coef, q)
                errs = []
                for k in range(n):
                    if status[k] != 0:
                        errs.append(f"frame {k}: status {status[k]}")
                    elif files[k] != refs[k]:
                        errs.append(f"frame {k}: {len(files[k])} bytes against libjpeg-turbo's {len(refs[k])}"
                                    + ("" if len(files[k]) != len(refs[k]) else f", first difference at {next(i for i in range(len(refs[k])) if files[k][i] != refs[k][i])}"))
                rec["ok"] = not errs
                if errs:
                    rec["error"] = errs
                    bad += 1
                del stage, coder, coef
            except Exception as e:  # noqa: BLE001
                rec["ok"] = False
                rec["error"] = f"{type(e).__name__}: {str(e)[:300]}"
                bad += 1
            f.write(json.dumps(rec) + "\n")
            f.flush()
            done += 1
            files_total += n
        summary = {"summary": True, "seed": args.seed, "calls": done, "files": files_total, "mismatching_calls": bad}
        f.write(json.dumps(summary) + "\n")
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
