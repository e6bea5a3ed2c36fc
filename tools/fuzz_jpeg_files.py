#!/usr/bin/env python3
"""Randomised parity sweep of the JPEG decode path on LARGE files (GPU; round 6).

tests/test_gpu_jpeg_random.py writes files up to 420 x 300; the parts of the entropy stage that only big scans reach -- many
workgroups per image, correction chains that cross workgroups, the dispatch order by scan size (wg_order) in batches of
unequal files, long runs of all-EOB blocks in flat regions (where a decoder that is bit-synchronous but a block out of phase
stays out of phase), 16-bit codes at q100 -- are covered there only by the two generated 4K pictures of
test_large_files_and_convergence.  This sweep writes files of 300 ... 4 200 pixels a side with Pillow (random content family per
file, quality 2 ... 100, 4:4:4 / 4:2:2 / 4:2:0 / gray, restart intervals, optimised tables), decodes batches of 1 ... 4 files of
one geometry through the entropy stage + pixel stage and compares every frame byte for byte with libjpeg-turbo's own decode
(Pillow); every third batch also compares the coefficient planes with the oracle's serial decoder.

    python tools/fuzz_jpeg_files.py [--seconds 400] [--seed 1] [--out gpurun_out/fuzz_jpeg.jsonl]"""
import argparse
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def picture(rng, w, h, gray):
    y, x = np.mgrid[0:h, 0:w]
    kind = int(rng.integers(0, 7))
    if kind == 0:        # noise
        a = rng.integers(0, 256, size=(h, w, 3))
    elif kind == 1:      # gradient (bench_codecs.rs:24-41)
        a = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1)
    elif kind == 2:      # waves + noise
        a = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 100 * np.cos(y / 5.0), 128 + 80 * np.sin((x + y) / 11.0)], -1) + rng.integers(-25, 26, size=(h, w, 3))
    elif kind == 3:      # flat: long runs of all-EOB blocks
        a = np.broadcast_to(rng.integers(0, 256, 3), (h, w, 3)).copy()
        if rng.random() < 0.5:
            a[h // 3:h // 3 + 9, :, :] = rng.integers(0, 256, size=(9, w, 3))[:a[h // 3:h // 3 + 9].shape[0]]
    elif kind == 4:      # half flat, half noise (split along a random axis)
        a = np.broadcast_to(rng.integers(0, 256, 3), (h, w, 3)).copy()
        if rng.random() < 0.5:
            a[:, w // 2:] = rng.integers(0, 256, size=(h, w - w // 2, 3))
        else:
            a[h // 2:] = rng.integers(0, 256, size=(h - h // 2, w, 3))
    elif kind == 5:      # checker of two colours, period not a multiple of 8
        p, q = int(rng.integers(3, 40)), int(rng.integers(3, 40))
        a = np.where(((x // p + y // q) % 2)[..., None] == 0, rng.integers(0, 256, 3), rng.integers(0, 256, 3))
    else:                # very smooth: a slow ramp in one channel only
        a = np.stack([np.full((h, w), int(rng.integers(0, 256))), (x // 16 + y // 16) % 256, np.full((h, w), int(rng.integers(0, 256)))], -1)
    a = np.clip(a, 0, 255).astype(np.uint8)
    return (a[..., 0] if gray else a), kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=400.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "fuzz_jpeg.jsonl"))
    args = ap.parse_args()
    import torch
    from PIL import Image
    from imageflow_amd.codecs import mozjpeg_decoder as D
    from oracle import oracle as O
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    done = bad = files_total = 0
    max_rounds = 0
    with open(args.out, "w") as f:
        while time.time() < t_end:
            w, h = int(rng.integers(300, 4201)), int(rng.integers(200, 2401))
            gray = rng.random() < 0.12
            q = int(rng.choice([int(rng.integers(2, 101)), 85, 90, 100, 75]))
            kw = dict(quality=q, optimize=bool(rng.integers(0, 2)))
            if not gray:
                kw["subsampling"] = ["4:4:4", "4:2:2", "4:2:0", "4:2:0"][int(rng.integers(0, 4))]
            r = int(rng.integers(0, 4))
            if r == 1:
                kw["restart_marker_rows"] = int(rng.integers(1, 6))
            elif r == 2:
                kw["restart_marker_blocks"] = int(rng.integers(1, 400))
            n = int(rng.integers(1, 5))
            files, refs, kinds = [], [], []
            for _ in range(n):
                pic, kind = picture(rng, w, h, gray)
                kinds.append(kind)
                buf = io.BytesIO()
                try:
                    Image.fromarray(pic).save(buf, "JPEG", **kw)
                except OSError:                  # libjpeg "Suspension not allowed here": optimize + restarts on some sizes
                    kw["optimize"] = False
                    buf = io.BytesIO()
                    Image.fromarray(pic).save(buf, "JPEG", **kw)
                files.append(buf.getvalue())
                refs.append(np.asarray(Image.open(io.BytesIO(files[-1])).convert("RGB")))
            rec = {"case": done, "size": [w, h], "gray": gray, "n": n, "kinds": kinds, "bytes": [len(x) for x in files],
                   "save": {k: (v if not isinstance(v, (np.integer,)) else int(v)) for k, v in kw.items()}}
            try:
                ent = D.JpegEntropyBatch(files, "cuda:0")
                # guard regions behind the coefficient planes and the decoded frames: no kernel may store there (the write pass did,
                # on unsettled exit states, until round 6's position bound)
                coef, guards = [], []
                for c in range(3):                                   # (a grayscale batch carries two one-block dummies, as read_coefficients makes them)
                    shape = (ent.n, max(ent.blocks_h[c], 1), max(ent.blocks_w[c], 1), 64)
                    count = int(np.prod(shape))
                    buf = torch.full((count + (1 << 18),), 0x5A5A, dtype=torch.int16, device="cuda:0")
                    guards.append(buf[count:])
                    coef.append(buf[:count].view(shape))
                ent.read_coefficients(coef)
                rec["rounds"], rec["subsequences"] = int(ent.rounds), int(ent.n_subsequences)
                max_rounds = max(max_rounds, rec["rounds"])
                stage = D.JpegPixelStage(ent.width, ent.height, ent.ncomp, ent.h_samp, ent.v_samp, ent.n, "cuda:0")
                qt = torch.from_numpy(ent.qt[:, :ent.ncomp].copy().view(np.int16)).to("cuda:0")
                from imageflow_amd.graphics.bitmaps import Bitmap, get_stride
                fst = get_stride(stage.out_w)
                fbuf = torch.full((ent.n + 1, stage.out_h * fst), 0xA5, dtype=torch.uint8, device="cuda:0")
                frames = stage.read_frames(coef, qt, Bitmap(fbuf[:ent.n], stage.out_w, stage.out_h, fst)).to_numpy()
                errs = []
                if not all(bool((g == 0x5A5A).all()) for g in guards):
                    errs.append("a store landed behind a coefficient plane")
                if not bool((fbuf[ent.n] == 0xA5).all()):
                    errs.append("a store landed behind the last decoded frame")
                for k in range(n):
                    px = frames[k][:, :4 * w].reshape(h, w, 4)
                    if not np.array_equal(px[..., [2, 1, 0]], refs[k]):
                        errs.append(f"frame {k}: {int((px[..., [2, 1, 0]] != refs[k]).sum())} bytes differ from libjpeg-turbo")
                    if not np.all(px[..., 3] == 255):
                        errs.append(f"frame {k}: alpha")
                if done % 3 == 0:
                    for k, data in enumerate(files):
                        j = O.jpeg_read_coefficients(data)
                        for c in range(ent.ncomp):
                            if not np.array_equal(coef[c][k].cpu().numpy(), j["coef"][c]):
                                errs.append(f"frame {k} component {c}: coefficients differ from the serial decoder")
                    rec["coefficients_checked"] = True
                rec["ok"] = not errs
                if errs:
                    rec["error"] = errs[:6]
                    bad += 1
            except Exception as e:  # noqa: BLE001
                rec["ok"] = False
                rec["error"] = f"{type(e).__name__}: {str(e)[:300]}"
                bad += 1
            f.write(json.dumps(rec) + "\n")
            f.flush()
            done += 1
            files_total += n
        summary = {"summary": True, "seed": args.seed, "batches": done, "files": files_total, "mismatching_batches": bad, "max_rounds": max_rounds}
        f.write(json.dumps(summary) + "\n")
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
