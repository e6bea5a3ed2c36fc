#!/usr/bin/env python3
"""Experiment driver (development aid): time the entropy stage of several library builds on the same 16 files.
usage: exp_entropy_variants.py gen | run   (run: IFHIP_LIB selects the build)"""
import io, json, os, pickle, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = "/tmp/ifhip_entropy_files.pkl"
if sys.argv[1] == "gen":
    from PIL import Image
    n, w, h = 16, 3840, 2160
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(1)
    files = []
    for k in range(n):
        base = np.stack([(x + 3 * k) * 255 // (w + 60), (y + 5 * k) * 255 // (h + 90), (x + y) * 255 // (w + h)], -1).astype(np.int16)
        tex = (40 * np.sin(x / (3.0 + k % 5)) * np.cos(y / (4.0 + k % 3)))[..., None] + rng.integers(-12, 13, size=(h, w, 3))
        buf = io.BytesIO()
        Image.fromarray(np.clip(base + tex, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
        files.append(buf.getvalue())
    pickle.dump(files, open(P, "wb"))
else:
    import torch
    from imageflow_amd.codecs import mozjpeg_decoder as D
    files = pickle.load(open(P, "rb"))
    torch.zeros(1, device="cuda").item()
    ent = D.JpegEntropyBatch(files)
    coef = ent.read_coefficients()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ent.read_coefficients(coef)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    chk = int(sum(int(c.to(torch.int64).sum().item()) for c in coef))
    print(json.dumps({"lib": os.path.basename(os.environ.get("IFHIP_LIB", "default")), "decode_ms": round(t * 1e3, 3), "rounds": ent.rounds,
                      "subs": ent.n_subsequences, "checksum": chk}))
