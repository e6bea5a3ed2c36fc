#!/usr/bin/env python3
"""How many synchronisation rounds (and repeated count + write passes) a decode of BASELINE cfg4's files takes, and what a
decode costs, for a given cap of the in-workgroup fixpoint iteration (`ent_test_inner`; default: the library's).

    tools/exp_entropy_rounds.py [files=64] [caps=24,0] [first_file=0] [stream=0|1]     (cap 0 = the library's default)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from imageflow_amd import _native  # noqa: E402
from imageflow_amd.codecs import mozjpeg_decoder as D  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    caps = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "24,0").split(",")]
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    own_stream = len(sys.argv) > 4 and sys.argv[4] == "1"
    files = bench.cfg4_files(first, n)
    torch.zeros(1, device="cuda").item()
    st = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
    for cap in caps:
        _native.debug_set("ent_test_inner", str(cap) if cap else None)
        with torch.cuda.stream(st):
            ent = D.JpegEntropyBatch(files, device="cuda:0")            # (the cap is read when the batch is made)
            coef = ent.read_coefficients()
            torch.cuda.synchronize()
            rounds, ms = [], []
            for _ in range(10):
                t0 = time.perf_counter()
                ent.read_coefficients(coef)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3)
                rounds.append(ent.rounds)
        print(json.dumps({"files": n, "first_file": first, "own_stream": own_stream, "inner_cap": cap or "default", "rounds": rounds,
                          "decode_ms_median": round(float(np.median(ms)), 3), "decode_ms_min": round(min(ms), 3)}), flush=True)
    _native.debug_set("ent_test_inner", None)


if __name__ == "__main__":
    main()
