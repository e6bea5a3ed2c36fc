#!/usr/bin/env python3
"""Kernel times of several bench.py workloads in ONE process (one library build: IFHIP_LIB selects it), for A/B runs of
variant builds on one box.  usage: IFHIP_LIB=<lib> tools/ab_variants.py <tag> workload ...   -> one JSON line per workload
with ms_per_step (timed loop), kernel_ms (hipEvents) and a checksum of the output bitmaps (equal across exact builds)."""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag, wls = sys.argv[1], sys.argv[2:]
for rep in range(int(os.environ.get('AB_REPS', '2'))):
    for wl in wls:
        sys.argv = ["bench.py", "--workload", wl, "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-strong-field", "--gather", "none"]
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                bench.main()
            j = json.loads(buf.getvalue().strip().splitlines()[-1])
            print(json.dumps({"tag": tag, "workload": wl, "rep": rep, "ms_per_step": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"],
                              "frac": j["roofline"]["frac"]}), flush=True)
        except BaseException as e:  # noqa: BLE001
            print(json.dumps({"tag": tag, "workload": wl, "rep": rep, "error": f"{type(e).__name__}: {e}"}), flush=True)
