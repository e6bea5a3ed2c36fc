#!/usr/bin/env python3
"""Experiment builds in seconds (NOT part of the product): recompile only api.cpp and ONE ring size of resample_fused.hip with
extra -D flags and link them with the product's other objects -> lib/libimageflow_hip_<name>.so.
usage: quick_variant.py <name> <K> DEFINE[=V] ...   (valid only for workloads whose ring size is K)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd import build as B
name, K = sys.argv[1], int(sys.argv[2])
defs = [x for d in sys.argv[3:] for x in (d.split() if d.startswith("-") else [f"-D{d}"])]    # "-mllvm -flag=v" passes through
lib = os.path.join(B.HERE, "lib")
tmp = os.path.join(lib, f"variant_{name}")
os.makedirs(tmp, exist_ok=True)
base = [B.HIPCC, "-x", "hip", "--offload-arch=gfx950"] + B.COMMON + defs
jobs = [(base + ["-c", os.path.join(B.CSRC, "api.cpp"), "-o", os.path.join(tmp, "api.cpp.o")], None),
        (base + B.FUSED_FLAGS + [f"-DIFHIP_FUSED_K={K}", "-c", os.path.join(B.CSRC, B.FUSED), "-o", os.path.join(tmp, f"resample_fused_k{K}.o")], None)]
B.run_jobs(jobs)
objs = []
for f in sorted(os.listdir(lib)):
    if not f.endswith(".o"): continue
    objs.append(os.path.join(tmp, f) if f in ("api.cpp.o", f"resample_fused_k{K}.o") else os.path.join(lib, f))
out = os.path.join(lib, f"libimageflow_hip_{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
