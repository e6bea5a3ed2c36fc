#!/usr/bin/env python3
"""Experiment build in seconds: only csrc/jpeg_entropy.hip recompiled with extra -D flags, linked with the product's other
objects into lib/libimageflow_hip_<name>.so (IFHIP_LIB selects it).   usage: entropy_variant.py <name> [DEFINE[=v] ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd import build as B

name, defs = sys.argv[1], sys.argv[2:]
lib = os.path.join(B.HERE, "lib")
obj = os.path.join(lib, f"variant_{name}_jpeg_entropy.o")
subprocess.run([B.HIPCC, "-x", "hip", "--offload-arch=gfx950"] + B.COMMON + [f"-D{d}" for d in defs] +
               ["-c", os.path.join(B.CSRC, "jpeg_entropy.hip"), "-o", obj], check=True)
others = [o for _, o in B.compile_jobs() if not o.endswith("jpeg_entropy.hip.o")]
out = os.path.join(lib, f"libimageflow_hip_{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + others, check=True)
print(out)
