"""Build libimageflow_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m imageflow_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (every FMA in the kernels is an explicit fmaf; nothing else may
fuse), IEEE f32 division (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), denormals preserved
(hipcc default on gfx9+).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libimageflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "imageflow_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Experiment builds (tools/): same sources with extra -D flags into lib/libimageflow_hip_<name>.so."""
    out = os.path.join(HERE, "lib", f"libimageflow_hip_{name}.so")
    srcs = []
    for src in sources():
        srcs += (["-x", "hip"] if src.endswith(".cpp") else ["-x", "hip"]) + [src]
    cmd = [HIPCC, "--offload-arch=gfx950"] + COMMON + [f"-D{d}" for d in defines] + ["-shared"] + srcs + ["-o", out]
    subprocess.run(cmd, check=True)
    return out


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    for src in sources():
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        cmd = [HIPCC, "--offload-arch=gfx950"] + COMMON + ["-c", src, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")          # host-only translation units still include hip_runtime.h
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
