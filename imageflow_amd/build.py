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


FUSED = "resample_fused.hip"          # compiled once per ring size K (-DIFHIP_FUSED_K=K), in parallel
# its step loop is unrolled by hand-over depth (up to 16 rows): past clang's default size limit for `#pragma unroll`
FUSED_FLAGS = ["-mllvm", "-pragma-unroll-threshold=131072"]
FUSED_KS = range(1, 9)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")) and f != FUSED)


def compile_jobs():
    """(command, object) pairs for every translation unit."""
    jobs = []
    defs = []
    for src in sources():
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        jobs.append(([HIPCC, "-x", "hip", "--offload-arch=gfx950"] + COMMON + defs + ["-c", src, "-o", obj], obj))
    for k in FUSED_KS:
        obj = os.path.join(HERE, "lib", f"resample_fused_k{k}.o")
        jobs.append(([HIPCC, "-x", "hip", "--offload-arch=gfx950"] + COMMON + FUSED_FLAGS + defs + [f"-DIFHIP_FUSED_K={k}", "-c",
                     os.path.join(CSRC, FUSED), "-o", obj], obj))
    return jobs


def run_jobs(jobs, verbose=False):
    from concurrent.futures import ThreadPoolExecutor

    def one(job):
        cmd, obj = job
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        return list(ex.map(one, jobs))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "imageflow_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def headers_mtime():
    """newest header any translation unit may include (csrc/*.hpp|*.inc|*.h, include/*.h) and this file (the flags)"""
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".h"))]
    deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")] + [os.path.abspath(__file__)]
    return max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    jobs = compile_jobs()
    if not force:                                   # objects newer than their source and every header are kept
        h = headers_mtime()
        todo = [(cmd, obj) for cmd, obj in jobs
                if not os.path.exists(obj) or os.path.getmtime(obj) <= max(h, os.path.getmtime(cmd[cmd.index("-c") + 1]))]
        run_jobs(todo, verbose)
        objs = [obj for _, obj in jobs]
    else:
        objs = run_jobs(jobs, verbose)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
