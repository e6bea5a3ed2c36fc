"""The host-side investigation tools of round 5 stay runnable (no GPU): the job harness of tools/bench_abi_jobs.cpp builds with
the compiler of this image and refuses a bad command line; tools/sample_stacks.py symbolises and ranks a samples file of the
harness's format (module+offset frames, leaf first) against the library built here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "imageflow_amd", "lib", "libimageflow_hip.so")


def test_job_harness_builds_and_prints_its_usage(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import bench_abi_jobs as B
    finally:
        sys.path.pop(0)
    exe = B.build_harness(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=30)
    assert r.returncode == 2 and "usage:" in r.stderr
    for kind in ("cfg1", "cfg4", "cfg4h"):                       # the three job kinds are JSON the library's reader accepts
        assert "framewise" in B.JOBS[kind]


def test_sample_stacks_ranks_a_samples_file(tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    nm = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True, check=True).stdout
    off = {ln.split()[2]: int(ln.split()[0], 16) for ln in nm.splitlines() if len(ln.split()) == 3}
    a, b = off["ifhip_cache_stats"], off["imageflow_context_send_json"]
    lines = [f"{LIB}+{a + 4:#x};{LIB}+{b + 8:#x}"] * 3 + [f"{LIB}+{b + 8:#x}", "?+0x0;" + f"{LIB}+{b + 8:#x}"]
    f = tmp_path / "s.samples"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sample_stacks.py"), str(f), "5"], capture_output=True, text=True,
                         check=True, timeout=120).stdout
    assert "5 samples" in out
    leaf = out.split("-- leaf")[1].split("--")[0]
    assert "60.0 %" in leaf and "ifhip_cache_stats" in leaf
    ours = out.split("-- first frame in libimageflow_hip.so")[1].split("--")[0]
    assert "imageflow_context_send_json" in ours and "ifhip_cache_stats" in ours
