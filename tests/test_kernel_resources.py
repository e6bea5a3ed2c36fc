"""Static guard on the kernels that set the measured numbers (no GPU: hipcc cross-compiles and reports resource usage).
A change that pushes one of them into scratch memory or over a register / LDS budget shows up here, not first on a
rocprof trace: scratch loads share `vmcnt` with the source rows in flight, and the entropy kernels
are sized to the 160 KiB of LDS a workgroup can have."""
import os
import re
import subprocess

import pytest

from imageflow_amd import build as B


def resource_usage(src, extra=()):
    cmd = [B.HIPCC, "-x", "hip", "--offload-arch=gfx950"] + B.COMMON + list(extra) + ["--cuda-device-only", "-c", src, "-o", os.devnull,
                                                                                    "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    rows, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: \s*(Function Name|[A-Za-z ]+\[?[A-Za-z/]*\]?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = rows.setdefault(re.sub(r"\(.*", "", name).replace("ifhip::", ""), {})
        elif cur is not None:
            cur[k] = v
    return rows


def _int(r, key):
    return int(r[key])


def test_entropy_kernels_fit_one_workgroup_per_cu_without_scratch():
    rows = resource_usage(os.path.join(B.CSRC, "jpeg_entropy.hip"))
    for name, lanes in (("entropy_round_kernel", 1024), ("entropy_count_kernel", 1024), ("entropy_write_kernel", 512)):
        r = rows[name]
        assert _int(r, "ScratchSize [bytes/lane]") == 0, (name, r)
        assert _int(r, "LDS Size [bytes/block]") <= 160 * 1024, (name, r)
        assert _int(r, "VGPRs") <= 512 // (lanes // 256), (name, r)           # 512 registers per SIMD lane shared by lanes / 256 waves


@pytest.mark.parametrize("k", [4])
def test_headline_resample_kernel_keeps_its_shape(k):
    """fused_resample_kernel<4, false, true, true, 0> is the BASELINE cfg2 kernel: 1 024 lanes need <= 128 VGPRs; the fast
    horizontal forms (cfg3 / cfg4 / cfg1 resizes) must stay out of scratch."""
    rows = resource_usage(os.path.join(B.CSRC, B.FUSED), [f"-DIFHIP_FUSED_K={k}"])
    head = [r for n, r in rows.items() if n.startswith("void fused_resample_kernel<4, false, true, true, 0, false>")]      # (.., planar source)
    assert head and all(_int(r, "VGPRs") <= 128 for r in head), head
    fast = [r for n, r in rows.items() if re.search(r"fused_resample_kernel<4, (false|true), true, true, [234], (false|true)>", n)]
    assert len(fast) >= 3 and all(_int(r, "ScratchSize [bytes/lane]") == 0 and _int(r, "VGPRs") <= 128 for r in fast), fast


def test_jpeg_block_routines_keep_their_occupancy():
    """jpeg_idct_block_per_lane_kernel: the scaler forms (MODE 2) run two 512-lane workgroups per CU -- <= 80 KiB of LDS each
    and <= 128 registers (4 waves per SIMD); the plain 8x8 form four 256-lane workgroups of <= 40 KiB.  The routines are
    bound by instruction issue, so a lost wave per SIMD or a spill-heavy build shows up as time."""
    rows = resource_usage(os.path.join(B.CSRC, "jpeg_kernels.hip"))
    seen = 0
    for name, r in rows.items():
        m = re.match(r"void jpeg_idct_block_per_lane_kernel<(\d), (\d)>", name)
        if not m:
            continue
        seen += 1
        mode = int(m.group(1))
        assert _int(r, "VGPRs") <= 128, (name, r)
        assert _int(r, "LDS Size [bytes/block]") <= (80 if mode == 2 else 40) * 1024, (name, r)
        assert _int(r, "ScratchSize [bytes/lane]") <= 64, (name, r)            # a handful of values at the column -> row turn
    assert seen == 9
