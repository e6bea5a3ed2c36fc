#!/usr/bin/env python3
"""Extract the DATA of imageflow's generated 8x8 -> NxN block scalers (weights, divisors, 12-bit LUTs) from
c_components/lib/codecs_jpeg_idct_fast.c (read-only reference tree; run in the build container only).

Why data and not a generator: the committed C file was produced by an older weight generator -- no filter of today's
catalogue reproduces its integer weights through tests/integration/variation.rs (we tried all 31; see
tests/test_block_scalers.py::test_no_catalogue_filter_regenerates_the_committed_weights) -- so the only faithful
description of what the reference executes is the numbers themselves.  Outputs:
  tests/golden/block_scaler_tables.npz          weights[8][7][8] int8, log2div[8][7], lut_s2l[256] u16, lut_l2s[4096] u8
  imageflow_amd/csrc/block_scaler_weights.inc   the same weights/divisors as a C++ initialiser (28 rows of 8 small ints)
"""
import os
import re

import numpy as np

REF = os.environ.get("IMAGEFLOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(REF, "c_components/lib/codecs_jpeg_idct_fast.c")).read()

weights = np.zeros((8, 7, 8), np.int8)
log2div = np.zeros((8, 7), np.uint8)
seen = {}
pat = r"FLOW_EXPORT void flow_scale_spatial_(srgb_)?(\d)x\2\(uint8_t input\[64\], uint8_t \*\* output_rows, uint32_t output_col\)\n\{(.*?)\n\}\n"
for m in re.finditer(pat, src, re.S):
    srgb, n, body = bool(m.group(1)), int(m.group(2)), m.group(3)
    rows, cur = {}, None
    for line in body.splitlines():
        mm = re.search(r"// Begin work for output row (\d+)", line)
        if mm:
            cur = int(mm.group(1)); rows[cur] = [0] * 8; continue
        mm = re.search(r"temp\[i \+ \d+\] = (-?\d+) \* (?:input|linearized)\[i \+ (\d+)\]", line)
        if mm and cur is not None:
            rows[cur][int(mm.group(2)) // 8] = int(mm.group(1))
    mat = np.array([rows[r] for r in range(n)], np.int64)
    div = mat.sum(1)
    assert all(d > 0 and (d & (d - 1)) == 0 for d in div), (n, div)
    # rounding constants in the file must be div_r * div_c / 2 in row-major (r, c) order
    halves = [int(x) for x in re.findall(r"sum = (\d+);", body) if int(x) != 0]
    assert halves == [int(div[r] * div[c] // 2) for r in range(n) for c in range(n)], (n, srgb)
    if n in seen:
        assert np.array_equal(seen[n], mat)          # the srgb and plain variants share one weight matrix
    seen[n] = mat
    weights[n, :n] = mat
    log2div[n, :n] = np.log2(div).astype(np.uint8)


def c_array(name):
    body = src[src.index(name):]
    body = body[body.index("{") + 1: body.index("};")]
    return [int(v) for v in re.findall(r"\d+", body)]


s2l = np.array(c_array("lut_srgb_to_linear[256]"), np.uint16)
l2s = np.array(c_array("lut_linear_to_srgb[4096]"), np.uint8)
assert s2l.shape == (256,) and l2s.shape == (4096,)
np.savez_compressed(os.path.join(HERE, "block_scaler_tables.npz"), weights=weights, log2div=log2div, lut_s2l=s2l, lut_l2s=l2s)

inc = os.path.join(HERE, "..", "..", "imageflow_amd", "csrc", "block_scaler_weights.inc")
with open(inc, "w") as f:
    f.write("// block_scaler_weights.inc -- DATA: integer weights and divisors of imageflow's 8x8 -> NxN block scalers, as\n"
            "// executed by c_components/lib/codecs_jpeg_idct_fast.c (flow_scale_spatial[_srgb]_NxN share one matrix per N).\n"
            "// Extracted by tests/golden/make_block_scaler_tables.py; row r of size N = weights of source rows/cols 0..7 for\n"
            "// output r, summing to 1 << log2_div.  { N, r, log2_div, w0..w7 }\n")
    for n in range(1, 8):
        for r in range(n):
            f.write("{%d, %d, %d, {%s}},\n" % (n, r, log2div[n, r], ", ".join(str(int(v)) for v in weights[n, r])))
print("ok", {n: seen[n].sum(1).tolist() for n in sorted(seen)})

# The two 12-bit LUTs.  Regenerating them with variation.rs:107-151 and today's libm powf reproduces lut_srgb_to_linear
# exactly but misses 34 of the 4096 lut_linear_to_srgb entries by one (rounding boundaries; the committed file came from
# another powf), so both ship as data, the big one as the 255 thresholds of the (monotone) table.
assert np.all(np.diff(l2s.astype(int)) >= 0)
thr = np.searchsorted(l2s, np.arange(1, 256), side="left").astype(np.uint16)     # first index with value >= v
assert np.array_equal(np.searchsorted(thr, np.arange(4096), side="right").astype(np.uint8), l2s)
lut_inc = os.path.join(HERE, "..", "..", "imageflow_amd", "csrc", "block_scaler_luts.inc")
with open(lut_inc, "w") as f:
    f.write("// block_scaler_luts.inc -- DATA: the 12-bit LUTs of c_components/lib/codecs_jpeg_idct_fast.c.\n"
            "// kScalerS2L[256] = lut_srgb_to_linear; kScalerL2SThr[255]: kScalerL2SThr[v-1] = first index i with\n"
            "// lut_linear_to_srgb[i] >= v, i.e. lut_linear_to_srgb[i] = #{v : thr[v-1] <= i}.\n"
            "// Extracted by tests/golden/make_block_scaler_tables.py.\n")
    f.write("static const uint16_t kScalerS2L[256] = {\n")
    for i in range(0, 256, 16):
        f.write("    " + ", ".join(str(int(v)) for v in s2l[i:i + 16]) + ",\n")
    f.write("};\nstatic const uint16_t kScalerL2SThr[255] = {\n")
    for i in range(0, 255, 16):
        f.write("    " + ", ".join(str(int(v)) for v in thr[i:i + 16]) + ",\n")
    f.write("};\n")
print("luts ok")
