#!/usr/bin/env python3
"""Instructions of one kernel in the gfx950 ISA hipcc makes of a source file, by class and by phase.  Phases are cut by
`asm volatile("; ifhip-phase: <name>")` comment lines (no instruction) which THIS TOOL puts into a temporary copy of the
source in front of the anchor lines listed in MARKERS -- the product's source carries none, and the marked build differs
from the product's by the scheduling freedom a volatile asm takes away (jpeg luma routine: 2 304 against 2 352
instructions, 48 against 40 bytes of scratch per lane).

    tools/isa_count.py imageflow_amd/csrc/jpeg_kernels.hip 'jpeg_idct_block_per_lane_kernelILi2ELi4E' [more name fragments]

Static counts of the straight-line code: a loop body counts once (the block-per-lane IDCT routines have no loops), and
both sides of a wave-uniform choice are listed (24- / 32-bit column pass, sRGB / plain scaler): a file runs one of each."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imageflow_amd import build as B  # noqa: E402


# file name -> [(anchor: the first source line of the phase, exactly as in the file, phase name)]
MARKERS = {"jpeg_kernels.hip": [
    ("    const uint32_t wv = t >> 6, ln = t & 63u;\n    const uint32_t wave_block0 = wg * bpl_threads(MODE) + wv * 64u;", "coefficients: global -> LDS -> own block"),
    ("    int32_t dmax = 0, dmin = 0;", "de-quantisation (64 x 16-bit multiply, range test)"),
    ("        if (small) bpl_column_pass<true>(ws);\n        else bpl_column_pass<false>(ws);\n        locate();",
     None),                                              # rewritten below: a marker inside each side of the choice
    ("            int32_t (&lin)[8][8] = ws;", "islow row pass + range limit / linear light table (64 LDS reads)"),
    ("            // row pass (CONST_BITS + PASS1_BITS + 3: always in 24-bit range)", "islow row pass + range-limit table + store"),
    ("            if (srgb) bpl_scale_block<N, true>(lin, plane, by, bx, a.g.pw[c], l2s_lds);\n            else bpl_scale_block<N, false>(lin, plane, by, bx, a.g.pw[c], l2s_lds);", None),
]}
REWRITES = {"jpeg_kernels.hip": [
    ("        if (small) bpl_column_pass<true>(ws);\n        else bpl_column_pass<false>(ws);\n        locate();",
     '        if (small) { PH("islow column pass, 24-bit multiplies (what every 8-bit file runs)"); bpl_column_pass<true>(ws); }\n'
     '        else { PH("islow column pass, 32-bit multiplies (not run by 8-bit files)"); bpl_column_pass<false>(ws); }\n'
     '        PH("block position (one division)");\n        locate();'),
    ("            if (srgb) bpl_scale_block<N, true>(lin, plane, by, bx, a.g.pw[c], l2s_lds);\n            else bpl_scale_block<N, false>(lin, plane, by, bx, a.g.pw[c], l2s_lds);",
     '            if (srgb) { PH("flow_scale_spatial_srgb_NxN: rows, columns, linear -> sRGB table, store"); bpl_scale_block<N, true>(lin, plane, by, bx, a.g.pw[c], l2s_lds); }\n'
     '            else { PH("flow_scale_spatial_NxN (no gamma): rows, columns, store"); bpl_scale_block<N, false>(lin, plane, by, bx, a.g.pw[c], l2s_lds); }'),
]}


def marked_copy(src, td):
    name = os.path.basename(src)
    text = open(src).read()
    if name not in MARKERS:
        return src
    for anchor, phase in MARKERS[name]:
        assert text.count(anchor) == 1, f"anchor not found exactly once: {anchor[:60]!r}"
        if phase is not None:
            text = text.replace(anchor, f'    PH("{phase}");\n' + anchor)
    for old, new in REWRITES.get(name, []):
        text = text.replace(old, new)
    text = '#define PH(name) asm volatile("; ifhip-phase: " name)\n' + text
    out = os.path.join(td, name)
    open(out, "w").write(text)
    return out


def klass(m):
    if m.startswith(("v_mad_i32_i24", "v_mul_i32_i24", "v_mad_u32_u24", "v_mul_u32_u24")):
        return "valu 24-bit multiply(-add)"
    if m.startswith(("v_mul_lo", "v_mad_u64", "v_mad_i64", "v_mul_hi")):
        return "valu 32-bit multiply (quarter rate)"
    if m.startswith("v_"):
        return "valu other"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_"):
        return "salu / control"
    return "other"


def main():
    src, frags = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        msrc = marked_copy(src, td)
        cmd = [B.HIPCC, "-x", "hip", "--offload-arch=gfx950"] + B.COMMON + ["-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(os.path.abspath(src)),
                                                                        "-S", "--cuda-device-only", "-o", out, msrc]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    for frag in frags:
        m = re.search(r"\n(_Z\w*" + re.escape(frag) + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S)
        if not m:
            print(f"{frag}: not found")
            continue
        phase, per = "(before the first marker)", collections.OrderedDict()
        for line in m.group(2).split("\n"):
            t = line.strip()
            pm = re.match(r";\s*ifhip-phase:\s*(.*)", t)
            if pm:
                phase = pm.group(1).strip()
                continue
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            per.setdefault(phase, collections.Counter())[klass(t.split()[0])] += 1
        total = collections.Counter()
        print(f"== {m.group(1)}")
        for ph, c in per.items():
            total.update(c)
            print(f"  {ph:44s} {sum(c.values()):5d}   " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))
        print(f"  {'TOTAL':44s} {sum(total.values()):5d}   " + ", ".join(f"{k} {v}" for k, v in sorted(total.items())))


if __name__ == "__main__":
    main()
