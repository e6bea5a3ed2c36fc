#!/usr/bin/env python3
"""Regenerate tests/golden/* from the read-only reference tree (run in the build container only).

The reference's golden vectors for the hot path are data files, not code:
  imageflow_core/tests/integration/weights.txt         (30 filters x 22 size pairs)
  imageflow_core/tests/integration/weights_params.txt  (12 filters x 14 parameter variants x 10 size pairs)
  imageflow_core/src/graphics/lut.rs                   (LINEAR_TO_SRGB_LUT, 16384 u8 entries)
  c_components/tests/test_idct_scaling.rs:5-19         (KAT: alternating 0/255 block -> 188)
They are normalised into compact fixtures so the GPU box (which has no /root/reference) can replay them:
  weights_golden.json.gz   {"plain": [[filter_id, from, to, [[w6dp,...] per output]], ...],
                            "params": [[filter_name, variant, from, to, [[...]]], ...]}
  linear_to_srgb_lut.bin   16384 bytes
  ref_block_scalers.npz    outputs of the reference's own compiled flow_scale_spatial[_srgb]_NxN
                           (oracle/_ref/libref_idct.so) on seeded random 8x8 blocks
"""
import ctypes, gzip, json, os, re, sys
import numpy as np

REF = os.environ.get("IMAGEFLOW_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROW = re.compile(r"x=(\d+) from \(([^)]*)\)")


def parse_rows(text):
    return [[w for w in m.group(2).split()] for m in ROW.finditer(text)]


def weights_plain():
    out = []
    path = os.path.join(REF, "imageflow_core/tests/integration/weights.txt")
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"filter_(\d+) \(\s*(\d+)px to\s*(\d+)px\): (.*)$", line)
        if not m:
            continue
        out.append([int(m.group(1)), int(m.group(2)), int(m.group(3)), parse_rows(m.group(4))])
    return out


def weights_params():
    out = []
    path = os.path.join(REF, "imageflow_core/tests/integration/weights_params.txt")
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"(\w+) (\S+) \(\s*(\d+)px to\s*(\d+)px\): (.*)$", line)
        if not m:
            continue
        out.append([m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), parse_rows(m.group(5))])
    return out


def lut_table():
    src = open(os.path.join(REF, "imageflow_core/src/graphics/lut.rs")).read()
    body = src[src.index("LINEAR_TO_SRGB_LUT: [u8; 16384] = [") :]
    body = body[body.index("= [") + 3 : body.index("];")]
    vals = [int(v) for v in re.findall(r"\d+", body)]
    assert len(vals) == 16384, len(vals)
    return bytes(vals)


def ref_block_scalers():
    so = os.path.join(HERE, "..", "..", "oracle", "_ref", "libref_idct.so")
    lib = ctypes.CDLL(so)
    rng = np.random.default_rng(20260921)
    blocks = rng.integers(0, 256, size=(64, 64), dtype=np.uint8)
    blocks[0] = np.tile(np.array([0, 255] * 4, dtype=np.uint8), 8)          # the reference KAT block
    blocks[1] = 0
    blocks[2] = 255
    res = {"blocks": blocks}
    for srgb in (0, 1):
        for n in range(1, 8):
            name = f"flow_scale_spatial_{'srgb_' if srgb else ''}{n}x{n}"
            fn = getattr(lib, name)
            outs = np.zeros((len(blocks), n, n), dtype=np.uint8)
            for b, blk in enumerate(blocks):
                rows = [np.zeros(8, dtype=np.uint8) for _ in range(n)]
                ptrs = (ctypes.POINTER(ctypes.c_uint8) * n)(*[r.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) for r in rows])
                inp = np.ascontiguousarray(blk)
                fn(inp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ptrs, ctypes.c_uint32(0))
                for r in range(n):
                    outs[b, r] = rows[r][:n]
            res[name] = outs
    assert int(res["flow_scale_spatial_srgb_1x1"][0, 0, 0]) == 188, "reference KAT (test_idct_scaling.rs:5-19) failed"
    return res


def main():
    g = {"plain": weights_plain(), "params": weights_params()}
    print("plain rows", len(g["plain"]), "param rows", len(g["params"]))
    with gzip.GzipFile(os.path.join(HERE, "weights_golden.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(g, separators=(",", ":")).encode())
    open(os.path.join(HERE, "linear_to_srgb_lut.bin"), "wb").write(lut_table())
    np.savez_compressed(os.path.join(HERE, "ref_block_scalers.npz"), **ref_block_scalers())
    print("ok")


if __name__ == "__main__":
    sys.exit(main())
