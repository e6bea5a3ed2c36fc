"""The block scalers' tables in the product (CPU-side checks; the kernel is checked in tests/test_gpu_block_scalers.py)."""
import os

import numpy as np

from imageflow_amd.codecs import block_scalers as BS
from imageflow_amd.graphics import weights as W


def test_tables_equal_the_reference_files_data(golden_dir):
    z = np.load(os.path.join(golden_dir, "block_scaler_tables.npz"))
    for n in range(1, 8):
        w, d, s2l, l2s = BS.tables(n)
        assert np.array_equal(w, z["weights"][n, :n]) and np.array_equal(d, z["log2div"][n, :n])
        assert np.array_equal(w.astype(int).sum(1), 1 << d.astype(int))
        assert np.array_equal(s2l, z["lut_s2l"]) and np.array_equal(l2s, z["lut_l2s"])


def test_lut_roundtrip_12bit(golden_dir):
    """tests/integration/variation.rs:133-151: lut_linear_to_srgb[lut_srgb_to_linear[i]] == i for every byte."""
    _, _, s2l, l2s = BS.tables(1)
    assert np.array_equal(l2s[s2l.astype(int)], np.arange(256))


def _find_integral(ws, left):
    f32 = np.float32
    for bits in range(10, 6, -1):
        div = 1 << bits
        d = f32(div)
        scalar = f32(div)
        while scalar < d + f32(10):
            eight, s, failed = [0] * 8, 0, False
            for i, v in enumerate(ws):
                t = int(f32(v) * scalar)
                if t > 127 or t < -128:
                    failed = True
                    break
                eight[left + i] = t
                s += t
            if not failed and s == div:
                return eight
            scalar = f32(d + (-(scalar - d + f32(0.125)) if scalar > d else -(scalar - d - f32(0.125))))
    return None


def test_no_catalogue_filter_regenerates_the_committed_weights(golden_dir):
    """Documents why the weights ship as data: the generator (variation.rs) fed with any of today's 31 filters does
    not give back the 8 -> 2 row the committed C file uses."""
    z = np.load(os.path.join(golden_dir, "block_scaler_tables.npz"))
    target = z["weights"][2, 0].tolist()
    hits = []
    for fid in range(1, 32):
        try:
            p = W.populate_weights(fid, 2, 8)
        except Exception:
            continue
        got = _find_integral(p.weights[: int(p.count[0])], int(p.left_pixel[0]))
        if got is not None:
            g = np.array(got)
            div = g.sum()
            for r in (16, 8, 4, 2):                      # the generator's power-of-two reduction
                if np.all(g % r == 0):
                    g = g // r
                    break
            if g.tolist() == target:
                hits.append(fid)
    assert hits == []
