"""Pin the oracle's weight builder against the reference's golden vectors
(imageflow_core/tests/integration/weights.txt + weights_params.txt, normalised by tests/golden/make_golden.py).
Comparison is the reference's own format: each weight printed with 6 decimals."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def golden(golden_dir):
    with gzip.open(os.path.join(golden_dir, "weights_golden.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def _fmt(rows_left_count_w):
    left, count, w = rows_left_count_w
    out, off = [], 0
    for n in count:
        out.append(["%.6f" % float(v) for v in w[off:off + n]])
        off += n
    return out


def _same(a, b):
    # "-0.000000" vs "0.000000": printing of a negative value that rounds to zero; compare numerically at 6 dp
    if len(a) != len(b):
        return False
    for ra, rb in zip(a, b):
        if len(ra) != len(rb):
            return False
        for x, y in zip(ra, rb):
            if x != y and float(x) != float(y):
                return False
    return True


def test_plain_golden_all_rows(golden):
    assert len(golden["plain"]) == 660
    bad = []
    for fid, frm, to, rows in golden["plain"]:
        got = _fmt(O.weights(fid, to, frm))
        if not _same(got, rows):
            bad.append((fid, frm, to))
    assert not bad, bad[:10]


def _parse_variant(v):
    mode, val, ks = O.LOBE_NATURAL, 0.0, 1.0
    if v != "default":
        for part in v.split("+"):
            k, x = part.split("=")
            if k == "sharpen":
                mode, val = O.LOBE_SHARPEN_PERCENT, float(x)
            elif k == "lobe_exact":
                mode, val = O.LOBE_EXACT, float(x)
            elif k == "kernel_scale":
                ks = float(x)
            else:
                raise AssertionError(k)
    return mode, val, ks


def test_param_golden_all_rows(golden):
    assert len(golden["params"]) == 1680
    bad = []
    for name, variant, frm, to, rows in golden["params"]:
        mode, val, ks = _parse_variant(variant)
        if not rows:
            # the reference printed no weights for this row: populate_weights returned an error
            # (TotalWeightZero, weights.rs:752-755) -- the oracle must refuse as well
            with pytest.raises(RuntimeError):
                O.weights(O.FILTER_IDS[name], to, frm, mode, val, ks)
            continue
        got = _fmt(O.weights(O.FILTER_IDS[name], to, frm, mode, val, ks))
        if not _same(got, rows):
            bad.append((name, variant, frm, to))
    assert not bad, (len(bad), bad[:10])


def test_tap_counts_at_baseline_shapes():
    # SURVEY.md section 8a tap table
    left, count, w = O.weights(2, 200, 3840)
    assert (count.min(), count.max()) == (48, 77) and len(w) == 15282
    left, count, w = O.weights(2, 200, 2160)
    assert (count.min(), count.max()) == (27, 44) and len(w) == 8598
    left, count, w = O.weights(6, 400, 7680, O.LOBE_SHARPEN_PERCENT, 15.0)
    assert count.max() == 116 and len(w) == 45906
    assert abs(O.lib().ifo_natural_negative_ratio(2) - 0.027003) < 1e-6
    assert abs(O.lib().ifo_natural_negative_ratio(6) - 0.137268) < 1e-6


def test_weights_sum_to_one():
    for fid in (2, 6, 4, 16, 24):
        left, count, w = O.weights(fid, 113, 2160)
        off = 0
        for n in count:
            assert abs(float(np.sum(w[off:off + n].astype(np.float64))) - 1.0) < 1e-5
            off += n
