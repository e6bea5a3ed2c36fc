"""Pin the encode-side oracle (oracle/jpeg_oracle.c jo_jpeg_forward: rgb_ycc + down-sampling + islow FDCT + quantisation
+ dummy blocks) against libjpeg-turbo: the committed files in tests/golden/jpeg_encode_cases.npz were written by
Pillow from the committed RGB pictures; entropy-decoding them (the oracle's baseline decoder, itself pinned by
test_oracle_jpeg.py) must give exactly the coefficient planes the oracle computes from the pictures."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from imageflow_amd.codecs import mozjpeg as M


@pytest.fixture(scope="module")
def cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_encode_cases.npz"))
    return z, [str(n) for n in z["names"]]


def to_bgra(rgb, stride=None):
    h, w = rgb.shape[:2]
    stride = stride or O.stride_for_width(w)
    out = np.zeros((h, stride), np.uint8)
    px = out[:, :4 * w].reshape(h, w, 4)
    px[..., 0], px[..., 1], px[..., 2], px[..., 3] = rgb[..., 2], rgb[..., 1], rgb[..., 0], 255
    return out, stride


def test_forward_oracle_reproduces_libjpeg_turbo_coefficients(cases):
    z, names = cases
    assert len(names) == 108
    seen = set()
    for i, name in enumerate(names):
        j = O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
        src = z[f"src_{i}"]
        h, w = src.shape[:2]
        assert (j["width"], j["height"], j["ncomp"]) == (w, h, 3), name
        bgra, stride = to_bgra(src)
        coef = O.jpeg_forward(bgra, w, h, stride, j["hs"], j["vs"], j["qt"])
        for c in range(3):
            assert coef[c].shape == j["coef"][c].shape, name
            assert np.array_equal(coef[c], j["coef"][c]), (name, c)
        seen.add((tuple(j["hs"]), tuple(j["vs"])))
    assert seen == {((1, 1, 1), (1, 1, 1)), ((2, 1, 1), (1, 1, 1)), ((2, 1, 1), (2, 1, 1))}


def test_alpha_and_row_padding_are_ignored(cases):
    z, _ = cases
    src = z["src_4"]
    h, w = src.shape[:2]
    qt = M.quant_tables_for_quality(80)
    a, stride = to_bgra(src)
    b, stride_b = to_bgra(src, stride + 64)
    b[:, 4 * w:] = 0xA5
    b[:, 3:4 * w:4] = 7
    for hs, vs in (([1, 1, 1], [1, 1, 1]), ([2, 1, 1], [2, 1, 1])):
        ca, cb = O.jpeg_forward(a, w, h, stride, hs, vs, qt), O.jpeg_forward(b, w, h, stride_b, hs, vs, qt)
        assert all(np.array_equal(x, y) for x, y in zip(ca, cb))


def test_quality_tables_match_the_ones_libjpeg_wrote(cases):
    """jpeg_set_quality's tables (host logic mirrored in imageflow_amd.codecs.mozjpeg) == the DQT segments in the files."""
    z, names = cases
    for i, name in enumerate(names):
        q = int(name.rsplit("_q", 1)[1])
        j = O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
        assert np.array_equal(M.quant_tables_for_quality(q), j["qt"]), name
    assert np.all(M.quant_tables_for_quality(100) == 1)
    assert M.quant_tables_for_quality(0)[0, 0] == M.quant_tables_for_quality(1)[0, 0] == 255      # force_baseline clamp


def test_sampling_factors_follow_the_reference_mapping():
    assert M.sampling_factors((2, 2), (2, 2)) == ([2, 1, 1], [2, 1, 1])           # mozjpeg.rs:141-149
    assert M.sampling_factors((2, 1), (2, 1)) == ([2, 1, 1], [1, 1, 1])
    assert M.sampling_factors((1, 1), (1, 1)) == ([1, 1, 1], [1, 1, 1])


def test_forward_then_inverse_is_close_to_the_picture():
    """Size-independent property: at quality 100 (all Q = 1) decode(encode(x)) stays within a few levels for 4:4:4."""
    rng = np.random.default_rng(11)
    w, h = 61, 43
    y, x = np.mgrid[0:h, 0:w]
    src = np.stack([x * 4 % 256, y * 5 % 256, (x + y) * 2 % 256], -1).astype(np.uint8)
    bgra, stride = to_bgra(src)
    qt = M.quant_tables_for_quality(100)
    hs, vs = [1, 1, 1], [1, 1, 1]
    coef = O.jpeg_forward(bgra, w, h, stride, hs, vs, qt)
    bw, bh = O.jpeg_block_geometry(w, h, hs, vs)
    back = O.jpeg_idct_color(dict(width=w, height=h, ncomp=3, hs=hs, vs=vs, bw=bw, bh=bh, coef=coef, qt=qt))
    d = np.abs(back[:, :4 * w].reshape(h, w, 4)[..., :3].astype(int) - bgra[:, :4 * w].reshape(h, w, 4)[..., :3].astype(int))
    assert d.max() <= 3
