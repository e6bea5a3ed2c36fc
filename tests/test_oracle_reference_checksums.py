"""Pin of the resampler oracle against outputs the REFERENCE ITSELF stored.

The reference's visual tests hash their BGRA result with seahash and keep the id in `*.checksums`
(imageflow_core/tests/integration/common/mod.rs:307-324).  Three of those tests build their input from
create_canvas / fill_rect / expand_canvas, so they can be replayed offline (tests/reference_canvases.py).

What this establishes (DESIGN.md section 2 carries the table):
  * flags = 0 (the oracle's arithmetic contract) reproduces all three stored checksums of commit 8ca16e2d / 59b0ceb7:
    populate_weights for up-scaling at integer and non-integer ratios, linear-light filtering, premultiplied alpha,
    un-premultiply + encode, the alpha byte, ReplaceSelf semantics -- pinned to the reference's own pixels.
  * the semantic negative controls (straight alpha, sRGB-space filtering, a different filter) MISS, so the checksums do
    discriminate those choices.
  * the 128 rounding-level variants (pass order x accumulation x reciprocal x encode x table x alpha rounding) ALL hit:
    the stored outputs cannot tell them apart.  Their distance on uniform-noise frames is measured here: max 1 LSB,
    on < 1 % of the bytes.  That number is what "+-1 LSB class" means for the part of the contract that stays
    oracle-defined.
"""
import ctypes as C
import itertools

import numpy as np
import pytest

from oracle import oracle as O
from tests import reference_canvases as R
from tests.seahash import bitmap_checksum, checksum_id_digits, seahash

HFIRST, RCP, ENC_EXACT, S2L_F64, A_RNE, NO_PREMUL, SRGB_SPACE = 1, 8, 16, 32, 64, 128, 256


def variant_render(src, ow, oh, filt, flags, alpha=True, working_space=O.LINEAR):
    L = O.lib()
    L.ifo_scale_and_render_variant.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_int, C.c_uint32]
    h, w, _ = src.shape
    st = O.stride_for_width(w)
    inp = np.zeros((h, st), np.uint8)
    inp[:, :w * 4] = src.reshape(h, w * 4)
    cst = O.stride_for_width(ow)
    can = np.zeros((oh, cst), np.uint8)
    rc = L.ifo_scale_and_render_variant(inp.ctypes.data, w, h, st, int(alpha), can.ctypes.data, cst, ow, oh, filt, 0.0,
                                        working_space, flags)
    assert rc == 0
    return can[:, :ow * 4].reshape(oh, ow, 4).copy()


def oracle_render(src, ow, oh, filt):
    h, w, _ = src.shape
    st = O.stride_for_width(w)
    inp = np.zeros((h, st), np.uint8)
    inp[:, :w * 4] = src.reshape(h, w * 4)
    cst = O.stride_for_width(ow)
    can = np.zeros((oh, cst), np.uint8)
    rc, _ = O.scale_and_render(inp, w, h, can, ow, oh, 0, 0, ow, oh, filter_id=filt, working_space=O.LINEAR,
                               compositing=O.REPLACE_SELF, alpha_meaningful=True)
    assert rc == 0
    return can[:, :ow * 4].reshape(oh, ow, 4).copy()


def rounding_variants():
    for hf, acc, rcp, enc, s2l, rne in itertools.product([0, 1], [0, 1, 2, 3], [0, 1], [0, 1], [0, 1], [0, 1]):
        yield hf * HFIRST | (acc << 1) | rcp * RCP | enc * ENC_EXACT | s2l * S2L_F64 | rne * A_RNE


def test_seahash_known_answer():
    assert seahash(b"to be or not to be") == 1988685042348123509          # the crate's documented example
    assert seahash(b"") == seahash(b"")


@pytest.mark.parametrize("name", list(R.calibration_canvases()))
def test_hash_layout_and_id_format(name):
    img, want = R.calibration_canvases()[name]
    assert checksum_id_digits(img) == want, bitmap_checksum(img)


@pytest.mark.parametrize("name", list(R.RESAMPLE_CASES))
def test_oracle_reproduces_reference_checksum(name):
    make, ow, oh, filt, want = R.RESAMPLE_CASES[name]
    out = oracle_render(make(), ow, oh, filt)
    assert checksum_id_digits(out) == want, bitmap_checksum(out)
    assert np.array_equal(out, variant_render(make(), ow, oh, filt, 0))   # the variant harness at flags 0 IS the oracle


def test_semantic_controls_miss():
    make, ow, oh, filt, want = R.RESAMPLE_CASES["test_expand_rect fill_expand_hermite_linear"]
    src = make()
    assert checksum_id_digits(variant_render(src, ow, oh, filt, NO_PREMUL)) != want
    assert checksum_id_digits(variant_render(src, ow, oh, filt, SRGB_SPACE)) != want
    for other in (2, 6, 13, 22, 24):                                      # Robidoux, Lanczos, CatmullRom, Triangle, Box
        assert checksum_id_digits(variant_render(src, ow, oh, other, 0)) != want


def test_rounding_variants_all_hit_and_their_distance_is_one_lsb():
    hits = {}
    for name, (make, ow, oh, filt, want) in R.RESAMPLE_CASES.items():
        src = make()
        hits[name] = sum(checksum_id_digits(variant_render(src, ow, oh, filt, f)) == want for f in rounding_variants())
    assert all(v == 128 for v in hits.values()), hits
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, (301, 333, 4), dtype=np.uint8)
    report = []
    for alpha in (True, False):
        for ow, oh, filt in ((41, 37, 2), (150, 140, 6), (500, 450, 4)):
            base = variant_render(noise, ow, oh, filt, 0, alpha=alpha).astype(np.int16)
            worst, frac = 0, 0.0
            for f in rounding_variants():
                d = np.abs(variant_render(noise, ow, oh, filt, f, alpha=alpha).astype(np.int16) - base)
                worst, frac = max(worst, int(d.max())), max(frac, float((d > 0).mean()))
            report.append((alpha, ow, oh, filt, worst, frac))
            assert worst <= 1 and frac < 0.01, report[-1]
    print(report)
