"""Oracle colour / flatten known-answer tests, restating the reference's own
(imageflow_core/tests/integration/color_conversion.rs:376-417, :702-747, :824-1100;
 imageflow_core/tests/integration/variation.rs:133-151 style round trips)."""
import os

import numpy as np

from oracle import oracle as O


def test_linear_to_srgb_lut_matches_reference_table(golden_dir):
    # graphics/lut.rs table, extracted by tests/golden/make_golden.py
    ref = np.frombuffer(open(os.path.join(golden_dir, "linear_to_srgb_lut.bin"), "rb").read(), np.uint8)
    _, _, l2s = O.tables()
    assert ref.shape == (16384,)
    assert np.array_equal(ref, l2s)


def test_lut_formula_f64():
    # color_conversion.rs:376-404
    _, _, l2s = O.tables()
    i = np.arange(16384, dtype=np.float64) / 16383.0
    srgb = np.where(i <= 0.0031308, 12.92 * i, 1.055 * np.power(i, 1.0 / 2.4) - 0.055)
    exp = np.clip(srgb * 255.0 + 0.5, 0, 255).astype(np.uint8)
    assert np.array_equal(exp, l2s)


def test_lut_function_bounds():
    # color_conversion.rs:408-417
    L = O.lib()
    assert L.ifo_linear_to_srgb_lut(0.0) == 0
    assert L.ifo_linear_to_srgb_lut(1.0) == 255
    assert L.ifo_linear_to_srgb_lut(-0.1) == 0
    assert L.ifo_linear_to_srgb_lut(1.5) == 255
    assert abs(L.ifo_linear_to_srgb_lut(0.5) - 188) <= 1
    assert L.ifo_linear_to_srgb_lut(float("nan")) == 0


def test_srgb_linear_roundtrip_lossless():
    # color_conversion.rs:702-747: all 256 values must round trip exactly
    s2l, s2f, _ = O.tables()
    L = O.lib()
    for v in range(256):
        assert L.ifo_linear_to_srgb_lut(float(s2l[v])) == v
        assert L.ifo_uchar_clamp_ff(255.0 * float(s2f[v])) == v


def test_s2l_table_definition():
    # color.rs:31-45,85-91 evaluated in f32
    s2l, s2f, _ = O.tables()
    v = np.arange(256, dtype=np.float32) * np.float32(1.0 / 255.0)
    assert np.array_equal(v, s2f)
    lo = v / np.float32(12.92)
    assert np.array_equal(s2l[v <= np.float32(0.04045)], lo[v <= np.float32(0.04045)])
    hi = np.power(((v + np.float32(0.055)) / np.float32(1.055)).astype(np.float64), 2.4)
    assert np.allclose(s2l[v > 0.04045], hi[v > 0.04045], rtol=2e-7)
    assert s2l[0] == 0.0 and s2l[255] == 1.0
    assert np.all(np.diff(s2l) > 0)


def test_uchar_clamp_ff():
    # color.rs:101-108
    f = O.lib().ifo_uchar_clamp_ff
    assert f(0.0) == 0 and f(0.49) == 0 and f(0.5) == 1 and f(254.5) == 255 and f(255.4) == 255
    assert f(256.0) == 255 and f(1e9) == 255 and f(-0.4) == 0 and f(-3.0) == 0 and f(-1e9) == 0
    assert f(float("nan")) == 0
    assert f(127.49) == 127 and f(127.5) == 128


def _matte_ref(px, matte):
    """blend.rs:21-52 in numpy float32 for one pixel."""
    s2l, _, l2s = O.tables()
    f = np.float32
    a255 = f(1.0) / f(255.0)
    pa = px[3]
    if pa == 0:
        return list(matte)
    if pa == 255:
        return list(px)
    paf = f(pa) * a255
    ma = (f(1.0) - paf) * (f(matte[3]) * a255)
    fa = ma + paf
    out = []
    for c in range(3):
        v = (s2l[px[c]] * paf + s2l[matte[c]] * ma) / fa
        idx = int(np.clip(f(v) * f(16383.0), 0, 16383))
        out.append(int(l2s[idx]))
    out.append(int(O.lib().ifo_uchar_clamp_ff(float(f(255.0) * fa))))
    return out


def test_apply_matte_semantics():
    rng = np.random.default_rng(7)
    w, h = 37, 5
    stride = O.stride_for_width(w)
    assert stride == 192
    img = rng.integers(0, 256, size=(h, stride), dtype=np.uint8)
    img[0, 3:4 * w:8] = 0
    img[1, 3:4 * w:8] = 255
    for matte in ((255, 255, 255, 255), (0, 0, 255, 128), (10, 200, 30, 0)):
        m32 = matte[0] | (matte[1] << 8) | (matte[2] << 16) | (matte[3] << 24)
        got = img.copy()
        assert O.apply_matte(got, w, h, stride, m32) == 0
        for y in range(h):
            for x in range(w):
                exp = _matte_ref(img[y, 4 * x:4 * x + 4], matte)
                assert list(got[y, 4 * x:4 * x + 4]) == exp, (matte, x, y)
        assert np.array_equal(got[:, 4 * w:], img[:, 4 * w:])     # padding untouched
    same = img.copy()
    O.apply_matte(same, w, h, stride, 0xFFFFFFFF, alpha_meaningful=False)   # blend.rs:11-13 no-op
    assert np.array_equal(same, img)


def test_matte_compositing_kats():
    """color_conversion.rs:824-981: 10x10 RGBA(255,0,0,128) -> 5x5 over white: R~255, G=B~LUT(1-128/255) (+-2).
    :989: alpha=0 pixels become the matte."""
    s2l, _, l2s = O.tables()
    w = h = 10
    st = O.stride_for_width(w)
    src = np.zeros((h, st), np.uint8)
    src[:, 0:4 * w:4] = 0      # B
    src[:, 1:4 * w:4] = 0      # G
    src[:, 2:4 * w:4] = 255    # R
    src[:, 3:4 * w:4] = 128
    cst = O.stride_for_width(5)
    canvas = np.zeros((5, cst), np.uint8)
    canvas[:, :20] = 255       # canvas pre-filled with matte (bitmaps.rs:829-837)
    rc, _ = O.scale_and_render(src, w, h, canvas, 5, 5, 0, 0, 5, 5, filter_id=2, compositing=O.BLEND_WITH_MATTE,
                               matte_bgra=0xFFFFFFFF, alpha_meaningful=True)
    assert rc == 0
    expected_gb = int(l2s[int((1.0 - 128 / 255.0) * 16383)])
    px = canvas[:, :20].reshape(5, 5, 4)
    assert np.all(np.abs(px[..., 2].astype(int) - 255) <= 2)
    assert np.all(np.abs(px[..., 0].astype(int) - expected_gb) <= 2)
    assert np.all(np.abs(px[..., 1].astype(int) - expected_gb) <= 2)
    assert np.all(px[..., 3] == 255)
    # fully transparent source -> matte everywhere
    src[:, 3:4 * w:4] = 0
    canvas[:] = 0
    rc, _ = O.scale_and_render(src, w, h, canvas, 5, 5, 0, 0, 5, 5, compositing=O.BLEND_WITH_MATTE,
                               matte_bgra=0xFF0000FF, alpha_meaningful=True)
    px = canvas[:, :20].reshape(5, 5, 4)
    assert np.all(px == np.array([255, 0, 0, 255], np.uint8))


def test_resample_identity_and_constant():
    """A constant image stays constant under every filter (weights sum to 1 within f32 rounding ->
    the encoded byte must round back), and alpha is forced to 255 when not meaningful (scaling.rs:227-232)."""
    for fid in (2, 6, 4, 16, 24, 22):
        for val in (0, 1, 17, 128, 254, 255):
            w, h, ow, oh = 64, 48, 20, 13
            st = O.stride_for_width(w)
            src = np.full((h, st), val, np.uint8)
            src[:, 3::4] = 9
            cst = O.stride_for_width(ow)
            canvas = np.zeros((oh, cst), np.uint8)
            rc, f32 = O.scale_and_render(src, w, h, canvas, ow, oh, 0, 0, ow, oh, filter_id=fid, want_f32=True)
            assert rc == 0
            px = canvas[:, :4 * ow].reshape(oh, ow, 4)
            assert np.all(px[..., 3] == 255)
            assert np.all(np.abs(px[..., :3].astype(int) - val) <= 1), (fid, val)
            assert np.all(f32[..., 3] == 1.0)


def test_scale_and_render_errors():
    src = np.zeros((4, 64), np.uint8)
    canvas = np.zeros((4, 64), np.uint8)
    rc, _ = O.scale_and_render(src, 4, 4, canvas, 4, 4, 2, 0, 4, 4)    # x+w > cw  (scaling.rs:24-29)
    assert rc == 1
    rc, _ = O.scale_and_render(src, 4, 4, canvas, 4, 4, 0, 0, 4, 4, filter_id=99)
    assert rc == 1


def test_blend_with_self_matches_in_tree_formula():
    """scaling.rs:254-287 replayed in numpy on the oracle's own f32 buffer."""
    rng = np.random.default_rng(3)
    s2l, _, l2s = O.tables()
    w, h, ow, oh = 40, 30, 16, 12
    st, cst = O.stride_for_width(w), O.stride_for_width(ow)
    src = rng.integers(0, 256, size=(h, st), dtype=np.uint8)
    canvas0 = rng.integers(0, 256, size=(oh, cst), dtype=np.uint8)
    canvas = canvas0.copy()
    rc, f32 = O.scale_and_render(src, w, h, canvas, ow, oh, 0, 0, ow, oh, compositing=O.BLEND_WITH_SELF,
                                 alpha_meaningful=True, want_f32=True)
    assert rc == 0
    f = np.float32
    L = O.lib()
    for y in range(oh):
        for x in range(ow):
            sp = f32[y, x]
            cp = canvas0[y, 4 * x:4 * x + 4]
            if sp[3] > f(0.994):
                exp = [L.ifo_linear_to_srgb_lut(float(sp[c])) for c in range(3)] + [255]
            else:
                dc = (f(1) - sp[3]) * (f(1.0) / f(255.0) * f(int(cp[3])) + f(0))
                fa = sp[3] + dc
                exp = [L.ifo_linear_to_srgb_lut(float((sp[c] + dc * s2l[cp[c]]) / fa)) for c in range(3)]
                exp.append(L.ifo_uchar_clamp_ff(float(fa * f(255.0))))
            assert list(canvas[y, 4 * x:4 * x + 4]) == exp


def test_device_form_of_uchar_clamp_ff_equals_the_f64_form():
    """The kernels evaluate uchar_clamp_ff without f64 (csrc/resample_device.hpp, csrc/bitmap_ops.hip): negative and NaN
    -> 0, else min(255, floor(v) + (frac(v) >= 0.5)).  Checked against the oracle's literal `(v as f64 + 0.5) as i16 as
    u16` on a dense sample of float bit patterns (every 64th pattern of the whole range, every pattern in [0, 260] is
    covered by the stride-1 band) -- the exhaustive 2^32 comparison was run once on the host (0 mismatches)."""
    f = np.float32

    def device_form(v):
        c = np.minimum(v, f(300.0))                       # fminf: NaN -> 300 (fixed by the final select)
        c = np.where(np.isnan(v), f(300.0), c)
        fl = np.floor(c)
        i = fl.astype(np.int64) + ((c - fl) >= f(0.5))
        i = np.minimum(i, 255)
        return np.where(v >= 0, i, 0).astype(np.uint8)

    def f64_form(v):
        t = v.astype(np.float64) + 0.5
        i = np.where(np.isnan(t), 0, np.clip(np.trunc(np.nan_to_num(t, nan=0.0, posinf=40000.0, neginf=-40000.0)), -32768, 32767)).astype(np.int64)
        r = i & 0xFFFF
        return np.where(r > 255, np.where(v < 0, 0, 255), r).astype(np.uint8)

    with np.errstate(invalid="ignore", over="ignore"):
        bits = np.arange(0, 1 << 32, 64, dtype=np.uint64).astype(np.uint32)
        v = bits.view(np.float32)
        assert np.array_equal(device_form(v), f64_form(v))
        lo, hi = np.float32(0).view(np.uint32), np.float32(260).view(np.uint32)
        band = np.arange(int(hi) - (1 << 22), int(hi), dtype=np.uint32).view(np.float32)       # [~128, 260): every pattern
        assert np.array_equal(device_form(band), f64_form(band))
        # the f64 form IS the oracle's: spot-check through the C function
        rng = np.random.default_rng(0)
        pick = np.concatenate([rng.choice(v, 2000), rng.choice(band, 2000), np.array([np.nan, np.inf, -np.inf, -0.0, 0.5, 254.5, 255.5], np.float32)])
        L = O.lib()
        assert [int(L.ifo_uchar_clamp_ff(float(x))) for x in pick] == [int(x) for x in f64_form(pick)]
