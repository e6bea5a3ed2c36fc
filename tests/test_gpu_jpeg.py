"""JPEG pixel stage on the GPU (through the C ABI) vs the oracle, bit for bit: every committed file, random
coefficient blocks (incl. values that exercise libjpeg's range-limit wrap), 4K 4:2:0 frames, batches, host drop-in."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs.mozjpeg_decoder import JpegPixelStage, jpeg_idct_color_host  # noqa: E402
from imageflow_amd.errors import ErrorKind, FlowError  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import util as U  # noqa: E402

DEV = "cuda:0"


def run_stage(js):
    """js: list of oracle coefficient dicts with identical geometry -> uint8 [n, h, stride] from the GPU."""
    j0 = js[0]
    n = len(js)
    st = JpegPixelStage(j0["width"], j0["height"], j0["ncomp"], j0["hs"], j0["vs"], n, DEV)
    assert st.blocks_w[:j0["ncomp"]] == j0["bw"][:j0["ncomp"]] and st.blocks_h[:j0["ncomp"]] == j0["bh"][:j0["ncomp"]]
    coef = [torch.from_numpy(np.stack([j["coef"][c] for j in js])).to(DEV) for c in range(j0["ncomp"])]
    qt = torch.from_numpy(np.stack([j["qt"][:j0["ncomp"]] for j in js]).astype(np.int16)).to(DEV)
    out = st.read_frames(coef, qt)
    torch.cuda.synchronize()
    assert out.alpha_meaningful is False
    return out.to_numpy()


def test_every_committed_file_matches_oracle_and_libjpeg(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    for i, name in enumerate(z["names"]):
        j = O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
        exp = O.jpeg_idct_color(j)
        got = run_stage([j])[0]
        assert np.array_equal(got, exp), str(name)
        h, w = z[f"rgb_{i}"].shape[:2]
        assert np.array_equal(got[:, :4 * w].reshape(h, w, 4)[..., [2, 1, 0]], z[f"rgb_{i}"]), str(name)


def _random_case(rng, w, h, hs, vs, ncomp, amp):
    hmax, vmax = max(hs[:ncomp]), max(vs[:ncomp])
    mw, mh = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    j = dict(width=w, height=h, ncomp=ncomp, hs=list(hs), vs=list(vs), bw=[0, 0, 0], bh=[0, 0, 0], coef=[], qt=None)
    for c in range(3):
        if c < ncomp:
            j["bw"][c], j["bh"][c] = mw * hs[c], mh * vs[c]
            co = rng.integers(-amp, amp + 1, size=(j["bh"][c], j["bw"][c], 64)).astype(np.int16)
            co[..., 10:] = (co[..., 10:] * (rng.random(size=co[..., 10:].shape) < 0.2)).astype(np.int16)
            j["coef"].append(co)
        else:
            j["coef"].append(np.zeros((1, 1, 64), np.int16))
    j["qt"] = rng.integers(1, 64, size=(3, 64)).astype(np.uint16)
    return j


@pytest.mark.parametrize("hs,vs,ncomp", [((2, 1, 1), (2, 1, 1), 3), ((2, 1, 1), (1, 1, 1), 3), ((1, 1, 1), (1, 1, 1), 3),
                                          ((1, 0, 0), (1, 0, 0), 1)])
@pytest.mark.parametrize("amp", [20, 400])
def test_random_coefficients(hs, vs, ncomp, amp):
    rng = np.random.default_rng(amp + ncomp)
    for (w, h) in ((129, 67), (16, 8), (1, 1), (333, 100)):
        js = [_random_case(rng, w, h, hs, vs, ncomp, amp) for _ in range(3)]
        got = run_stage(js)
        for k, j in enumerate(js):
            assert np.array_equal(got[k], O.jpeg_idct_color(j)), (w, h, k)


def test_4k_420_frames():
    """BASELINE config 4 geometry: 3840x2160 4:2:0 (240x135 MCUs, 194 400 blocks, 24 883 200 coefficient bytes)."""
    rng = np.random.default_rng(4)
    js = [_random_case(rng, 3840, 2160, (2, 1, 1), (2, 1, 1), 3, 60) for _ in range(2)]
    assert sum(js[0]["bw"][c] * js[0]["bh"][c] for c in range(3)) == 194400
    got = run_stage(js)
    for k, j in enumerate(js):
        assert np.array_equal(got[k], O.jpeg_idct_color(j))


def test_host_buffer_drop_in(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    for i in (0, 6, 12, 35):
        j = O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
        got = jpeg_idct_color_host(j["coef"], j["qt"], j["ncomp"], j["hs"], j["vs"], j["width"], j["height"])
        assert np.array_equal(got, O.jpeg_idct_color(j))


def test_unsupported_and_invalid_arguments():
    with pytest.raises(FlowError) as e:
        JpegPixelStage(16, 16, 3, (2, 2, 1), (1, 1, 1), 1, DEV)          # chroma components must be 1x1
    assert e.value.kind == ErrorKind.MethodNotImplemented
    with pytest.raises(FlowError) as e:
        JpegPixelStage(16, 16, 3, (2, 1, 1), (2, 1, 1), 1, DEV, scale_num=7)   # 7/8 is never requested by the decoder
    assert e.value.kind == ErrorKind.MethodNotImplemented
    with pytest.raises(FlowError) as e:
        JpegPixelStage(0, 16, 3, (2, 1, 1), (2, 1, 1), 1, DEV)
    assert e.value.kind == ErrorKind.InvalidArgument
    with pytest.raises(FlowError) as e:
        JpegPixelStage(16, 16, 4, (1, 1, 1), (1, 1, 1), 1, DEV)          # CMYK stays on the CPU path
    assert e.value.kind == ErrorKind.MethodNotImplemented


# ---- reduced-size decode (scale_num 4, 2, 1) ---------------------------------------------------------------------
def run_stage_scaled(js, scale_num, luma_mode):
    j0 = js[0]
    st = JpegPixelStage(j0["width"], j0["height"], j0["ncomp"], j0["hs"], j0["vs"], len(js), DEV, scale_num=scale_num,
                        luma_spatial=luma_mode != 0, luma_srgb=luma_mode == 2)
    coef = [torch.from_numpy(np.stack([j["coef"][c] for j in js])).to(DEV) for c in range(j0["ncomp"])]
    qt = torch.from_numpy(np.stack([j["qt"][:j0["ncomp"]] for j in js]).astype(np.int16)).to(DEV)
    out = st.read_frames(coef, qt)
    torch.cuda.synchronize()
    assert (out.w, out.h) == ((j0["width"] * scale_num + 7) // 8, (j0["height"] * scale_num + 7) // 8)
    return out.to_numpy()


@pytest.mark.parametrize("scale_num", [4, 2, 1])
@pytest.mark.parametrize("luma_mode", [0, 1, 2])
def test_reduced_size_decode_committed_files(golden_dir, scale_num, luma_mode):
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    n_checked = 0
    for i, name in enumerate(z["names"]):
        j = O.jpeg_read_coefficients(z[f"jpg_{i}"].tobytes())
        if "4:2:2" in str(name):
            continue
        exp = O.jpeg_idct_color_scaled(j, scale_num, luma_mode)
        got = run_stage_scaled([j], scale_num, luma_mode)[0]
        assert np.array_equal(got, exp), (str(name), scale_num, luma_mode)
        key = f"rgb_{i}_s{scale_num}"
        if luma_mode == 0 and key in z.files:          # libjpeg's own reduced IDCT: equals Pillow's draft-mode decode
            oh, ow = z[key].shape[:2]
            assert np.array_equal(got[:, : 4 * ow].reshape(oh, ow, 4)[..., [2, 1, 0]], z[key]), str(name)
        n_checked += 1
    assert n_checked >= 25


@pytest.mark.parametrize("hs,vs,ncomp", [((2, 1, 1), (2, 1, 1), 3), ((1, 1, 1), (1, 1, 1), 3), ((1, 0, 0), (1, 0, 0), 1)])
def test_reduced_size_random_coefficients_and_4k(hs, vs, ncomp):
    rng = np.random.default_rng(17 + ncomp)
    for (w, h) in ((129, 67), (16, 8), (1, 1), (3840, 2160) if ncomp == 3 and hs[0] == 2 else (333, 100)):
        js = [_random_case(rng, w, h, hs, vs, ncomp, 200) for _ in range(2)]
        for scale_num, luma_mode in ((4, 2), (2, 1), (1, 2), (4, 0), (1, 0)):
            got = run_stage_scaled(js, scale_num, luma_mode)
            for k, j in enumerate(js):
                assert np.array_equal(got[k], O.jpeg_idct_color_scaled(j, scale_num, luma_mode)), (w, h, scale_num, luma_mode, k)


def test_reference_default_preshrink_path_cfg1():
    """BASELINE config 1 as the reference runs it: 3840x2160 `width=200` -> hints pick scale 1/8 with
    spatial sRGB luma -> 480x270, then Resample2D to 200x113."""
    from imageflow_amd.codecs.mozjpeg_decoder import apply_downscaling, idct_method_for_luma
    from imageflow_amd.graphics.bitmaps import Bitmap
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render
    from tests import util as U
    rng = np.random.default_rng(8)
    j = _random_case(rng, 3840, 2160, (2, 1, 1), (2, 1, 1), 3, 40)
    num, w, h = apply_downscaling(3840, 2160, 420, 236, 420, 236)
    assert (num, w, h) == (1, 480, 270) and idct_method_for_luma(num, True, True) == ("spatial_srgb", 1)
    st = JpegPixelStage(3840, 2160, 3, j["hs"], j["vs"], 1, DEV, scale_num=num, luma_spatial=True, luma_srgb=True)
    coef = [torch.from_numpy(j["coef"][c][None]).to(DEV) for c in range(3)]
    qt = torch.from_numpy(j["qt"][None].astype(np.int16)).to(DEV)
    decoded = st.read_frames(coef, qt)
    out = Bitmap.create_u8(1, 200, 113, DEV)
    scale_and_render(decoded, out, ScaleAndRenderParams(0, 0, 200, 113))
    torch.cuda.synchronize()
    exp_dec = O.jpeg_idct_color_scaled(j, 1, 2)
    assert np.array_equal(decoded.to_numpy()[0], exp_dec)
    exp = np.zeros((1, 113, U.stride_for(200)), np.uint8)
    U.oracle_render(exp_dec[None], 480, 270, exp, 200, 113, 0, 0, 200, 113)
    assert np.array_equal(out.to_numpy(), exp)


def test_unsupported_scales():
    for bad in (7, 0, 9):
        with pytest.raises(FlowError) as e:
            JpegPixelStage(64, 64, 3, (2, 1, 1), (2, 1, 1), 1, DEV, scale_num=bad)
        assert e.value.kind == ErrorKind.MethodNotImplemented
    st = JpegPixelStage(64, 64, 3, (2, 1, 1), (1, 1, 1), 1, DEV, scale_num=4)      # 4:2:2 reduced: chroma stays 4x4, h2v1 up-sampled
    assert (st.out_w, st.out_h) == (32, 32)


# ---- decode + resample as one call: component planes -> resampler (no decoded BGRA frame in HBM) -----------------------
def _decode_resample_case(j, n, scale_num, luma_mode, tw, th, compose="ReplaceSelf", matte=0, x=0, y=0, cw=None, ch=None):
    """-> (canvas bytes of the one-call form, canvas bytes of read_frames + scale_and_render, fused flag)."""
    from imageflow_amd.graphics.bitmaps import Bitmap, BitmapCompositing
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render
    st = JpegPixelStage(j["width"], j["height"], j["ncomp"], j["hs"], j["vs"], n, DEV, scale_num=scale_num,
                        luma_spatial=luma_mode != 0, luma_srgb=luma_mode == 2)
    coef = [torch.from_numpy(np.stack([j["coef"][c]] * n)).to(DEV) for c in range(3)]
    for c in range(j["ncomp"]):
        coef[c][1:] = torch.roll(coef[c][1:], 1, dims=2)                # frames of a batch differ
    qt = torch.from_numpy(np.stack([j["qt"]] * n).astype(np.int16)).to(DEV)
    cw, ch = cw or tw, ch or th
    info = ScaleAndRenderParams(x, y, tw, th)
    canv = [Bitmap.create_u8(n, cw, ch, DEV, compose=BitmapCompositing[compose], matte=matte) for _ in range(2)]
    for c in canv:
        c.data.fill_(0x5A)
    fused = st.read_frames_into(coef, qt, canv[0], info)
    decoded = st.read_frames(coef, qt)
    scale_and_render(decoded, canv[1], info)
    torch.cuda.synchronize()
    return canv[0].to_numpy(), canv[1].to_numpy(), fused, decoded


@pytest.mark.parametrize("scale_num,luma_mode", [(4, 0), (4, 1), (4, 2), (2, 2), (1, 0)])
def test_decode_resample_one_call_equals_the_two_calls_420(scale_num, luma_mode):
    """4:2:0 decoded at 1/8 .. 4/8: chroma runs the twice-larger IDCT, the three planes have the output size -> fused."""
    rng = np.random.default_rng(100 + scale_num * 3 + luma_mode)
    for (w, h, tw, th) in ((1001, 653, 211, 137), (640, 480, 100, 75), (2000, 1504, 333, 250)):
        j = _random_case(rng, w, h, (2, 1, 1), (2, 1, 1), 3, 300)
        ow, oh = -(-w * scale_num // 8), -(-h * scale_num // 8)
        tw, th = min(tw, ow), min(th, oh)
        one, two, fused, decoded = _decode_resample_case(j, 3, scale_num, luma_mode, tw, th)
        pitch = -(-w // 16) * 2 * scale_num                              # samples per plane row: MCUs x 2 luma blocks x n
        assert fused == (pitch % 4 == 0), (w, h, scale_num)             # 4-byte reads of 4 samples need a 4-aligned pitch
        assert np.array_equal(one, two), (w, h, scale_num, luma_mode)
        assert np.array_equal(decoded.to_numpy()[0], O.jpeg_idct_color_scaled(j, scale_num, luma_mode))     # and the chain is the oracle's


def test_decode_resample_one_call_444_sub_rect_and_matte():
    """4:4:4 at full size is fused too; the rect / compositing arguments behave as in scale_and_render."""
    rng = np.random.default_rng(7)
    j = _random_case(rng, 517, 389, (1, 1, 1), (1, 1, 1), 3, 250)
    one, two, fused, _ = _decode_resample_case(j, 2, 8, 0, 120, 90, x=16, y=9, cw=200, ch=120)
    assert fused and np.array_equal(one, two)
    assert np.all(one[:, 0, :64] == 0x5A)                                # outside the rect: untouched
    one, two, fused, _ = _decode_resample_case(j, 2, 8, 0, 120, 90, compose="BlendWithMatte", matte=0xFF336699)
    assert fused and np.array_equal(one, two)


def test_decode_resample_falls_back_where_planes_are_not_at_output_size():
    """Full-size 4:2:0 needs the fancy up-sampler, grayscale has one plane: same call, two-step chain inside, same bytes."""
    rng = np.random.default_rng(9)
    j = _random_case(rng, 640, 400, (2, 1, 1), (2, 1, 1), 3, 200)
    one, two, fused, _ = _decode_resample_case(j, 2, 8, 0, 160, 100)
    assert not fused and np.array_equal(one, two)
    g = _random_case(rng, 320, 200, (1, 0, 0), (1, 0, 0), 1, 200)
    one, two, fused, _ = _decode_resample_case(g, 2, 4, 0, 80, 50)
    assert not fused and np.array_equal(one, two)
    # an up-scale the fused resampler does not take (more than 8 live rows): planes at output size, chain nevertheless
    j2 = _random_case(rng, 160, 96, (2, 1, 1), (2, 1, 1), 3, 200)
    one, two, fused, _ = _decode_resample_case(j2, 2, 4, 0, 320, 192)
    assert np.array_equal(one, two)


def test_decode_resample_scratch_comes_from_the_block_cache_in_stream_order():
    """The two-step chain's scratch bitmap is a block of the library's cache released BEHIND the stream (devmem.cpp
    cached_free_after): no host wait, no driver call after the first, reused by the next call on the same stream while the
    previous one may still be running, on another stream only once the first has passed the release point.  (Until round 5 it
    was hipMallocAsync / hipFreeAsync, whose pool trimmed itself at synchronisations: the no-hint ABI job ran at 1 200 or
    2 800 jobs/s depending on the process.)"""
    from imageflow_amd import _native
    from imageflow_amd.graphics.bitmaps import Bitmap
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render
    rng = np.random.default_rng(19)
    j = _random_case(rng, 640, 400, (2, 1, 1), (2, 1, 1), 3, 200)
    n = 2
    st = JpegPixelStage(j["width"], j["height"], j["ncomp"], j["hs"], j["vs"], n, DEV, scale_num=8)
    coef = [torch.from_numpy(np.stack([j["coef"][c]] * n)).to(DEV) for c in range(3)]
    qt = torch.from_numpy(np.stack([j["qt"]] * n).astype(np.int16)).to(DEV)
    info = ScaleAndRenderParams(0, 0, 160, 100)
    want = Bitmap.create_u8(n, 160, 100, DEV)
    scale_and_render(st.read_frames(coef, qt), want, info)
    torch.cuda.synchronize()
    want = want.to_numpy()
    canv = [Bitmap.create_u8(n, 160, 100, DEV) for _ in range(12)]
    assert not st.read_frames_into(coef, qt, canv[0], info)              # (warm: the plan, the first scratch block)
    torch.cuda.synchronize()
    import gc
    gc.collect()                                                         # (earlier tests' stages and plans free their blocks now, not inside the window)
    s0 = _native.cache_stats()
    for c in canv[1:7]:                                                  # six calls queued back to back on one stream, no wait between
        assert not st.read_frames_into(coef, qt, c, info)
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):                                        # and six on another stream, concurrently
        for c in canv[7:]:
            assert not st.read_frames_into(coef, qt, c, info)
    torch.cuda.synchronize()
    s1 = _native.cache_stats()
    assert s1["device_hits"] - s0["device_hits"] >= 10                   # the scratch came from the cache ...
    assert s1["device_driver_allocs"] - s0["device_driver_allocs"] <= 2  # ... (the other stream may need a block of its own)
    assert (s1["device_driver_frees"], s1["device_wide_syncs"]) == (s0["device_driver_frees"], s0["device_wide_syncs"]), (s0, s1)
    for c in canv:
        assert np.array_equal(c.to_numpy(), want)
    _native.trim_cache()                                                 # blocks still parked behind a stream are cache like the rest
    assert _native.cache_stats()["device_bytes_cached"] == 0


def test_decode_resample_cfg4_shape():
    """BASELINE config 4 as the reference decodes it: 3840x2160 4:2:0 at 4/8 with the spatial sRGB luma scaler -> 800x450."""
    rng = np.random.default_rng(44)
    j = _random_case(rng, 3840, 2160, (2, 1, 1), (2, 1, 1), 3, 60)
    one, two, fused, _ = _decode_resample_case(j, 2, 4, 2, 800, 450)
    assert fused and np.array_equal(one, two)
    # ... and against the ORACLE chain directly (frame 0 of the batch carries j's coefficients unchanged): scaled decode with
    # the reference's compiled spatial luma scaler, then the CPU resize
    exp_dec = O.jpeg_idct_color_scaled(j, 4, 2)
    exp = np.zeros((1, 450, U.stride_for(800)), np.uint8)
    U.oracle_render(exp_dec[None], 1920, 1080, exp, 800, 450, 0, 0, 800, 450)
    assert np.array_equal(one[0], exp[0])


@pytest.mark.parametrize("scale_num,ow,oh,tw,th", [(1, 480, 270, 200, 113), (2, 960, 540, 200, 113)])
def test_decode_resample_one_call_at_4k_equals_the_oracle_chain(scale_num, ow, oh, tw, th):
    """The 1/8 and 2/8 one-call shapes of a 3840x2160 file (what `width=200` asks its decoder for, BASELINE config 1) against
    oracle decode + oracle resize, not only against the two-call form."""
    rng = np.random.default_rng(50 + scale_num)
    j = _random_case(rng, 3840, 2160, (2, 1, 1), (2, 1, 1), 3, 60)
    one, two, fused, _ = _decode_resample_case(j, 2, scale_num, 2, tw, th)
    assert fused and np.array_equal(one, two)
    exp_dec = O.jpeg_idct_color_scaled(j, scale_num, 2)
    exp = np.zeros((1, th, U.stride_for(tw)), np.uint8)
    U.oracle_render(exp_dec[None], ow, oh, exp, tw, th, 0, 0, tw, th)
    assert np.array_equal(one[0][:, :4 * tw], exp[0][:, :4 * tw])       # (the row padding keeps the canvases' 0x5A fill)


def test_decode_resample_rejects_a_plan_of_another_size():
    from imageflow_amd.graphics.bitmaps import Bitmap
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, plan_for
    from imageflow_amd.graphics.weights import Filter
    rng = np.random.default_rng(3)
    j = _random_case(rng, 320, 240, (2, 1, 1), (2, 1, 1), 3, 100)
    st = JpegPixelStage(320, 240, 3, j["hs"], j["vs"], 1, DEV, scale_num=4)
    coef = [torch.from_numpy(j["coef"][c][None]).to(DEV) for c in range(3)]
    qt = torch.from_numpy(j["qt"][None].astype(np.int16)).to(DEV)
    out = Bitmap.create_u8(1, 40, 30, DEV)
    with pytest.raises(FlowError) as e:
        st.read_frames_into(coef, qt, out, ScaleAndRenderParams(0, 0, 40, 30), plan=plan_for(320, 240, 40, 30, Filter.Robidoux, 0.0, torch.device(DEV)))
    assert e.value.kind == ErrorKind.InvalidArgument


@pytest.mark.parametrize("scale_num,luma_mode", [(8, 0), (4, 0), (4, 2), (2, 1)])
def test_outputs_beyond_the_range_limit_tables_clamp_region(scale_num, luma_mode):
    """IDCT outputs far outside [-384, 383]: libjpeg's range-limit table wraps there, so the block-per-lane kernels must
    take the table form, not the clamp -- mixed in one batch with tame blocks so that both wave-uniform choices occur.
    (Values stay below what overflows the 32-bit first pass: beyond that libjpeg's own 64-bit JLONG arithmetic and any
    32-bit restatement part ways, and no 8-bit file gets there.)"""
    rng = np.random.default_rng(77 + scale_num + luma_mode)
    for hs, vs in (((2, 1, 1), (2, 1, 1)), ((1, 1, 1), (1, 1, 1))):
        j = _random_case(rng, 200, 120, hs, vs, 3, 30)
        j["qt"] = np.ones((3, 64), np.uint16)
        for c in range(3):
            co = j["coef"][c]
            wild = rng.random(size=co.shape[:2]) < 0.3                     # 30 % of the blocks
            big = np.zeros(co.shape, np.int16)
            big[..., 0] = rng.integers(1500, 3000, size=co.shape[:2]) * rng.choice([-1, 1], size=co.shape[:2])
            for k in (1, 8, 9):
                big[..., k] = rng.integers(-1200, 1201, size=co.shape[:2])
            co[wild] = big[wild]
        got = run_stage_scaled([j, j], scale_num, luma_mode) if scale_num != 8 else run_stage([j, j])
        exp = O.jpeg_idct_color_scaled(j, scale_num, luma_mode) if scale_num != 8 else O.jpeg_idct_color(j)
        assert np.array_equal(got[0], exp) and np.array_equal(got[1], exp), (hs, scale_num, luma_mode)


def test_decode_resample_one_call_wide_source_strips_and_blend_with_self():
    """A source wider than one workgroup's strip (two column strips, XCD renumbering) and BlendWithSelf onto a canvas that
    already holds pixels, sRGB working space excluded: same bytes as the two calls."""
    rng = np.random.default_rng(12)
    j = _random_case(rng, 5000, 304, (1, 1, 1), (1, 1, 1), 3, 120)
    one, two, fused, _ = _decode_resample_case(j, 3, 8, 0, 700, 43, compose="BlendWithSelf")
    assert fused and np.array_equal(one, two)
    j = _random_case(rng, 4096, 2048, (2, 1, 1), (2, 1, 1), 3, 80)                    # 4/8 -> 2048x1024 -> 333x167: bands
    one, two, fused, _ = _decode_resample_case(j, 5, 4, 1, 333, 167)
    assert fused and np.array_equal(one, two)
