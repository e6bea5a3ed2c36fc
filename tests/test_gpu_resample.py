"""Parity of the HIP resample/render path (through the C ABI) against the CPU oracle, bit for bit.

BGRA8 surfaces: exact.  f32 working buffer: exact too (tolerance stated by north_star is 1 ULP; we assert 0).
Cases follow SURVEY.md section 8: cfg2/cfg5-shaped resizes at reduced size, ragged widths, x/y sub-rect render,
all three compositing modes, alpha meaningful or not, srgb/linear working space, sharpen, upscale (generic path),
plus full-size 4K frames and size-independent properties.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.errors import ErrorKind, FlowError  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap, BitmapCompositing  # noqa: E402
from imageflow_amd.graphics.color import WorkingFloatspace  # noqa: E402
from imageflow_amd.graphics.scaling import (ResamplePlan, ScaleAndRenderParams, scale_and_render,  # noqa: E402
                                            scale_and_render_host)
from imageflow_amd.graphics.weights import Filter  # noqa: E402
from tests import util as U  # noqa: E402

DEV = "cuda:0"


def run_case(in_w, in_h, out_w, out_h, *, n=2, filt=Filter.Robidoux, sharpen=0.0, space=WorkingFloatspace.LinearRGB,
             compose=BitmapCompositing.ReplaceSelf, matte=0, alpha=False, x=0, y=0, cw=None, ch=None, force=-1,
             frames=None, seed=0):
    cw = cw or out_w + x
    ch = ch or out_h + y
    if frames is None:
        frames = U.random_frames(n, in_w, in_h, seed0=1000 + seed, alpha=True)
    n = frames.shape[0]
    cst = U.stride_for(cw)
    rng = np.random.default_rng(77 + seed)
    canvas0 = rng.integers(0, 256, size=(n, ch, cst), dtype=np.uint8)
    exp = canvas0.copy()
    exp_f32 = U.oracle_render(frames, in_w, in_h, exp, cw, ch, x, y, out_w, out_h, filter_id=int(filt), sharpen=sharpen,
                              working_space=int(space), compositing=int(compose), matte_bgra=matte,
                              alpha_meaningful=alpha, want_f32=True)
    inp = Bitmap.from_numpy(frames, in_w, in_h, frames.shape[2], DEV, alpha_meaningful=alpha)
    # canvases and the f32 dump are the first n frames of buffers with one guard frame behind them: no kernel may store there
    can_all = torch.cat([torch.from_numpy(canvas0.reshape(n, -1)), torch.full((1, ch * cst), 0xA5, dtype=torch.uint8)]).to(DEV)
    can = Bitmap(can_all[:n], cw, ch, cst, False, compose, matte)
    f32_all = torch.zeros((n + 1, out_h, out_w, 4), dtype=torch.float32, device=DEV)
    f32_all[n] = 12345.0
    f32 = f32_all[:n]
    info = ScaleAndRenderParams(x, y, out_w, out_h, sharpen, filt, space)
    plan = scale_and_render(inp, can, info, f32_out=f32, force_kernel=force)
    torch.cuda.synchronize()
    assert bool((can_all[n] == 0xA5).all()) and bool((f32_all[n] == 12345.0).all()), "a store landed behind the last frame"
    got = can.to_numpy()
    got_f32 = f32.cpu().numpy()
    assert np.array_equal(got_f32.view(np.uint32), exp_f32.view(np.uint32)), \
        f"f32 working buffer differs: max abs {np.abs(got_f32 - exp_f32).max()}"
    if not np.array_equal(got, exp):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{len(bad)} bytes differ, first at {bad[0]}: got {got[tuple(bad[0])]} exp {exp[tuple(bad[0])]}")
    return plan


SHAPES = [
    (384, 216, 20, 20),      # cfg2 ratio (19.2x / 10.8x)
    (768, 432, 40, 23),      # cfg5-like ratio
    (101, 67, 33, 21),       # ragged, ~3x
    (37, 29, 5, 4),
    (64, 64, 63, 63),        # nearly 1:1
    (16, 16, 1, 1),
    (4, 4, 2, 2),
    (1000, 10, 100, 3),
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("alpha", [False, True])
def test_replace_self_linear_fused_and_generic(shape, alpha):
    iw, ih, ow, oh = shape
    p = run_case(iw, ih, ow, oh, alpha=alpha, force=-1)
    run_case(iw, ih, ow, oh, alpha=alpha, force=1)
    if p.kernel_kind() == 0:
        run_case(iw, ih, ow, oh, alpha=alpha, force=0)


@pytest.mark.parametrize("filt,sharpen", [(Filter.Lanczos, 15.0), (Filter.Ginseng, 0.0), (Filter.Hermite, 0.0),
                                          (Filter.Box, 0.0), (Filter.Triangle, 0.0), (Filter.CatmullRom, 5.0),
                                          (Filter.Mitchell, 0.0), (Filter.NCubic, 0.0), (Filter.Jinc, 0.0),
                                          (Filter.Robidoux, 50.0), (Filter.Fastest, 0.0), (Filter.LanczosSharp, 0.0)])
def test_filters(filt, sharpen):
    run_case(300, 170, 31, 18, filt=filt, sharpen=sharpen, alpha=True)
    run_case(300, 170, 31, 18, filt=filt, sharpen=sharpen, alpha=False, force=1)


@pytest.mark.parametrize("space", [WorkingFloatspace.StandardRGB, WorkingFloatspace.LinearRGB])
@pytest.mark.parametrize("compose", list(BitmapCompositing))
@pytest.mark.parametrize("alpha", [False, True])
def test_compositing_modes(space, compose, alpha):
    for matte in (0xFFFFFFFF, 0x80FF2010):
        run_case(200, 120, 23, 14, space=space, compose=compose, matte=matte, alpha=alpha)
        run_case(200, 120, 23, 14, space=space, compose=compose, matte=matte, alpha=alpha, force=1)


def test_subrect_render_leaves_rest_of_canvas_untouched():
    run_case(320, 200, 30, 17, x=5, y=3, cw=50, ch=40, alpha=True, compose=BitmapCompositing.BlendWithSelf)
    run_case(320, 200, 30, 17, x=5, y=3, cw=50, ch=40, alpha=False)
    run_case(320, 200, 30, 17, x=20, y=23, cw=50, ch=40, alpha=True, compose=BitmapCompositing.BlendWithMatte,
             matte=0xFF336699, force=1)


def test_upscale_uses_generic_path():
    p = run_case(20, 14, 57, 41, filt=Filter.Ginseng, alpha=True)
    assert p.kernel_kind() == 1
    run_case(64, 48, 128, 96, filt=Filter.Robidoux, alpha=False)
    run_case(10, 10, 10, 30, filt=Filter.Lanczos, alpha=True)


def test_alpha_edge_values():
    fr = U.random_frames(2, 128, 96, seed0=5)
    fr[0, :, 3::4] = 0                 # fully transparent frame
    fr[1, :48, 3::4] = 255
    fr[1, 48:, 3::4] = 0
    for compose in BitmapCompositing:
        run_case(128, 96, 16, 12, frames=fr, alpha=True, compose=compose, matte=0xFF102030)


def test_constant_and_extreme_frames():
    for val in (0, 255):
        fr = np.full((1, 90, U.stride_for(160)), val, np.uint8)
        run_case(160, 90, 16, 9, frames=fr, alpha=False)
        run_case(160, 90, 16, 9, frames=fr, alpha=True, filt=Filter.Lanczos, sharpen=15.0)


def test_batch_of_gradient_frames_cfg2_shape_reduced():
    fr = U.gradient_frames(8, 960, 540)
    run_case(960, 540, 50, 50, frames=fr, alpha=False)


def test_full_size_4k_frame_bit_exact():
    """BASELINE config 2 at full size: 3840x2160 -> 200x200 Robidoux linear, gradient + random frames."""
    fr = np.concatenate([U.gradient_frames(1, 3840, 2160, k0=3), U.random_frames(1, 3840, 2160, seed0=1000, alpha=False)])
    p = run_case(3840, 2160, 200, 200, frames=fr, alpha=False)
    assert p.kernel_kind() == 0
    fr = U.random_frames(1, 3840, 2160, seed0=1001, alpha=True)
    run_case(3840, 2160, 200, 113, frames=fr, alpha=True, compose=BitmapCompositing.BlendWithMatte, matte=0xFFFFFFFF)


def test_full_size_fused_equals_generic_checksum():
    """Size-independent property at BASELINE sizes: two independent kernel families agree on every byte."""
    n = 4
    fr = U.random_frames(n, 3840, 2160, seed0=2000, alpha=True)
    inp = Bitmap.from_numpy(fr, 3840, 2160, fr.shape[2], DEV, alpha_meaningful=True)
    info = ScaleAndRenderParams(0, 0, 200, 200)
    outs = []
    for force in (0, 1):
        can = Bitmap.create_u8(n, 200, 200, DEV)
        f32 = torch.zeros((n, 200, 200, 4), dtype=torch.float32, device=DEV)
        scale_and_render(inp, can, info, f32_out=f32, force_kernel=force)
        torch.cuda.synchronize()
        outs.append((can.to_numpy(), f32.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))


def test_8k_lanczos_sharpen_matte_cfg5_reduced_and_strips():
    """cfg5 shape: 7680 wide forces two column strips in the fused kernel."""
    fr = U.random_frames(1, 7680, 432, seed0=31, alpha=True)
    p = run_case(7680, 432, 400, 23, frames=fr, filt=Filter.Lanczos, sharpen=15.0, alpha=True,
                 compose=BitmapCompositing.BlendWithMatte, matte=0xFFFFFFFF)
    assert p.kernel_kind() == 0


def test_full_size_8k_cfg5_frame_with_f32_dump():
    """BASELINE config 5 at FULL size: 7680x4320 -> 400x225 Lanczos (window 3), sharpen_percent 15, alpha meaningful,
    flattened over a white matte in linear light; BGRA8 bit-exact and the whole f32 working buffer 0 ULP."""
    fr = U.random_frames(2, 7680, 4320, seed0=77, alpha=True)
    p = run_case(7680, 4320, 400, 225, frames=fr, filt=Filter.Lanczos, sharpen=15.0, alpha=True,
                 compose=BitmapCompositing.BlendWithMatte, matte=0xFFFFFFFF)
    assert p.kernel_kind(True) == 0


def test_full_size_cfg3_pyramid_four_frames():
    """BASELINE config 3 at full size with n = 4 frames: 3840x2160 -> 1600x900 -> {1200x675 -> 400x225, 800x450}, every
    level of every frame byte-equal to the oracle chain (levels feed each other on the device)."""
    n = 4
    fr = np.concatenate([U.gradient_frames(2, 3840, 2160, k0=11), U.random_frames(2, 3840, 2160, seed0=3100, alpha=False)])
    src = Bitmap.from_numpy(fr, 3840, 2160, fr.shape[2], DEV)
    sizes = {"1600": (1600, 900), "1200": (1200, 675), "800": (800, 450), "400": (400, 225)}
    dev = {k: Bitmap.create_u8(n, w, h, DEV) for k, (w, h) in sizes.items()}
    for a, b in (("src", "1600"), ("1600", "1200"), ("1600", "800"), ("1200", "400")):
        scale_and_render(src if a == "src" else dev[a], dev[b], ScaleAndRenderParams(0, 0, *sizes[b]))
    torch.cuda.synchronize()
    host = {"src": fr}
    for a, b in (("src", "1600"), ("1600", "1200"), ("1600", "800"), ("1200", "400")):
        w, h = sizes[b]
        iw, ih = (3840, 2160) if a == "src" else sizes[a]
        exp = np.zeros((n, h, U.stride_for(w)), np.uint8)
        U.oracle_render(host[a], iw, ih, exp, w, h, 0, 0, w, h)
        host[b] = exp
        assert np.array_equal(dev[b].to_numpy(), exp), b


def test_host_buffer_drop_in_matches_oracle():
    from oracle import oracle as O
    fr = U.random_frames(1, 333, 222, seed0=9)[0]
    cst = U.stride_for(60)
    canvas0 = np.random.default_rng(4).integers(0, 256, size=(50, cst), dtype=np.uint8)
    for compose, alpha in ((BitmapCompositing.ReplaceSelf, False), (BitmapCompositing.BlendWithSelf, True),
                           (BitmapCompositing.BlendWithMatte, True)):
        exp = canvas0.copy()
        rc, _ = O.scale_and_render(fr, 333, 222, exp, 60, 50, 7, 9, 40, 27, compositing=int(compose),
                                   matte_bgra=0xFF00FF00, alpha_meaningful=alpha)
        assert rc == 0
        got = canvas0.copy()
        scale_and_render_host(fr, 333, 222, fr.shape[1], alpha, got, 60, 50, cst,
                              ScaleAndRenderParams(7, 9, 40, 27), compose, 0xFF00FF00)
        assert np.array_equal(got, exp)


def test_error_kinds_match_reference():
    src = np.zeros((4, 64), np.uint8)
    dst = np.zeros((4, 64), np.uint8)
    with pytest.raises(FlowError) as e:      # scaling.rs:24-29
        scale_and_render_host(src, 4, 4, 64, False, dst, 4, 4, 64, ScaleAndRenderParams(2, 0, 4, 4))
    assert e.value.kind == ErrorKind.InvalidArgument
    with pytest.raises(FlowError) as e:
        ResamplePlan(10, 10, 0, 5)
    assert e.value.kind == ErrorKind.InvalidArgument
    with pytest.raises(FlowError) as e:
        ResamplePlan(10, 10, 5, 5, filter=77)
    assert e.value.kind == ErrorKind.InvalidArgument


def test_linearity_property_alpha_weights_partition_of_unity():
    """Size-independent property: resizing an opaque constant-colour frame returns that colour exactly for every
    filter whose weights sum to one in f32 (checked against the oracle, which has the same property test on CPU)."""
    for val in (1, 64, 200):
        fr = np.full((1, 2160, U.stride_for(3840)), val, np.uint8)
        inp = Bitmap.from_numpy(fr, 3840, 2160, fr.shape[2], DEV)
        can = Bitmap.create_u8(1, 200, 200, DEV)
        scale_and_render(inp, can, ScaleAndRenderParams(0, 0, 200, 200))
        torch.cuda.synchronize()
        px = can.to_numpy()[0, :, :800].reshape(200, 200, 4)
        assert np.all(px[..., 3] == 255)
        assert np.all(np.abs(px[..., :3].astype(int) - val) <= 1)


def test_very_wide_and_very_tall_frames():
    """Maximum-size edge cases: many column strips (fused) and a tall thin frame."""
    fr = U.random_frames(1, 20000, 40, seed0=41, alpha=True)
    p = run_case(20000, 40, 1000, 8, frames=fr, alpha=False)
    assert p.kernel_kind(False) == 0
    run_case(20000, 40, 1000, 8, frames=fr, alpha=True, compose=BitmapCompositing.BlendWithSelf)
    fr = U.random_frames(1, 12, 30000, seed0=42, alpha=True)
    run_case(12, 30000, 3, 2000, frames=fr, alpha=True)


def test_empty_batch_is_a_no_op():
    inp = Bitmap.create_u8(0, 64, 64, DEV)
    can = Bitmap.create_u8(0, 8, 8, DEV)
    scale_and_render(inp, can, ScaleAndRenderParams(0, 0, 8, 8))
    torch.cuda.synchronize()


def test_every_ring_size_and_shape_variant():
    """Vertical ratios chosen so that the ring holds 1..8 live rows (all fused kernel instantiations, both alpha forms)."""
    seen = set()
    for (ih, oh, filt) in ((64, 64, Filter.Box), (200, 100, Filter.Box), (120, 64, Filter.Triangle), (300, 100, Filter.Hermite),
                           (400, 37, Filter.Robidoux), (400, 90, Filter.Robidoux), (500, 45, Filter.Lanczos),
                           (330, 200, Filter.Lanczos), (640, 48, Filter.Ginseng), (512, 100, Filter.NCubic), (600, 520, Filter.Jinc)):
        for alpha in (False, True):
            p = run_case(96, ih, 17, oh, filt=filt, alpha=alpha, n=1, seed=ih + oh)
            seen.add((p.kernel_kind(alpha)))
    assert 0 in seen


@pytest.mark.parametrize("case", [(1600, 90, 600, 34, 3), (1600, 90, 1200, 68, 5), (480, 135, 200, 57, 11), (480, 135, 20, 6, 11),
                                  (800, 60, 333, 25, 7), (1920, 108, 800, 45, 3)])
@pytest.mark.parametrize("alpha", [False, True])
def test_several_frames_per_workgroup(case, alpha):
    """Sources narrower than half a workgroup share one workgroup (F frames side by side, csrc/api.cpp); batch sizes that
    are not a multiple of F leave idle frame slots in the last workgroup."""
    iw, ih, ow, oh, n = case
    run_case(iw, ih, ow, oh, n=n, alpha=alpha, compose=BitmapCompositing.BlendWithSelf if alpha else BitmapCompositing.ReplaceSelf)
    run_case(iw, ih, ow, oh, n=1, alpha=alpha, x=3, y=2, cw=ow + 5, ch=oh + 4)


def test_host_drop_in_from_several_threads():
    """imageflow runs one Context per thread (imageflow_abi/src/lib.rs:20-27): the host-buffer entry point keeps a stream
    and staging buffers per calling thread, so concurrent callers neither serialise on the null stream nor see each
    other's pixels.  8 threads x 6 calls of different shapes and modes, every result equal to the oracle."""
    import threading
    from oracle import oracle as O
    shapes = [(640, 360, 64, 36), (333, 222, 40, 27), (1920, 1080, 200, 113), (96, 64, 96, 64)]
    errors = []

    def worker(k):
        try:
            rng = np.random.default_rng(900 + k)
            for i in range(6):
                iw, ih, ow, oh = shapes[(k + i) % len(shapes)]
                alpha = (k + i) % 2 == 1
                compose = BitmapCompositing.BlendWithMatte if alpha and i % 3 == 0 else BitmapCompositing.ReplaceSelf
                fr = U.random_frames(1, iw, ih, seed0=5000 + 10 * k + i, alpha=True)[0]
                cst = U.stride_for(ow)
                canvas0 = rng.integers(0, 256, size=(oh, cst), dtype=np.uint8)
                exp = canvas0.copy()
                rc, _ = O.scale_and_render(fr, iw, ih, exp, ow, oh, 0, 0, ow, oh, compositing=int(compose), matte_bgra=0xFF336699,
                                           alpha_meaningful=alpha)
                assert rc == 0
                got = canvas0.copy()
                scale_and_render_host(fr, iw, ih, fr.shape[1], alpha, got, ow, oh, cst, ScaleAndRenderParams(0, 0, ow, oh), compose, 0xFF336699)
                if not np.array_equal(got[:, :4 * ow], exp[:, :4 * ow]):
                    errors.append((k, i, "pixels differ"))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# ---- banded two-pass kernel (force_kernel 2): the generic pair fused through LDS, for up-scales and small frames ------------
BANDED_SHAPES = [
    (100, 100, 300, 300),    # the reference's test_trim_then_resize shape: 15 live output rows per source row
    (200, 200, 400, 400),    # test_fill_rect's shape
    (20, 14, 57, 41), (64, 48, 128, 96), (10, 10, 10, 30), (37, 23, 111, 70),
    (1, 1, 7, 5), (250, 3, 1000, 9),
    (101, 67, 33, 21), (16, 16, 1, 1), (4, 4, 2, 2), (130, 90, 129, 91),      # small down-scales and ~1:1 fit as well
]


@pytest.mark.parametrize("shape", BANDED_SHAPES)
@pytest.mark.parametrize("alpha", [False, True])
def test_banded_kernel_equals_the_oracle(shape, alpha):
    iw, ih, ow, oh = shape
    run_case(iw, ih, ow, oh, alpha=alpha, force=2, n=3)


@pytest.mark.parametrize("filt,sharpen", [(Filter.Lanczos, 15.0), (Filter.Ginseng, 0.0), (Filter.Hermite, 0.0), (Filter.Box, 0.0),
                                          (Filter.Jinc, 0.0), (Filter.Robidoux, 50.0), (Filter.CatmullRom, 5.0)])
def test_banded_kernel_filters(filt, sharpen, debug_switch):
    run_case(60, 40, 171, 113, filt=filt, sharpen=sharpen, alpha=True, force=2)
    run_case(150, 85, 31, 18, filt=filt, sharpen=sharpen, alpha=False, force=2)
    debug_switch("banded_flags", "0")                                               # no shortcut at all: tables from HBM, band rows by search
    run_case(60, 40, 171, 113, filt=filt, sharpen=sharpen, alpha=True, force=2)


@pytest.mark.parametrize("space", [WorkingFloatspace.StandardRGB, WorkingFloatspace.LinearRGB])
@pytest.mark.parametrize("compose", list(BitmapCompositing))
@pytest.mark.parametrize("alpha", [False, True])
def test_banded_kernel_compositing_modes(space, compose, alpha):
    for matte in (0xFFFFFFFF, 0x80FF2010):
        run_case(40, 24, 93, 57, space=space, compose=compose, matte=matte, alpha=alpha, force=2, x=3, y=5, cw=100, ch=70)


def test_banded_kernel_frame_loop_and_wide_rows(debug_switch):
    run_case(600, 40, 1200, 80, alpha=True, force=2)           # 4 800 source pixels per band: no prefetch registers, tables from HBM
    debug_switch("banded_wgs", "1")                      # one workgroup per band takes every frame: the prefetch of frame i + 1
    run_case(100, 100, 300, 300, alpha=True, force=2, n=5)     # under the passes of frame i
    run_case(600, 40, 1200, 80, alpha=False, force=2, n=3)
    debug_switch("banded_wgs", "30")                     # 13 bands: two workgroups per band, frames 0 2 4 / 1 3
    run_case(100, 100, 300, 300, alpha=False, force=2, n=5, compose=BitmapCompositing.BlendWithMatte, matte=0xFF405060)


@pytest.mark.parametrize("strip", [16, 50, 64])
@pytest.mark.parametrize("shape", [(100, 100, 300, 300), (64, 48, 128, 96), (37, 23, 111, 70), (250, 3, 1000, 9), (130, 90, 129, 91),
                                   (101, 67, 33, 21)])
def test_banded_kernel_column_strips_forced_on_small_frames(shape, strip, debug_switch):
    """Column strips (wide frames) on shapes the suite can afford: the hook forces a strip width, every strip stages its own
    source columns and its slice of the horizontal tables."""
    iw, ih, ow, oh = shape
    debug_switch("banded_strip", str(strip))
    run_case(iw, ih, ow, oh, alpha=True, force=2, n=3)
    run_case(iw, ih, ow, oh, alpha=False, force=2, n=2, filt=Filter.Ginseng)
    debug_switch("banded_flags", "0")                          # tables from HBM, band rows by search
    run_case(iw, ih, ow, oh, alpha=True, force=2, n=2, filt=Filter.Lanczos, sharpen=15.0)


@pytest.mark.parametrize("case", [(960, 54, 1920, 108, Filter.Ginseng), (1280, 40, 1920, 60, Filter.Ginseng), (700, 20, 2100, 60, Filter.Robidoux),
                                  (1500, 30, 1700, 34, Filter.Lanczos), (1100, 16, 4400, 64, Filter.Hermite)])
@pytest.mark.parametrize("alpha", [False, True])
def test_wide_up_scales_take_the_banded_kernel_in_column_strips(case, alpha):
    """An HD frame up-scaled with the node's default up filter (scale_render.rs:255-259): too many live rows for the fused
    kernel, rows too wide for a band of whole rows -- auto mode cuts bands into column strips."""
    iw, ih, ow, oh, filt = case
    plan = run_case(iw, ih, ow, oh, alpha=alpha, filt=filt, n=2)
    run_case(iw, ih, ow, oh, alpha=alpha, filt=filt, n=3, force=2, compose=BitmapCompositing.BlendWithSelf, x=5, y=3, cw=ow + 9, ch=oh + 4)
    run_case(iw, ih, ow, oh, alpha=alpha, filt=filt, n=1, force=2, compose=BitmapCompositing.BlendWithMatte, matte=0x80FF2010,
             space=WorkingFloatspace.StandardRGB)


@pytest.mark.parametrize("case", [(1440, 36, 2880, 72, Filter.Robidoux), (1920, 30, 2400, 38, Filter.Hermite), (1600, 28, 2400, 42, Filter.Box),
                                  (1920, 24, 3840, 48, Filter.Triangle)])
def test_wide_alpha_up_scales_the_fused_kernel_could_take_go_to_the_banded_kernel(case):
    """Auto mode, alpha, a source of 1 440 - 2 048 columns up-scaled by 1.25 or more: the fused kernel's geometry would be one
    workgroup of at most four waves per CU, the banded kernel's column strips are 1.1 - 2.9 x faster there (api.cpp); whichever
    runs, the pixels are the oracle's -- in every compositing mode."""
    iw, ih, ow, oh, filt = case
    run_case(iw, ih, ow, oh, alpha=True, filt=filt, n=3)
    run_case(iw, ih, ow, oh, alpha=True, filt=filt, n=2, compose=BitmapCompositing.BlendWithSelf, x=7, y=2, cw=ow + 8, ch=oh + 5)
    run_case(iw, ih, ow, oh, alpha=True, filt=filt, n=2, compose=BitmapCompositing.BlendWithMatte, matte=0xFF102030, force=0)    # and the fused kernel still agrees
    run_case(iw, ih, ow, oh, alpha=False, filt=filt, n=2)


def test_banded_kernel_refuses_what_does_not_fit():
    with pytest.raises(FlowError) as e:
        run_case(3840, 216, 200, 20, n=1, force=2)            # 76 source rows x 3840 columns x 16 bytes per band of one row
    assert e.value.kind == ErrorKind.InvalidState


def test_auto_mode_picks_the_banded_kernel_where_the_fused_one_does_not_apply():
    run_case(100, 100, 300, 300, alpha=True)                  # 3x up-scale: too many live rows for the fused kernel -> banded
    run_case(200, 200, 400, 400, alpha=True, filt=Filter.Hermite)     # fused
    run_case(384, 216, 20, 20, alpha=False)                   # a down-scale the fused kernel takes
    run_case(640, 360, 7, 4, alpha=True)                      # too many live rows for the fused kernel, too many source rows per band: generic pair


@pytest.mark.parametrize("case", [(1600, 90, 1200, 68, Filter.Robidoux, 4), (400, 300, 300, 225, Filter.Robidoux, 4),
                                  (640, 48, 533, 40, Filter.Hermite, 2), (200, 120, 400, 240, Filter.Hermite, 2),
                                  (8000, 24, 6000, 18, Filter.Robidoux, 4), (333, 100, 250, 75, Filter.CatmullRom, 4)])
def test_two_column_groups_of_the_fast_horizontal_pass(case):
    """Windows of a few taps (ratios below ~1.6, up-scales): the fast horizontal pass runs groups of TWO source columns where
    that computes at most two thirds of the taps per output (csrc/api.cpp, resample_fused.hip FG >= 16) -- BGRA sources without
    alpha; same taps in the same order, so the oracle's bytes and f32 values, also in a sub-rectangle of the canvas, over
    several column strips (8000 wide) and with several frames per workgroup.  With alpha the four-column form runs."""
    iw, ih, ow, oh, filt, g2 = case
    p = run_case(iw, ih, ow, oh, filt=filt, alpha=False, n=3, seed=iw + ow)
    four, two = p.horizontal_groups()
    assert p.kernel_kind(False) == 0 and two == g2 and 3 * two <= 4 * four, (four, two)
    run_case(iw, ih, ow, oh, filt=filt, alpha=False, n=2, seed=iw, x=3, y=2, cw=ow + 9, ch=oh + 5,
             space=WorkingFloatspace.StandardRGB)
    run_case(iw, ih, ow, oh, filt=filt, alpha=True, n=2, seed=ow, compose=BitmapCompositing.BlendWithSelf)


def test_two_column_groups_only_where_they_save_a_third_of_the_taps():
    assert ResamplePlan(1600, 90, 1200, 68, Filter.Robidoux, 0.0).horizontal_groups() == (3, 4)      # 12 taps or 8
    assert ResamplePlan(1200, 68, 400, 23, Filter.Robidoux, 0.0).horizontal_groups() == (4, 0)       # 16 or 12: stays
    assert ResamplePlan(3840, 216, 1600, 90, Filter.Robidoux, 0.0).horizontal_groups() == (3, 0)     # 12 or 12


@pytest.mark.parametrize("budget", [248, 200, 17, 1])
def test_cu_budget_changes_the_launch_geometry_not_the_pixels(budget):
    """ifhip_set_cu_budget makes choose_bands plan for fewer CUs (finer bands, more halo rows re-read): BGRA8 and the f32 working
    buffer stay the oracle's, for a thumbnail shape, a moderate ratio with several frames per workgroup and a strip-split source."""
    from imageflow_amd import _native
    _native.set_cu_budget(budget)
    try:
        run_case(960, 540, 50, 50, n=5, seed=budget)
        run_case(400, 225, 300, 169, n=7, alpha=True, compose=BitmapCompositing.BlendWithMatte, matte=0xFF203040, seed=budget + 1)
        run_case(4400, 120, 900, 25, n=2, seed=budget + 2)
    finally:
        _native.set_cu_budget(0)
