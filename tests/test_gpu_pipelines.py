"""BASELINE configs 3 and 4 as device-resident chains, bit-exact against the same chain run through the oracle.
cfg3: export_4_sizes pyramid (imageflow_tool/src/self_test.rs:185-198): src -> 1600x900 -> {1200x675 -> 400x225, 800x450}.
cfg4: JPEG pixel stage (4:2:0) -> resample to 800 px wide."""
import io
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs.mozjpeg_decoder import JpegPixelStage  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import util as U  # noqa: E402

DEV = "cuda:0"


def _oracle_resize(frames, w, h, ow, oh):
    out = np.zeros((frames.shape[0], oh, U.stride_for(ow)), np.uint8)
    U.oracle_render(frames, w, h, out, ow, oh, 0, 0, ow, oh)
    return out


def _gpu_resize(b, ow, oh):
    out = Bitmap.create_u8(b.n, ow, oh, DEV)
    scale_and_render(b, out, ScaleAndRenderParams(0, 0, ow, oh))
    return out


@pytest.mark.parametrize("src", [(960, 540, 2, 4), (3840, 2160, 1, 1)])
def test_export_4_sizes_pyramid(src):
    w, h, n, div = src
    sizes = [(1600 // div, 900 // div), (1200 // div, 675 // div), (800 // div, 450 // div), (400 // div, 225 // div)]
    frames = U.gradient_frames(n, w, h, k0=11)
    frames[-1] = U.random_frames(1, w, h, seed0=77, alpha=False)[0]
    g0 = Bitmap.from_numpy(frames, w, h, frames.shape[2], DEV)
    g1 = _gpu_resize(g0, *sizes[0])
    g2 = _gpu_resize(g1, *sizes[1])
    g3 = _gpu_resize(g1, *sizes[2])
    g4 = _gpu_resize(g2, *sizes[3])
    torch.cuda.synchronize()
    o1 = _oracle_resize(frames, w, h, *sizes[0])
    o2 = _oracle_resize(o1, *sizes[0], *sizes[1])
    o3 = _oracle_resize(o1, *sizes[0], *sizes[2])
    o4 = _oracle_resize(o2, *sizes[1], *sizes[3])
    for g, o, name in ((g1, o1, "1600"), (g2, o2, "1200"), (g3, o3, "800"), (g4, o4, "400")):
        assert np.array_equal(g.to_numpy(), o), name


def test_jpeg_decode_then_resize_to_800(golden_dir):
    PIL = pytest.importorskip("PIL.Image")
    w, h = 1920, 1080
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([x * 255 // (w - 1), y * 255 // (h - 1), (x + y) * 255 // (w + h - 2)], -1).astype(np.uint8)
    img[::7, ::5] ^= 0x55
    buf = io.BytesIO()
    PIL.fromarray(img).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
    j = O.jpeg_read_coefficients(buf.getvalue())
    full = O.jpeg_idct_color(j)
    exp = _oracle_resize(full[None], w, h, 800, 450)
    st = JpegPixelStage(w, h, 3, j["hs"], j["vs"], 1, DEV)
    coef = [torch.from_numpy(j["coef"][c][None]).to(DEV) for c in range(3)]
    qt = torch.from_numpy(j["qt"][None].astype(np.int16)).to(DEV)
    decoded = st.read_frames(coef, qt)
    small = _gpu_resize(decoded, 800, 450)
    torch.cuda.synchronize()
    assert np.array_equal(decoded.to_numpy()[0], full)
    assert np.array_equal(small.to_numpy(), exp)
    ref = np.asarray(PIL.open(io.BytesIO(buf.getvalue())).convert("RGB"))      # and the decode equals libjpeg-turbo's
    assert np.array_equal(decoded.to_numpy()[0][:, : 4 * w].reshape(h, w, 4)[..., [2, 1, 0]], ref)
