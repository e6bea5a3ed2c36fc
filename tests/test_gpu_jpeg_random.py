"""Randomised end-to-end check of the JPEG decode path on the GPU against libjpeg-turbo itself: files written by
Pillow on the spot (random size, quality, chroma sub-sampling, restart interval, optimised or standard tables, colour
or grayscale) go through the entropy stage + pixel stage (`decode_frames`) and must equal Pillow's own decode byte for
byte; the forward stage is swept the same way against the oracle.  Skipped where Pillow is not installed."""
import io

import numpy as np
import pytest

torch = pytest.importorskip("torch")
PIL = pytest.importorskip("PIL.Image")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs import mozjpeg as MJ  # noqa: E402
from imageflow_amd.codecs import mozjpeg_decoder as D  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda:0"


def picture(rng, w, h, gray):
    y, x = np.mgrid[0:h, 0:w]
    kind = rng.integers(0, 4)
    if kind == 0:
        a = rng.integers(0, 256, size=(h, w, 3))
    elif kind == 1:
        a = np.stack([x * 255 // max(w - 1, 1), y * 255 // max(h - 1, 1), (x + y) * 255 // max(w + h - 2, 1)], -1)
    elif kind == 2:
        a = np.stack([128 + 100 * np.sin(x / 7.0), 128 + 100 * np.cos(y / 5.0), 128 + 80 * np.sin((x + y) / 11.0)], -1) \
            + rng.integers(-25, 26, size=(h, w, 3))
    else:
        a = np.where(((x // 5 + y // 3) % 2)[..., None] == 0, rng.integers(0, 256, 3), rng.integers(0, 256, 3))
    a = np.clip(a, 0, 255).astype(np.uint8)
    return a[..., 0] if gray else a


@pytest.mark.parametrize("block", range(6))
def test_files_decode_like_libjpeg_turbo(block):
    rng = np.random.default_rng(9000 + block)
    for _ in range(12):
        w, h = int(rng.integers(1, 420)), int(rng.integers(1, 300))
        gray = rng.random() < 0.15
        kw = dict(quality=int(rng.integers(3, 101)), optimize=bool(rng.integers(0, 2)))
        if not gray:
            kw["subsampling"] = ["4:4:4", "4:2:2", "4:2:0"][int(rng.integers(0, 3))]
        r = rng.integers(0, 3)
        if r == 1:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        elif r == 2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        n = int(rng.integers(1, 4))
        files, refs = [], []
        for _k in range(n):
            pic = PIL.fromarray(picture(rng, w, h, gray))
            buf = io.BytesIO()
            try:
                pic.save(buf, "JPEG", **kw)
            except OSError:                      # libjpeg "Suspension not allowed here": optimize + restarts on some sizes
                kw["optimize"] = False
                buf = io.BytesIO()
                pic.save(buf, "JPEG", **kw)
            files.append(buf.getvalue())
            refs.append(np.asarray(PIL.open(io.BytesIO(files[-1])).convert("RGB")))
        frames = D.decode_frames(files, DEV).to_numpy()
        for k in range(n):
            px = frames[k][:, :4 * w].reshape(h, w, 4)
            assert np.array_equal(px[..., [2, 1, 0]], refs[k]), (w, h, gray, kw, k)
            assert np.all(px[..., 3] == 255)


@pytest.mark.parametrize("block", range(3))
def test_forward_stage_random_sweep(block):
    rng = np.random.default_rng(7000 + block)
    for _ in range(15):
        w, h, n = int(rng.integers(1, 500)), int(rng.integers(1, 300)), int(rng.integers(1, 4))
        hs, vs = [([1, 1, 1], [1, 1, 1]), ([2, 1, 1], [1, 1, 1]), ([2, 1, 1], [2, 1, 1])][int(rng.integers(0, 3))]
        stride = O.stride_for_width(w) + 4 * int(rng.integers(0, 3))
        frames = rng.integers(0, 256, size=(n, h, stride), dtype=np.uint8)
        qts = np.stack([MJ.quant_tables_for_quality(int(rng.integers(1, 101))) for _ in range(n)])
        coef = MJ.JpegForwardStage(w, h, hs, vs, n).write_frames(Bitmap.from_numpy(frames, w, h, stride, DEV),
                                                                torch.from_numpy(qts.view(np.int16)).to(DEV))
        for k in range(n):
            ref = O.jpeg_forward(frames[k], w, h, stride, hs, vs, qts[k])
            for c in range(3):
                assert np.array_equal(coef[c][k].cpu().numpy(), ref[c]), (w, h, hs, vs, k, c)
