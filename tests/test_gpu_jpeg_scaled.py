"""GPU JPEG pixel stage at every scale MzDec::apply_downscaling can ask for (1..6, 8; mozjpeg_decoder.rs:603-617), all
samplings: byte equal to what libjpeg-turbo decodes (tests/golden/jpeg_scaled_cases.npz) through the C ABI, and -- with
the reference's luma selector (codec_jpeg_wrapper.c:274-343) -- equal to the oracle, whose block scalers are pinned to
the reference's own compiled functions."""
import io
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs import mozjpeg_decoder as D  # noqa: E402
from imageflow_amd.errors import FlowError  # noqa: E402
from oracle import oracle as O  # noqa: E402


def cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_scaled_cases.npz"))
    for i, name in enumerate(z["names"]):
        yield i, str(name), z[f"jpg_{i}"].tobytes(), z


def test_host_entry_point_equals_libjpeg_at_every_scale(golden_dir):
    n = 0
    for i, name, data, z in cases(golden_dir):
        j = O.jpeg_read_coefficients(data)
        for s in (int(v) for v in z["scales"]):
            ref = z[f"ref_{i}_{s}"]
            oh, ow = ref.shape[:2]
            got = D.jpeg_idct_color_host(j["coef"], j["qt"], j["ncomp"], j["hs"], j["vs"], j["width"], j["height"], scale_num=s)
            px = got[:, :4 * ow].reshape(oh, ow, 4)
            assert got.shape[0] == oh and np.array_equal(px[..., [2, 1, 0]], ref), (name, s)
            n += 1
    assert n == 66 * 7


@pytest.mark.parametrize("scale", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("luma_mode", [1, 2])
def test_luma_selector_at_every_scale_equals_the_oracle(golden_dir, scale, luma_mode):
    for i, name, data, z in cases(golden_dir):
        if i % 3:
            continue
        j = O.jpeg_read_coefficients(data)
        exp = O.jpeg_idct_color_scaled(j, scale, luma_mode)
        got = D.jpeg_idct_color_host(j["coef"], j["qt"], j["ncomp"], j["hs"], j["vs"], j["width"], j["height"], scale_num=scale,
                                     luma_spatial=True, luma_srgb=luma_mode == 2)
        ow = (j["width"] * scale + 7) // 8
        assert np.array_equal(got[:, :4 * ow], exp[:, :4 * ow]), (name, scale, luma_mode)


@pytest.mark.parametrize("scale,subsampling", [(3, "4:2:0"), (5, "4:2:0"), (6, "4:2:0"), (3, "4:2:2"), (6, "4:2:2"), (5, "4:4:4"), (2, "4:2:2")])
def test_device_batch_from_files(scale, subsampling):
    """whole chain on the device: GPU Huffman decode -> scaled pixel stage, a batch of 3 files 1000x700"""
    PIL = pytest.importorskip("PIL.Image")
    files = []
    for k in range(3):
        rng = np.random.default_rng(50 + k)
        y, x = np.mgrid[0:700, 0:1000]
        img = np.stack([(x + 3 * k) % 256, (y * 2) % 256, (x + y) % 256], -1).astype(np.int16) + rng.integers(-20, 21, (700, 1000, 3))
        b = io.BytesIO()
        PIL.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=85, subsampling=subsampling)
        files.append(b.getvalue())
    frames = D.decode_frames(files, scale_num=scale, luma_spatial=True, luma_srgb=True).to_numpy()
    for k, data in enumerate(files):
        exp = O.jpeg_idct_color_scaled(O.jpeg_read_coefficients(data), scale, 2)
        assert np.array_equal(frames[k], exp), (k, scale, subsampling)


def test_seven_eighths_is_refused():
    with pytest.raises(FlowError) as e:
        D.JpegPixelStage(64, 64, 3, [2, 1, 1], [2, 1, 1], 1, scale_num=7)
    assert "MethodNotImplemented" in str(e.value)
