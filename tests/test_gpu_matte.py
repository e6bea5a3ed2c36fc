"""apply_matte (graphics/blend.rs:6-59) on the GPU vs the oracle, bit for bit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.graphics import blend  # noqa: E402
from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import util as U  # noqa: E402


@pytest.mark.parametrize("matte", [0xFFFFFFFF, 0xFF000000, 0x80336699, 0x00123456])
@pytest.mark.parametrize("shape", [(37, 5), (640, 360), (1, 1)])
def test_apply_matte_device(matte, shape):
    w, h = shape
    fr = U.random_frames(3, w, h, seed0=11)
    fr[0, :, 3:4 * w:16] = 0
    fr[0, :, 7:4 * w:16] = 255
    exp = fr.copy()
    for i in range(3):
        assert O.apply_matte(exp[i], w, h, fr.shape[2], matte) == 0
    b = Bitmap.from_numpy(fr.copy(), w, h, fr.shape[2], "cuda:0", alpha_meaningful=True)
    blend.apply_matte(b, matte)
    torch.cuda.synchronize()
    assert np.array_equal(b.to_numpy(), exp)
    assert b.alpha_meaningful == ((matte >> 24) != 255)


def test_apply_matte_noop_when_alpha_not_meaningful():
    fr = U.random_frames(1, 64, 8, seed0=2)
    b = Bitmap.from_numpy(fr.copy(), 64, 8, fr.shape[2], "cuda:0", alpha_meaningful=False)
    blend.apply_matte(b, 0xFFFFFFFF)
    torch.cuda.synchronize()
    assert np.array_equal(b.to_numpy(), fr)


def test_apply_matte_host_drop_in():
    fr = U.random_frames(1, 100, 40, seed0=3)[0]
    exp = fr.copy()
    O.apply_matte(exp, 100, 40, fr.shape[1], 0xFFFFFFFF)
    got = fr.copy()
    blend.apply_matte_host(got, 100, 40, fr.shape[1], 0xFFFFFFFF)
    assert np.array_equal(got, exp)


def test_apply_matte_full_4k_frame():
    fr = U.random_frames(1, 3840, 2160, seed0=8)
    exp = fr.copy()
    O.apply_matte(exp[0], 3840, 2160, fr.shape[2], 0xFFEEDDCC)
    b = Bitmap.from_numpy(fr.copy(), 3840, 2160, fr.shape[2], "cuda:0", alpha_meaningful=True)
    blend.apply_matte(b, 0xFFEEDDCC)
    torch.cuda.synchronize()
    assert np.array_equal(b.to_numpy(), exp)
