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


def test_block_cache_stats_trim_and_limits():
    """csrc/devmem.cpp through the C ABI: blocks of destroyed plans are recycled (hits grow, bytes stay cached),
    ifhip_cache_trim gives them back to the driver, a zero limit stops recycling."""
    import imageflow_amd
    from imageflow_amd import _native
    from imageflow_amd.graphics.scaling import ResamplePlan
    from imageflow_amd.graphics.weights import Filter
    torch.cuda.synchronize()
    _native.trim_cache()
    s0 = _native.cache_stats()
    for _ in range(3):
        p = ResamplePlan(1000, 700, 100, 70, Filter.Robidoux, 0.0)
        del p
    s1 = _native.cache_stats()
    assert s1["device_hits"] > s0["device_hits"] and s1["device_bytes_cached"] > 0
    assert s1["device_limit_bytes"] == 8 << 30 and s1["host_limit_bytes"] == 1 << 30
    dev_freed, _ = imageflow_amd.trim_cache()
    s2 = _native.cache_stats()
    assert dev_freed == s1["device_bytes_cached"] and s2["device_bytes_cached"] == 0
    assert s2["device_driver_frees"] > s1["device_driver_frees"]
    try:
        _native.set_cache_limits(0, 0)
        p = ResamplePlan(1000, 700, 100, 70, Filter.Robidoux, 0.0)
        del p
        assert _native.cache_stats()["device_bytes_cached"] == 0
    finally:
        _native.set_cache_limits(8 << 30, 1 << 30)
