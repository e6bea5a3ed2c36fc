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
        JpegPixelStage(16, 16, 3, (1, 1, 1), (2, 1, 1), 1, DEV)          # h1v2 not implemented
    assert e.value.kind == ErrorKind.MethodNotImplemented
    with pytest.raises(FlowError) as e:
        JpegPixelStage(0, 16, 3, (2, 1, 1), (2, 1, 1), 1, DEV)
    assert e.value.kind == ErrorKind.InvalidArgument
    with pytest.raises(FlowError) as e:
        JpegPixelStage(16, 16, 4, (1, 1, 1), (1, 1, 1), 1, DEV)          # CMYK stays on the CPU path
    assert e.value.kind == ErrorKind.MethodNotImplemented
