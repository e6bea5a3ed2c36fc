"""GPU parity of the device entropy coder (csrc/jpeg_encode.hip, ifhip_jpeg_encode_batch_device): the files it leaves in
HBM are byte-identical to libjpeg-turbo's (Pillow wrote the file, the oracle's entropy decoder gave the coefficients back)
and to the host writer's (ifhip_jpeg_write_baseline) on the coefficients of the forward stage."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from imageflow_amd.codecs import mozjpeg as M
from imageflow_amd.graphics.bitmaps import Bitmap
from tests.test_jpeg_device_coder import SAMPLINGS, photo, pillow_file

pytestmark = pytest.mark.gpu


def planes_of(js):
    """cuda planes [n, bh, bw, 64] of n files of one geometry"""
    ncomp = js[0]["ncomp"]
    return [torch.from_numpy(np.stack([np.ascontiguousarray(j["coef"][c], np.int16) for j in js])).cuda() for c in range(ncomp)]


def coder_for(j, n, **kw):
    return M.JpegEntropyStage(j["width"], j["height"], j["hs"], j["vs"], j["bw"][:j["ncomp"]], j["bh"][:j["ncomp"]], n, **kw)


@pytest.mark.parametrize("sampling", ["4:2:0", "4:2:2", "4:4:4"])
@pytest.mark.parametrize("size", [(1, 1), (17, 9), (64, 48), (203, 131), (640, 360), (1600, 900)])
def test_files_equal_libjpeg_turbos(sampling, size):
    w, h = size
    qualities = (5, 75, 90, 100)
    datas = [pillow_file(photo(w, h, w * 31 + h + q), q, sampling) for q in qualities]
    js = [O.jpeg_read_coefficients(d) for d in datas]
    assert (js[0]["hs"], js[0]["vs"]) == SAMPLINGS[sampling]
    coder = coder_for(js[0], 1)
    for j, q, data in zip(js, qualities, datas):                      # (one quality per call: the marker segments carry the tables)
        files, status = coder.encode(planes_of([j]), q)
        assert status == [0]
        assert files[0] == data, (q, len(files[0]), len(data))


def test_batch_of_different_images_and_repeated_calls():
    """Six images per call, three calls on one stage: the word stream is left clean by every call."""
    w, h, q = 320, 200, 85
    coder = None
    for call in range(3):
        datas = [pillow_file(photo(w, h, 100 * call + k, noise=10 + 20 * k), q, "4:2:0") for k in range(6)]
        js = [O.jpeg_read_coefficients(d) for d in datas]
        coder = coder or coder_for(js[0], 6)
        files, status = coder.encode(planes_of(js), q)
        assert status == [0] * 6
        assert files == datas


def test_grayscale_and_flat_and_noise():
    data = pillow_file(photo(150, 97, 3)[:, :, 0], 80)
    j = O.jpeg_read_coefficients(data)
    assert coder_for(j, 1).encode(planes_of([j]), 80) == ([data], [0])
    data = pillow_file(np.full((128, 128, 3), 77, np.uint8), 90, "4:2:0")          # eight blocks per stream word
    j = O.jpeg_read_coefficients(data)
    assert coder_for(j, 1).encode(planes_of([j]), 90) == ([data], [0])
    rng = np.random.default_rng(11)
    data = pillow_file(rng.integers(0, 256, (256, 384, 3), dtype=np.uint8), 100, "4:2:0")   # stuffing in every chunk
    assert data.count(b"\xff\x00") > 100
    j = O.jpeg_read_coefficients(data)
    assert coder_for(j, 1).encode(planes_of([j]), 100) == ([data], [0])


def test_dropped_images_leave_the_others_alone():
    w, h, q = 96, 64, 90
    datas = [pillow_file(photo(w, h, k), q, "4:4:4") for k in range(4)]
    js = [O.jpeg_read_coefficients(d) for d in datas]
    js[2]["coef"][1] = js[2]["coef"][1].copy()
    js[2]["coef"][1].reshape(-1)[64 * 7 + 9] = -1500                             # 11 magnitude bits
    coder = coder_for(js[0], 4)
    files, status = coder.encode(planes_of(js), q)
    assert status == [0, 0, M.ENC_BAD_COEFFICIENT, 0]
    assert files == [datas[0], datas[1], None, datas[3]]
    # file_pitch one byte short for the longest file
    js[2] = O.jpeg_read_coefficients(datas[2])
    longest = max(len(d) for d in datas)
    files, status = coder.encode(planes_of(js), q, file_pitch=max(longest - 1, 1024))
    for d, f, s in zip(datas, files, status):
        assert (f, s) == ((None, M.ENC_FILE_OVERFLOW) if len(d) == longest else (d, 0))
    # and the stage is clean afterwards
    assert coder.encode(planes_of(js), q) == (datas, [0] * 4)
    # a scan longer than the stage's scan_capacity
    small = coder_for(js[0], 4, scan_capacity=4096)
    files, status = small.encode(planes_of(js), q)
    for d, f, s in zip(datas, files, status):
        assert s in (0, M.ENC_SCAN_OVERFLOW) and (f == d if s == 0 else f is None)
    assert M.ENC_SCAN_OVERFLOW in status or all(len(d) < 9000 for d in datas)


@pytest.mark.parametrize("pitch_of", [lambda longest: 1024, lambda longest: (longest // 2) & ~15, lambda longest: (longest - 16) & ~15,
                                      lambda longest: (longest + 15) & ~15])
def test_files_that_do_not_fit_never_write_behind_their_slot(pitch_of):
    """Guard region behind the batch's slots, slots too short for some or all files: a file that overflows is dropped
    (ENC_FILE_OVERFLOW), its neighbours are complete and nothing is stored behind the last slot."""
    rng = np.random.default_rng(5)
    w, h, q = 200, 152, 100
    datas = [pillow_file(rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if k % 2 else photo(w, h, k), q, "4:2:0") for k in range(6)]
    js = [O.jpeg_read_coefficients(d) for d in datas]
    longest = max(len(d) for d in datas)
    pitch = pitch_of(longest)
    coder = coder_for(js[0], 6)
    guard = 1 << 20
    buf = torch.full((6 * pitch + guard,), 0xA5, dtype=torch.uint8, device="cuda:0")
    files, lengths, status = coder.encode_device(planes_of(js), q, file_pitch=pitch, files=buf[:6 * pitch].view(6, pitch))
    torch.cuda.synchronize()
    assert bool((buf[6 * pitch:] == 0xA5).all()), "a byte landed behind the last slot"
    lengths, status, host = lengths.cpu().numpy(), status.cpu().numpy(), files.cpu().numpy()
    for i, d in enumerate(datas):
        if len(d) <= pitch:
            assert status[i] == 0 and host[i, :lengths[i]].tobytes() == d, i
        else:
            assert status[i] == M.ENC_FILE_OVERFLOW and lengths[i] == 0, i


@pytest.mark.parametrize("sub", ["420", "444"])
def test_forward_stage_plus_device_coder_equals_host_writer(sub):
    """BGRA frames in HBM -> files in HBM: the pixel stage's planes never leave the device; same bytes as the host writer."""
    hs, vs = {"420": ([2, 1, 1], [2, 1, 1]), "444": ([1, 1, 1], [1, 1, 1])}[sub]
    w, h, n, q = 801, 451, 5, 90
    stride = O.stride_for_width(w)
    frames = np.zeros((n, h, stride), np.uint8)
    for k in range(n):
        rgb = photo(w, h, 50 + k, noise=5 + 25 * k)
        px = frames[k, :, :4 * w].reshape(h, w, 4)
        px[..., 0], px[..., 1], px[..., 2], px[..., 3] = rgb[..., 2], rgb[..., 1], rgb[..., 0], 255
    stage = M.JpegForwardStage(w, h, hs, vs, n)
    qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(q)] * n).view(np.int16)).cuda()
    coef = stage.write_frames(Bitmap.from_numpy(frames, w, h, stride, "cuda:0"), qt)
    coder = M.JpegEntropyStage(w, h, hs, vs, stage.blocks_w, stage.blocks_h, n)
    files, status = coder.encode(coef, q)
    assert status == [0] * n
    host = M.write_jpeg_batch([c.cpu().numpy() for c in coef], w, h, hs, vs, q)
    assert files == host


def test_encoder_mirror_uses_the_device_coder():
    w, h, n = 200, 120, 3
    stride = O.stride_for_width(w)
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, (n, h, stride), dtype=np.uint8)
    enc = M.MozjpegEncoder.create_classic(quality=88)
    on_device = enc.write_frames(Bitmap.from_numpy(frames.copy(), w, h, stride, "cuda:0"))
    on_host = enc.write_frames(Bitmap.from_numpy(frames.copy(), w, h, stride, "cuda:0"), device_entropy=False)
    assert on_device == on_host and all(f[:2] == b"\xff\xd8" and f[-2:] == b"\xff\xd9" for f in on_device)


def test_full_size_frames_equal_the_host_writer_and_decode():
    """BASELINE's frame size: two 3840x2160 frames (194 400 blocks each: 760 workgroups per pass, hundreds of chunks) ->
    the device coder's files equal the host writer's and are files libjpeg reads back to the pixels within JPEG's error."""
    import io
    from PIL import Image
    w, h, n, q = 3840, 2160, 2, 90
    hs, vs = [2, 1, 1], [2, 1, 1]
    stride = O.stride_for_width(w)
    frames = np.zeros((n, h, stride), np.uint8)
    rgbs = [photo(w, h, 77 + k, noise=6 + 30 * k) for k in range(n)]
    for k, rgb in enumerate(rgbs):
        px = frames[k, :, :4 * w].reshape(h, w, 4)
        px[..., 0], px[..., 1], px[..., 2], px[..., 3] = rgb[..., 2], rgb[..., 1], rgb[..., 0], 255
    stage = M.JpegForwardStage(w, h, hs, vs, n)
    qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(q)] * n).view(np.int16)).cuda()
    coef = stage.write_frames(Bitmap.from_numpy(frames, w, h, stride, "cuda:0"), qt)
    coder = M.JpegEntropyStage(w, h, hs, vs, stage.blocks_w, stage.blocks_h, n)
    files, status = coder.encode(coef, q)
    assert status == [0, 0]
    assert files == M.write_jpeg_batch([c.cpu().numpy() for c in coef], w, h, hs, vs, q)
    for f, rgb in zip(files, rgbs):
        back = np.asarray(Image.open(io.BytesIO(f)).convert("RGB"), np.int32)
        assert back.shape == (h, w, 3)
        assert np.abs(back - rgb.astype(np.int32)).mean() < 20.0      # (4.3 / 14 through libjpeg-turbo's own encoder: 4:2:0 drops the chroma noise)
