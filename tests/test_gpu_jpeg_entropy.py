"""GPU parity of the entropy stage (csrc/jpeg_entropy.hip): the self-synchronising parallel Huffman decoder against the
oracle's serial decoder (pinned to libjpeg-turbo), coefficient for coefficient, on every committed JPEG -- plain,
restart intervals, optimised tables, grayscale -- decoded in batches of equal geometry; plus whole files -> BGRA."""
import io
import os
from collections import defaultdict

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from imageflow_amd.codecs import mozjpeg_decoder as D  # noqa: E402
from imageflow_amd.errors import FlowError  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda:0"


def groups(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    g = defaultdict(list)
    for i, n in enumerate(z["names"]):
        data = z[f"jpg_{i}"].tobytes()
        info = D.get_image_info(data)
        g[(info["width"], info["height"], info["ncomp"], tuple(info["hs"]), tuple(info["vs"]))].append((str(n), data))
    return g


@pytest.mark.parametrize("fixture", ["jpeg_cases.npz", "jpeg_encode_cases.npz", "jpeg_entropy_cases.npz"])
def test_batches_equal_the_serial_decoder(golden_dir, fixture):
    total = 0
    for key, items in groups(golden_dir, fixture).items():
        files = [d for _, d in items]
        ent = D.JpegEntropyBatch(files, DEV)
        coef = ent.read_coefficients()
        assert ent.rounds <= ent.n_subsequences + 2, (key, ent.rounds)      # worst case: one sub-sequence per round (q100 noise)
        for k, (name, data) in enumerate(items):
            j = O.jpeg_read_coefficients(data)
            assert np.array_equal(ent.qt[k][:j["ncomp"]], j["qt"][:j["ncomp"]]), name
            for c in range(j["ncomp"]):
                got = coef[c][k].cpu().numpy()
                assert got.shape == j["coef"][c].shape, name
                if not np.array_equal(got, j["coef"][c]):
                    bad = np.argwhere(got != j["coef"][c])
                    raise AssertionError(f"{name} comp {c}: {len(bad)} coefficients differ, first at {bad[0]}")
            total += 1
    assert total in (36, 108, 62)


def test_prepared_files_with_and_without_the_early_upload(golden_dir):
    """The per-file form of the batch (ifhip_jpeg_entropy_prepare -> [ifhip_jpeg_prepared_upload on the owner's stream] ->
    ifhip_jpeg_entropy_create_prepared), which the libimageflow ABI's decode coalescer uses: the same coefficients as the
    one-call form, whether the scan went up early (device to device into the batch, behind the owner's copy) or not."""
    side = torch.cuda.Stream(DEV)
    checked = 0
    for key, items in list(groups(golden_dir, "jpeg_entropy_cases.npz").items())[:8]:
        files = [d for _, d in items]
        want = D.JpegEntropyBatch(files, DEV).read_coefficients()
        for stream in (None, side, torch.cuda.current_stream(DEV)):
            ent = D.JpegEntropyBatch(files, DEV, prepared=True, upload_stream=stream)
            got = ent.read_coefficients()
            for c in range(ent.ncomp):
                assert torch.equal(got[c], want[c]), (key, c, stream)
            checked += 1
    # a handle whose upload is still queued when it is destroyed: the destroy waits, nothing is written into a recycled block
    big = [d for _, d in max(groups(golden_dir, "jpeg_entropy_cases.npz").values(), key=lambda it: len(it[0][1]))]
    L = D._bind_entropy()
    import ctypes as C
    for _ in range(20):
        h = C.c_void_p()
        keep = np.frombuffer(big[0], np.uint8)
        assert L.ifhip_jpeg_entropy_prepare(C.byref(h), keep.ctypes.data, len(keep)) == 0
        assert L.ifhip_jpeg_prepared_upload(h, C.c_void_p(side.cuda_stream)) == 0
        L.ifhip_jpeg_prepared_destroy(h)
    torch.cuda.synchronize()
    assert checked == 24


def test_files_to_bgra_on_device(golden_dir):
    for key, items in list(groups(golden_dir, "jpeg_entropy_cases.npz").items())[:6]:
        files = [d for _, d in items]
        frames = D.decode_frames(files, DEV).to_numpy()
        for k, (name, data) in enumerate(items):
            assert np.array_equal(frames[k], O.jpeg_idct_color(O.jpeg_read_coefficients(data))), name


def test_large_files_and_convergence():
    """4K-class inputs (BASELINE cfg4 shape): files of 3840x2160 4:2:0 q85, gradient + noise, no restart markers -- 7 000 and
    ~20 000 sub-sequences -- in the order noise, gradient, noise: the synchronisation launch dispatches the smallest scan's
    blocks first (wg_order), so the dispatch order differs from the batch's; the decoder must converge in a handful of
    rounds and every image must match the serial decode in ITS place."""
    PIL = pytest.importorskip("PIL.Image")
    w, h = 3840, 2160
    y, x = np.mgrid[0:h, 0:w]
    rng = np.random.default_rng(3)
    pics = [np.stack([x * 255 // (w - 1), y * 255 // (h - 1), (x + y) * 255 // (w + h - 2)], -1).astype(np.uint8),
            np.clip(np.stack([128 + 90 * np.sin(x / 9.0), 128 + 90 * np.cos(y / 7.0), 128 + 60 * np.sin((x + y) / 5.0)], -1)
                    + rng.integers(-30, 31, size=(h, w, 3)), 0, 255).astype(np.uint8)]
    files = []
    for p in (pics[1], pics[0], pics[1][::-1].copy()):
        buf = io.BytesIO()
        PIL.fromarray(p).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
        files.append(buf.getvalue())
    assert len(files[1]) < len(files[0]) // 2
    ent = D.JpegEntropyBatch(files, DEV)
    coef = ent.read_coefficients()
    assert ent.rounds <= 16, ent.rounds                 # typical content re-synchronises within a few symbols
    for k, data in enumerate(files):
        j = O.jpeg_read_coefficients(data)
        for c in range(3):
            assert np.array_equal(coef[c][k].cpu().numpy(), j["coef"][c]), (k, c)


def test_no_store_leaves_the_coefficient_planes_while_the_fixpoint_is_unsettled():
    """The write pass is enqueued behind the count pass before the host knows whether the synchronisation settled; on exit states
    that are not final a lane's first block and its block-in-MCU phase need not agree and its MCU row can pass the frame's last one.
    Found by tools/scan_entropy_shapes.py as a device fault behind the last image's plane (256 q100 files whose planes end on an
    allocation boundary); here: guard regions behind every plane must keep their pattern, and the coefficients match the serial decode."""
    PIL = pytest.importorskip("PIL.Image")
    w, h, n = 640, 480, 40                             # (at 40 and 64 files the build before the fix stored one block behind the luma plane)
    y, x = np.mgrid[0:h, 0:w]
    files = []
    for k in range(4):
        rng = np.random.default_rng(k)
        base = 128 + 90 * np.sin(x / 97.0 + k) * np.cos(y / 61.0) + 20 * np.sin(x / 7.0) * np.sin(y / 5.0)
        a = np.clip(base[..., None] + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        PIL.fromarray(a, "RGB").save(buf, "JPEG", quality=100, subsampling="4:2:0")
        files.append(buf.getvalue())
    batch = [files[i % 4] for i in range(n)]
    ent = D.JpegEntropyBatch(batch, DEV)
    guard = 1 << 20                                        # int16 elements behind each plane: room for thousands of stray blocks
    coef, bufs = [], []
    for c in range(3):
        shape = (n, ent.blocks_h[c], ent.blocks_w[c], 64)
        count = int(np.prod(shape))
        buf = torch.full((count + guard,), 0x5A5A, dtype=torch.int16, device=DEV)
        bufs.append((buf, count))
        coef.append(buf[:count].view(shape))
    seen_unsettled = False
    for _ in range(3):
        ent.read_coefficients(coef)
        seen_unsettled = seen_unsettled or ent.rounds > 1
        for buf, count in bufs:
            assert bool((buf[count:] == 0x5A5A).all()), "a store landed behind a coefficient plane"
    assert seen_unsettled, "these files no longer need a second round: the test lost its subject"
    for k in range(4):
        j = O.jpeg_read_coefficients(batch[k])
        for c in range(3):
            assert np.array_equal(coef[c][k].cpu().numpy(), j["coef"][c]) and np.array_equal(coef[c][k + 36].cpu().numpy(), j["coef"][c]), (k, c)


def test_rejections_and_corruption(golden_dir):
    z = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    with pytest.raises(FlowError):
        D.JpegEntropyBatch([z["progressive"].tobytes()], DEV)
    a, b = z["jpg_0"].tobytes(), z["jpg_20"].tobytes()           # different geometry in one batch
    with pytest.raises(FlowError):
        D.JpegEntropyBatch([a, b], DEV)
    cut = a[: len(a) * 2 // 3] + b"\xff\xd9"                     # scan ends early
    with pytest.raises(FlowError):
        D.JpegEntropyBatch([cut], DEV).read_coefficients()


def test_many_small_files_in_one_batch(golden_dir):
    """Hundreds of tiny files: a workgroup of 256 sub-sequences spans dozens of images (uniform standard tables -> the LDS
    copy serves all of them; an optimised-table file in the batch switches the foreign lanes to the global tables)."""
    z = np.load(os.path.join(golden_dir, "jpeg_cases.npz"))
    names = [str(n) for n in z["names"]]
    small = [z[f"jpg_{i}"].tobytes() for i, n in enumerate(names) if n.startswith("16x16_") and "4:2:0" in n]
    assert len(small) >= 2
    files = [small[k % len(small)] for k in range(300)]
    zz = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    opt = [zz[f"jpg_{i}"].tobytes() for i, n in enumerate(zz["names"]) if str(n).startswith("16x16_") and "4:2:0" in str(n) and "optimize=True" in str(n)]
    for batch in (files, files[:150] + opt[:1] + files[150:]):
        ent = D.JpegEntropyBatch(batch, DEV)
        coef = ent.read_coefficients()
        ref = {}
        for k in (0, 1, 149, 150, 151, len(batch) - 1):
            data = batch[k]
            j = ref.setdefault(data, O.jpeg_read_coefficients(data))
            for c in range(3):
                assert np.array_equal(coef[c][k].cpu().numpy(), j["coef"][c]), (k, c)


@pytest.mark.parametrize("hook", [{"ent_test_pool": "0"}, {"ent_test_pool": "40"}, {"ent_test_inner": "1"},
                                  {"ent_test_inner": "2", "ent_test_pool": "8"}])
def test_rarely_taken_paths(golden_dir, debug_switch, hook):
    """The serial code search (sub-tables that overflow the second-level pool: forced by a pool of 0 / 40 / 8 entries) and the
    rounds after an unsettled count pass (forced by one or two fixpoint iterations per launch) decode the same
    coefficients as the normal path."""
    for k, v in hook.items():
        debug_switch(k, v)
    checked, most_rounds = 0, 0
    for key, items in groups(golden_dir, "jpeg_entropy_cases.npz").items():
        files = [d for _, d in items]
        ent = D.JpegEntropyBatch(files, DEV)
        coef = ent.read_coefficients()
        most_rounds = max(most_rounds, ent.rounds)
        for i, (name, data) in enumerate(items):
            j = O.jpeg_read_coefficients(data)
            for c in range(j["ncomp"]):
                assert np.array_equal(coef[c][i].cpu().numpy(), j["coef"][c]), (name, c, hook)
            checked += 1
    assert checked == 62
    if "IFHIP_ENT_TEST_INNER" in hook:
        assert most_rounds > 4, most_rounds              # the host did iterate


def test_corrupted_scans_never_crash(golden_dir):
    """Untrusted bytes on the device: random bytes written into the entropy-coded data (and Huffman tables swapped for
    another file's) decode to SOMETHING or are refused -- no fault, no hang -- and the next clean decode is exact."""
    z = np.load(os.path.join(golden_dir, "jpeg_entropy_cases.npz"))
    names = [str(n) for n in z["names"]]
    rng = np.random.default_rng(9)
    picks = [i for i, n in enumerate(names) if "64x" in n or "640x" in n or "203x" in n][:6] or list(range(6))
    outcomes = {"ok": 0, "refused": 0}
    for i in picks:
        data = z[f"jpg_{i}"].tobytes()
        sos = data.find(b"\xff\xda")
        for _ in range(12):
            m = bytearray(data)
            for _ in range(int(rng.integers(1, 12))):
                m[int(rng.integers(sos + 14, len(m) - 2))] = int(rng.integers(0, 256))
            try:
                D.JpegEntropyBatch([bytes(m)], DEV).read_coefficients()
                outcomes["ok"] += 1
            except FlowError:
                outcomes["refused"] += 1
        ent = D.JpegEntropyBatch([data], DEV)
        coef = ent.read_coefficients()
        j = O.jpeg_read_coefficients(data)
        for c in range(j["ncomp"]):
            assert np.array_equal(coef[c][0].cpu().numpy(), j["coef"][c]), names[i]
    assert outcomes["ok"] + outcomes["refused"] == 12 * len(picks) and outcomes["refused"] > 0, outcomes
