#!/usr/bin/env python3
"""bench.py -- megapixels/s of the resample hot path (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--workload ...]

Default workload (config.workload): BASELINE config 2 -- a batch of 256 synthetic 3840x2160 BGRA8 frames per GPU, device
resident, each resized to 200x200 with Robidoux in linear light (ReplaceSelf canvas, alpha not meaningful, as after a
JPEG decode).  One "step" = one pass of the hot path over the rank's whole batch.

Multi-GPU: images are independent, so frame i of a job belongs to rank floor(i * N / total)
(imageflow_amd/sharding.py) and nothing is exchanged on the data path; the only collective is the RCCL gather of a batch's
finished outputs to rank 0 -- ONE PER BATCH (= per step), asynchronous and double buffered so that the gather of batch k runs
beside the kernel of batch k + 1 (--gather every, the N > 1 default; RCCL's channels are capped and the resample grid is sized
for the CUs the gather leaves, or not: --reserve-cus auto measures both in the warm-up).  value = frames /
time of the steps WITH their gathers; the line also carries `value_without_gather` (a second timed loop without any),
`gather_ms` (what a gather adds per batch after overlap) and, for the default workload, `strong_1024`: the north_star job (a
1 024-image batch cut into N blocks) measured the same way in the same run, so one series of runs at N = 1, 2, 4, 8 gives both
scaling curves.  The warm-up steps gather as the timed ones do (RCCL's lazy peer-to-peer set-up is never timed).
  --scaling weak   (default) every rank owns --frames frames (256): per-GPU work fixed.
  --scaling strong the job is --total-frames frames (1024, the north_star batch) cut into N contiguous blocks.
`--gpus N` with no torch.distributed environment re-executes this file under `python -m torch.distributed.run` with N
ranks; under a launcher, WORLD_SIZE must equal --gpus (anything else is an error, never a silent 1-GPU run).

The default run (N = 1, cfg2) also measures BASELINE configs 5, 3 and 4 -- child runs of this file with --workload, reduced step
counts -- and reports them as `other_configs`: time per step, the kernel-timed roofline fraction, the parity stamp, the workload.

--workload cfg3 is BASELINE config 3 as a job: the export_4_sizes pyramid 3840x2160 -> 1600x900 -> {1200x675 -> 400x225,
800x450} (imageflow_tool/src/self_test.rs:185-198), four chained launches per batch, 58 737 600 algorithmic bytes per
image, 128 frames per GPU by default (1024 images over 8 GPUs).  --outputs files (default): every output goes through the
job's encoder as in the reference (`libjpeg_turbo` quality 90: forward pixel stage + entropy coder on the device) and
the batch's gather ships the FILES -- ~0.26 bytes per pixel instead of 4 (1.38 GB of BGRA per rank became ~90 MB);
--outputs bgra keeps the round-3 form (raw levels gathered).  `roofline` stays the four resample launches; the line
also carries what a step costs without the encoders.

--workload cfg4 is BASELINE config 4: 3840x2160 4:2:0 q85 baseline JPEG files (SURVEY.md 8d's recipe: the gradient of
bench_codecs.rs:24-41 and a noise variant, written once on the host with Pillow) -> GPU entropy decode -> 4/8 IDCT with the
spatial sRGB luma scaler + YCbCr (what the reference's decoder is asked for when the job wants 800 px:
codecs/mozjpeg_decoder.rs:295-420,588-618) -> 800x450 Robidoux (flow/nodes/scale_render.rs:304-313), 128 files per GPU
(1 024 over 8 GPUs) in batches of 64, --batches-in-flight of them on the device at a time (default 2: one host thread and HIP
stream each, a thread's batches one after the other).  The compressed scans are resident in HBM
(un-stuffed, as the entropy stage reads them) when the timed region starts.  value = source megapixels through the WHOLE
chain per second; `roofline` is the pixel stage + resize call alone on SURVEY 8d's 26 323 584 bytes per image (the entropy
walk is bit-serial work: its yardstick is instruction issue, `roofline_entropy`); cpu_baseline = libjpeg-turbo (Pillow)
DCT-domain 1/2 decode + resize on all host cores.  Images are sharded in contiguous blocks like every other workload and the
finished 800x450 outputs are gathered behind every step.

Prints ONE JSON line (rank 0).  value = source megapixels resized per second over all ranks, inputs already in HBM.
After the timed region (never inside it) rank 0 renders frames of the last step's output again on the CPU oracle and
compares: `parity_checked` (BASELINE config 2: "... bit-exact check vs CPU").  N > 1 (on unless --no-selfcheck): every rank's
first gathered frame is compared with one rendered locally on rank 0 -- catches a gather that lands at the wrong offset.
"""
import argparse
import json
import os
import shutil
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_PROC_BIND", "spread")                    # cpu_baseline: one frame per core, threads pinned apart
os.environ.setdefault("OMP_PLACES", "cores")

FRAMES_PER_GPU = 256
HBM_PEAK = 8.0e12                                                   # MI355X_MICROARCH.md: 8 TB/s spec

# name: (in_w, in_h, out_w, out_h, filter, sharpen, alpha_meaningful, compositing, matte, default frames per GPU)
WORKLOADS = {
    "cfg2": (3840, 2160, 200, 200, "Robidoux", 0.0, False, "ReplaceSelf", 0, 256),
    "cfg2-alpha": (3840, 2160, 200, 200, "Robidoux", 0.0, True, "ReplaceSelf", 0, 256),
    "cfg5": (7680, 4320, 400, 225, "Lanczos", 15.0, True, "BlendWithMatte", 0xFFFFFFFF, 64),
    "cfg3-l0": (3840, 2160, 1600, 900, "Robidoux", 0.0, False, "ReplaceSelf", 0, 256),
    "cfg3-l1": (1600, 900, 1200, 675, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg3-l2": (1600, 900, 800, 450, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg3-l3": (1200, 675, 400, 225, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg4-resize": (1920, 1080, 800, 450, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg1-resize": (480, 270, 200, 113, "Robidoux", 0.0, False, "ReplaceSelf", 0, 4096),
    # one strip of cfg5 as a frame of its own (same ring, same taps, rows contiguous in memory): tells strip access from arithmetic
    "cfg5-quarter": (1920, 4320, 100, 225, "Lanczos", 15.0, True, "BlendWithMatte", 0xFFFFFFFF, 256),
    # up-scaling (no BASELINE config; the shapes of the reference's offline-replayable tests, visuals/canvas.rs:8-33 and
    # trim.rs:131-158): 2x stays on the fused kernel, 3x has more than 8 live rows and takes the generic two-pass kernels
    "up2-hermite": (200, 200, 400, 400, "Hermite", 0.0, True, "ReplaceSelf", 0, 4096),
    "up3-robidoux": (100, 100, 300, 300, "Robidoux", 0.0, True, "ReplaceSelf", 0, 8192),
    # the up-scales a decoded JPEG meets (no alpha, the node's default up filter, scale_render.rs:255-259): HD frames 2x and 1.5x
    "up2-ginseng-hd": (960, 540, 1920, 1080, "Ginseng", 0.0, False, "ReplaceSelf", 0, 512),
    "up1.5-ginseng-hd": (1280, 720, 1920, 1080, "Ginseng", 0.0, False, "ReplaceSelf", 0, 512),
    # the whole export_4_sizes job: the tuple describes level 0, PYRAMID the chain
    "cfg3": (3840, 2160, 1600, 900, "Robidoux", 0.0, False, "ReplaceSelf", 0, 128),
}
CFG4 = {"in_w": 3840, "in_h": 2160, "dec_w": 1920, "dec_h": 1080, "out_w": 800, "out_h": 450, "files_per_gpu": 128, "batch": 64,
        "bytes_per_image": 26_323_584}                             # coefficients + quant tables in, 800x450 BGRA out
PYRAMID = [("src", "1600", 1600, 900), ("1600", "1200", 1200, 675), ("1600", "800", 800, 450), ("1200", "400", 400, 225)]
PYRAMID_BYTES_PER_IMAGE = 58_737_600
PYRAMID_WRITE_BYTES_PER_IMAGE = 10_800_000                         # of which written: the four levels (1600x900 + 1200x675 + 800x450 + 400x225) x 4 B


def make_frames(torch, n, first_index, seed, device, pattern, in_w, in_h):
    """frame k pixel (x,y): B=(x+k)&255, G=(y+k)&255, R=(x+y+k)&255, A=255 (bench_graphics.rs:403-414 + offset);
    the random half is uniform bytes (worst case for LUT bank conflicts and rounding)."""
    from imageflow_amd.graphics.bitmaps import Bitmap, get_stride
    stride = get_stride(in_w)
    data = torch.empty((n, in_h, stride), dtype=torch.uint8, device=device)
    x = torch.arange(in_w, device=device, dtype=torch.int32)[None, :]
    y = torch.arange(in_h, device=device, dtype=torch.int32)[:, None]
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + seed)
    for i in range(n):
        k = first_index + i
        use_random = pattern == "random" or (pattern == "mixed" and i >= n // 2)
        if use_random:
            data[i] = torch.randint(0, 256, (in_h, stride), dtype=torch.uint8, device=device, generator=gen)
            data[i, :, 3::4] = 255
        else:
            px = data[i, :, : in_w * 4].view(in_h, in_w, 4)
            px[..., 0] = ((x + k) & 255).to(torch.uint8)
            px[..., 1] = ((y + k) & 255).to(torch.uint8).expand(in_h, in_w)
            px[..., 2] = ((x + y + k) & 255).to(torch.uint8)
            px[..., 3] = 255
    return Bitmap(data.view(n, in_h * stride), in_w, in_h, stride, alpha_meaningful=False)


def _usable_cpus_now():
    """CPUs this process may really use: the affinity mask and the cgroup's CPU quota, not the host's logical CPU count (a GPU
    box of this pool shows 256 logical CPUs to a container that is allowed 16: 256 threads on it measure the scheduler)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                    # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                  # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


_USABLE_CPUS = _usable_cpus_now()        # read at import: once an OpenMP runtime has bound the main thread to its place (OMP_PROC_BIND,
                                         # set above for the baseline's threads) the affinity mask of this thread is one core


def usable_cpus():
    return _USABLE_CPUS


def cpu_baseline(sample_seconds=8.0):
    """The oracle (our C port of the reference's CPU path) timed on the host cores on a bounded sample of cfg2.  The
    Rust reference itself (cargo bench -p imageflow_core --bench bench_graphics -- full_scale_pipeline,
    benches/bench_graphics.rs:382-456) needs a Rust toolchain AND the reference tree; both are probed and reported."""
    import numpy as np
    from oracle import oracle as O
    from tests import util as U
    in_w, in_h, out_w, out_h = 3840, 2160, 200, 200
    cores = usable_cpus()
    fr = U.gradient_frames(1, in_w, in_h)
    cst = U.stride_for(out_w)
    can = np.zeros((1, out_h, cst), np.uint8)
    t0 = time.perf_counter()
    O.scale_and_render_batch(fr.reshape(1, -1), can.reshape(1, -1), in_w, in_h, fr.shape[2], out_w, out_h, cst,
                             0, 0, out_w, out_h, n_threads=1)
    t1 = time.perf_counter() - t0
    n = max(cores, 8)                                    # every thread has a frame in every pass (33 MB each)
    frames = np.concatenate([U.gradient_frames(n // 2, in_w, in_h), U.random_frames(n - n // 2, in_w, in_h, alpha=False)])
    cans = np.zeros((n, out_h * cst), np.uint8)
    flat = frames.reshape(n, -1)

    def one_pass():
        t0 = time.perf_counter()
        rc = O.scale_and_render_batch(flat, cans, in_w, in_h, frames.shape[2], out_w, out_h, cst, 0, 0, out_w, out_h,
                                      n_threads=cores)
        assert rc == 0
        return time.perf_counter() - t0
    first = one_pass()                                   # also warms the page cache / thread pool
    reps = int(max(1, min(200, sample_seconds / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    total = time.perf_counter() - t0
    mp = reps * n * in_w * in_h / 1e6
    cargo = shutil.which("cargo")
    ref_tree = os.path.isdir("/root/reference/imageflow_core")
    return {"value": round(mp / total, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "host_logical_cpus": os.cpu_count(), "single_thread_MPps": round(in_w * in_h / 1e6 / t1, 2),
            "sample": f"{reps} passes over {n} frames 3840x2160->200x200 Robidoux linear (half gradient, half random), "
                      f"{total:.1f} s wall, oracle/if_oracle.c -O3 x86-64-v3, {cores} OpenMP threads = the CPUs this process may use "
                      f"(affinity and cgroup quota; the host shows {os.cpu_count()}), one frame per thread, "
                      f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}",
            "note": "a NAIVE scalar port of the reference's arithmetic (test infrastructure), not a tuned resizer: expect it "
                    ">= 10x below the reference's own SIMD (zenresize) path on the same cores; never a target, never credit",
            "reference_leg": ("cargo present but the reference tree is not on this box" if cargo and not ref_tree else
                              "not run: no Rust toolchain on this box (cargo not found)" if not cargo else
                              "cargo and tree present: run `cargo bench -p imageflow_core --bench bench_graphics -- full_scale_pipeline` by hand")}


def host_dropin_rate(torch, seconds=3.0):
    """ifhip_scale_and_render -- the symbol graphics/scaling.rs would bind: host buffers in,
    host buffers out, PCIe both ways.  One 4K frame -> 200x200 per call, calls back to back from one thread."""
    import numpy as np
    from imageflow_amd.graphics.bitmaps import get_stride
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render_host
    in_w, in_h, out_w, out_h = 3840, 2160, 200, 200
    st, cst = get_stride(in_w), get_stride(out_w)
    frame = np.random.default_rng(5).integers(0, 256, (in_h, st), dtype=np.uint8)
    can = np.zeros((out_h, cst), np.uint8)
    info = ScaleAndRenderParams(0, 0, out_w, out_h)
    for _ in range(3):
        scale_and_render_host(frame, in_w, in_h, st, False, can, out_w, out_h, cst, info)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        scale_and_render_host(frame, in_w, in_h, st, False, can, out_w, out_h, cst, info)
        n += 1
    dt = time.perf_counter() - t0
    return {"images_per_s": round(n / dt, 1), "MPps": round(n * in_w * in_h / 1e6 / dt, 1), "ms_per_call": round(dt / n * 1e3, 3),
            "what": "ifhip_scale_and_render, one 3840x2160 frame -> 200x200 per call, pageable host buffers, one thread, "
                    "PCIe inclusive (persistent pinned staging + per-thread stream); never `value`"}


def resample_parity(torch, wl, inp, canvas, frames):
    """After the timed region: frames of the LAST step's canvas rendered again by the CPU oracle (oracle/if_oracle.c) from
    the same device-resident inputs; BGRA8 must be equal byte for byte (BASELINE cfg2: "bit-exact check vs CPU")."""
    import numpy as np
    from imageflow_amd.graphics.bitmaps import BitmapCompositing
    from imageflow_amd.graphics.weights import Filter
    from tests import util as U
    in_w, in_h, out_w, out_h = wl[:4]
    got = canvas.to_numpy()
    equal, checked = True, []
    for i in sorted(set(frames)):
        src = inp.data[i].cpu().numpy().reshape(1, in_h, inp.stride)
        exp = np.zeros((1, out_h, got.shape[2]), np.uint8)
        U.oracle_render(src, in_w, in_h, exp, out_w, out_h, 0, 0, out_w, out_h, filter_id=int(Filter[wl[4]]), sharpen=wl[5],
                        compositing=int(BitmapCompositing[wl[7]]), matte_bgra=wl[8], alpha_meaningful=wl[6])
        equal = equal and bool(np.array_equal(got[i][:, :4 * out_w], exp[0][:, :4 * out_w]))
        checked.append(int(i))
    return {"frames": len(checked), "which": checked, "equal": equal, "against": "oracle.scale_and_render (oracle/if_oracle.c), BGRA8 byte for byte"}


def pyramid_parity(torch, inp, levels):
    """The same for the export_4_sizes job: frame 0's four levels against the oracle's chain from the same source."""
    import numpy as np
    from tests import util as U
    cur = {"src": (inp.data[0].cpu().numpy().reshape(1, inp.h, inp.stride), inp.w, inp.h)}
    equal = True
    for a, b, w, h in PYRAMID:
        src, sw, sh = cur[a]
        exp = np.zeros((1, h, U.stride_for(w)), np.uint8)
        U.oracle_render(src, sw, sh, exp, w, h, 0, 0, w, h)
        cur[b] = (exp, w, h)
        equal = equal and bool(np.array_equal(levels[b].to_numpy()[0][:, :4 * w], exp[0][:, :4 * w]))
    return {"frames": 1, "which": [0], "levels": len(PYRAMID), "equal": equal,
            "against": "oracle.scale_and_render chained as the job chains its levels, BGRA8 byte for byte"}


def _cfg4_one_file(k):
    import io
    import numpy as np
    from PIL import Image
    w, h = CFG4["in_w"], CFG4["in_h"]
    y, x = np.mgrid[0:h, 0:w].astype(np.int32)
    rgb = np.stack([((x + 7 * k) % w) * 255 // w, ((y + 5 * k) % h) * 255 // h, ((x + y + k) * 3) & 255], -1).astype(np.int16)
    if k & 1:
        rgb = rgb + np.random.default_rng(4000 + k).integers(-12, 13, size=rgb.shape, dtype=np.int16)
    buf = io.BytesIO()
    Image.fromarray(np.clip(rgb, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=85, subsampling="4:2:0", optimize=False)
    return buf.getvalue()


def cfg4_files(first_index, n):
    """SURVEY.md 8d's cfg4 inputs: 3840x2160 baseline 4:2:0 q85 files of the gradient of bench_codecs.rs:24-41
    (R = 255 x / w, G = 255 y / h, B = 3 (x + y) & 255, shifted per file) -- every second file with uniform noise of +-12
    added -- written on the host with Pillow (libjpeg-turbo, optimize=False).  File k's content depends on k only.
    (About half a second of one core per file: written by a pool of forked workers -- numpy and Pillow only, they never
    touch HIP -- so that 128 files take seconds, not a minute; main() makes the files before it initialises the device.)"""
    ks = list(range(first_index, first_index + n))
    procs = min(len(ks), usable_cpus(), 32)
    if procs >= 2 and len(ks) >= 4:
        import multiprocessing as mp
        try:
            with mp.get_context("fork").Pool(procs) as pool:
                return pool.map_async(_cfg4_one_file, ks, chunksize=1).get(timeout=300)
        except Exception as e:  # noqa: BLE001 -- a pool that cannot start or stalls: the files are made here
            print(f"bench.py: file pool failed ({type(e).__name__}: {e}); writing the files serially", file=sys.stderr, flush=True)
    return [_cfg4_one_file(k) for k in ks]


def _cfg4_cpu_one(data):
    import io
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.draft("RGB", (CFG4["dec_w"], CFG4["dec_h"]))                # libjpeg-turbo's DCT-domain 1/2 decode, as the reference's decoder hint
    im = im.convert("RGB").resize((CFG4["out_w"], CFG4["out_h"]), Image.BICUBIC)
    return im.size[0]


def cfg4_cpu_baseline(files, sample_seconds=12.0):
    """libjpeg-turbo (Pillow) decode at 1/2 scale + bicubic resize to 800x450 on every host core, one file per process at a
    time, on a bounded sample of the run's own files.  Not the reference's Rust / mozjpeg path (no cargo here) and not a
    bit-exact stand-in for it: a yardstick for "what the host's cores do with these files", next to the 50.1 MP/s per core
    the reference publishes for its own full decode (benchmarks/c-vs-zen-codecs-2026-04-15-singlethread.txt:17)."""
    import multiprocessing as mp
    cores = usable_cpus()
    t0 = time.perf_counter()
    _cfg4_cpu_one(files[0])
    t1 = time.perf_counter() - t0
    procs = min(cores, 64)
    per_pass = max(procs, len(files))
    work = [files[i % len(files)] for i in range(per_pass)]
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_cfg4_cpu_one, work[:procs])                      # workers up, libraries loaded
        t0 = time.perf_counter()
        passes = 0
        while time.perf_counter() - t0 < sample_seconds and passes < 50:
            pool.map(_cfg4_cpu_one, work, chunksize=1)
            passes += 1
        dt = time.perf_counter() - t0
    mpix = passes * per_pass * CFG4["in_w"] * CFG4["in_h"] / 1e6
    return {"value": round(mpix / dt, 1), "unit": "MP/s", "cores": procs, "kind": "port",
            "single_thread_MPps": round(CFG4["in_w"] * CFG4["in_h"] / 1e6 / t1, 1),
            "sample": f"{passes} passes over {per_pass} of the run's files (3840x2160 4:2:0 q85): Pillow/libjpeg-turbo draft decode to "
                      f"1920x1080 + bicubic resize to 800x450, {procs} processes on {cores} cores, {dt:.1f} s wall",
            "note": "libjpeg-turbo through Pillow, not imageflow_core's mozjpeg decoder + zenresize (no Rust toolchain on this box): a host "
                    "yardstick, never a target; the reference's own published full decode is 50.1 MP/s on one core",
            "published_reference_decode_MPps_one_core": 50.1}


def run_cfg4(args, torch, dist, world, rank, local_rank, dev, dryrun, inputs, rccl_env, grouped):
    """BASELINE config 4 (see the module docstring): files -> entropy decode -> 4/8 pixel stage -> 800x450, sharded."""
    import threading

    import numpy as np

    from imageflow_amd.codecs import mozjpeg_decoder as D
    from imageflow_amd.graphics.bitmaps import Bitmap
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams
    from imageflow_amd.sharding import gather_to_root, max_over_ranks, shard_range
    distributed = grouped
    w, h, ow, oh = CFG4["in_w"], CFG4["in_h"], CFG4["out_w"], CFG4["out_h"]
    if args.scaling == "strong":
        total = args.total_frames
        lo, hi = shard_range(total, rank, world)
    else:
        per = args.frames or CFG4["files_per_gpu"]
        total = per * world
        lo, hi = rank * per, (rank + 1) * per
    n = hi - lo
    if n < 1:
        raise SystemExit(f"rank {rank} owns no files ({total} files over {world} ranks)")
    n_max = -(-total // world)
    T = max(1, min(args.batches_in_flight, n))
    per_batch = max(1, args.files_per_batch or CFG4["batch"])
    ranks_info = [{"rank": rank, "device": torch.cuda.get_device_name(local_rank), "local_rank": local_rank}]
    if distributed:
        objs = [None] * world
        dist.all_gather_object(objs, ranks_info[0])
        ranks_info = objs
    assert inputs[0] == lo and len(inputs[1]) == n, "main() wrote this rank's files before the device was initialised"
    files = inputs[1]
    torch.zeros(1, device=dev).item()
    info = ScaleAndRenderParams(0, 0, ow, oh)
    out_all = Bitmap.create_u8(n_max, ow, oh, dev)                 # this rank's outputs, gather-slot sized
    # T host threads, each with its contiguous share of the rank's files cut into batches of CFG4["batch"] files, its own HIP
    # stream and buffers: a thread decodes its batches one after the other, T batches are in flight on the device -- while one
    # batch's synchronisation tail leaves CUs idle, another one's dense passes fill them.  (Measured, 128 files per step,
    # profiles/r5_bench_cfg4_batch_sweep.txt: 2 x 64 files in flight 21 400 files/s, 2 x 32 20 200, 4 x 32 18 900, 4 x 16 19 200,
    # 4 x 8 15 000; the pixel stage + resize call runs at 0.30 of 8 TB/s on 64 frames, 0.27 on 16.)
    ctx = []
    for t in range(T):
        a, b = shard_range(n, t, T)
        st = torch.cuda.Stream(device=dev)
        batches = []
        with torch.cuda.stream(st):
            for a0 in range(a, b, per_batch):
                b0 = min(b, a0 + per_batch)
                ent = D.JpegEntropyBatch(files[a0:b0], device=str(dev))
                coef = ent.read_coefficients()
                stage = D.JpegPixelStage(w, h, 3, ent.h_samp, ent.v_samp, b0 - a0, device=str(dev), scale_num=4, luma_spatial=True, luma_srgb=True)
                qt = torch.from_numpy(ent.qt.view(np.int16)).to(dev)
                small = Bitmap(out_all.data[a0:b0], ow, oh, out_all.stride, False)
                fused = stage.read_frames_into(coef, qt, small, info)
                batches.append({"ent": ent, "coef": coef, "stage": stage, "qt": qt, "small": small, "fused": fused, "n": b0 - a0})
        ctx.append({"stream": st, "batches": batches})
    torch.cuda.synchronize()
    compressed = sum(len(f) for f in files)
    all_batches = [bt for c in ctx for bt in c["batches"]]

    def chain(c, steps):
        with torch.cuda.stream(c["stream"]):
            for _ in range(steps):
                for bt in c["batches"]:
                    bt["ent"].read_coefficients(bt["coef"])
                    bt["stage"].read_frames_into(bt["coef"], bt["qt"], bt["small"], info)
            c["stream"].synchronize()

    def run_steps(steps):
        th = [threading.Thread(target=chain, args=(c, steps)) for c in ctx[1:]]
        for t in th:
            t.start()
        chain(ctx[0], steps)
        for t in th:
            t.join()

    mode = "none" if (not distributed or args.no_gather) else args.gather
    gathered = None
    if mode != "none" and rank == 0:
        gathered = torch.empty((world, n_max, out_all.image_bytes), dtype=torch.uint8, device="cpu" if dryrun else dev)
    gather_note = {"none": "none",
                   "every": "rccl gather of the batch's 800x450 outputs to rank 0 EVERY step (one per batch), in line: step, gather, next step",
                   "final": "rccl gather of the 800x450 outputs to rank 0 once, at the end of the timed region (the same gather ran once during "
                            "warm-up); amortised over the steps: not the per-batch figure"}[mode]
    gathers = {"warmup": 0, "timed": 0}
    gather_ok = True

    def gather_outputs(phase):
        nonlocal gather_note, gather_ok
        if not gather_ok:
            return
        try:
            gather_to_root(out_all.data.cpu() if dryrun else out_all.data, 0, out=gathered if rank == 0 else None)
            gathers[phase] += 1
        except Exception as e:  # noqa: BLE001
            gather_ok = False
            gather_note = f"rccl gather failed: {type(e).__name__}: {e}"

    def timed(steps, phase):
        """-> seconds for `steps` batches; with phase and --gather every each batch is followed by its gather (the decode
        threads of a step have joined: the outputs are complete), with --gather final the steps by one gather."""
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if phase is not None and mode == "every":
            for _ in range(steps):
                run_steps(1)
                gather_outputs(phase)
        else:
            run_steps(steps)
        torch.cuda.synchronize()
        t_steps = time.perf_counter() - t0
        if phase is not None and mode == "final":
            gather_outputs(phase)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_steps

    timed(args.warmup, "warmup")
    if mode == "final" and gathers["warmup"] == 0:
        gather_outputs("warmup")
    el, t_steps = timed(args.steps, "timed")
    elapsed = max_over_ranks(el, dev)
    if mode == "every":
        compute_s = max_over_ranks(timed(args.steps, None)[0], dev)
    else:
        compute_s = max_over_ranks(t_steps if mode == "final" else el, dev)

    # the pixel stage + resize call alone (SURVEY 8d's unit: coefficient planes in, 800x450 out), hipEvents on its stream,
    # one batch at a time so that nothing else shares the device
    launches = max(10, min(args.steps, 20))
    if elapsed < 0.6 and args.steps > 0:               # a reduced run: bring the clocks up before the probe (see the resample workloads' probe)
        run_steps(min(400, int((0.6 - elapsed) / max(elapsed / args.steps, 1e-4)) + 1))
        torch.cuda.synchronize()
    px_ms, ent_ms = 0.0, 0.0
    for c in ctx:
        with torch.cuda.stream(c["stream"]):
            for bt in c["batches"]:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record(c["stream"])
                for _ in range(launches):
                    bt["stage"].read_frames_into(bt["coef"], bt["qt"], bt["small"], info)
                e1.record(c["stream"])
                for _ in range(launches):
                    bt["ent"].read_coefficients(bt["coef"])
                e2.record(c["stream"])
                c["stream"].synchronize()
                px_ms += e0.elapsed_time(e1) / launches
                ent_ms += e1.elapsed_time(e2) / launches
    torch.cuda.synchronize()

    parity = None
    selfcheck = None
    if rank == 0:
        parity = cfg4_parity(torch, files, out_all, [0, n - 1])
        if args.selfcheck and gathered is not None and gather_ok:
            selfcheck = cfg4_selfcheck(torch, gathered, total, world, dev, out_all)
        mp_per_step = total * w * h / 1e6
        algo = n * CFG4["bytes_per_image"]
        achieved = algo / (px_ms * 1e-3)
        out = {
            "metric": "megapixels/sec JPEG decode + resize (4K 4:2:0 q85 -> 800px)", "value": round(mp_per_step * args.steps / elapsed, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "i32/f32", "data": "synthetic",
            "gather_ms": round((elapsed - compute_s) / (args.steps if mode == "every" else 1) * 1e3, 4),
            "value_without_gather": round(mp_per_step * args.steps / compute_s, 1),
            "files_per_s": round(total * args.steps / elapsed, 1),
            "config": {"workload": f"BASELINE cfg4: {total} files ({n} on rank 0) 3840x2160 4:2:0 q85 baseline JPEG -> GPU entropy decode -> 4/8 IDCT "
                                   f"(spatial sRGB luma scaler) + YCbCr -> 800x450 Robidoux, linear light; compressed scans device resident",
                       "files_per_gpu": n, "total_files": total, "batches_in_flight": T, "files_per_batch": [bt["n"] for bt in all_batches],
                       "compressed_MB_per_gpu": round(compressed / 1e6, 2),
                       "one_call_chain": bool(all(bt["fused"] for bt in all_batches)),
                       "kernel": "jpeg entropy passes + luma / chroma IDCT kernels + the resampler reading the component planes",
                       "gather": gather_note, "gathers": gathers, "gather_mode": mode, "rccl_channels": rccl_env,
                       "rccl_ranks": dist.get_world_size() if distributed else 1,
                       "backend": (dist.get_backend() if distributed else "none"), "ranks": ranks_info,
                       "entropy_decode_ms_per_step": round(ent_ms, 4), "pixel_stage_and_resize_ms_per_step": round(px_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4),
                         "frac_timed": round(algo / (compute_s / args.steps) / HBM_PEAK, 4),
                         "traffic": None, "traffic_source": "profiles/ (rocprofv3 --pmc passes of tools/bench_jpeg.py --chain), not measured in this run",
                         "kernel_ms": round(px_ms, 4), "algorithmic_bytes_per_launch": algo,
                         "what": "the pixel stage + resize calls of one step (coefficient planes + quant tables in, 800x450 BGRA out: 26 323 584 B per image, "
                                 "SURVEY 8d), batches one after the other, hipEvents on their streams; frac_timed puts the same bytes over the WHOLE "
                                 "chain's step time, entropy decode included"},
            "parity_checked": parity,
        }
        try:
            out["roofline_entropy"] = cfg4_entropy_roofline(files, n, ent_ms)
        except Exception as e:  # noqa: BLE001
            out["roofline_entropy"] = {"bound": "issue", "frac": None, "error": f"{type(e).__name__}: {e}"}
        if selfcheck is not None:
            out["selfcheck"] = selfcheck
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cfg4_cpu_baseline(files)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "MP/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)


def cfg4_entropy_roofline(files, n_files, ent_ms):
    """The yardstick of the entropy stage (92 % of a cfg4 step): instruction issue, not HBM.  Symbols and table reads of a sample
    of the run's files (host walk, ifhip_jpeg_debug_scan_report) x the wave-instructions each pass issues per 64 lane-steps
    (static: profiles/entropy_issue_model.json, SQ counters of this build's kernels) against what the chip's SIMDs can issue
    in the measured time of the stage."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from imageflow_amd import _native

    class Report(C.Structure):
        _fields_ = ([(k, C.c_uint32) for k in ("segments", "sub_sequences", "scan_complete", "pool_entries", "pool_entries_used", "prefixes_left_to_search",
                                               "pair_entries", "segments_with_wrong_block_count", "segments_with_invalid_codes", "pair_walk_mismatches",
                                               "count_walk_mismatches")]
                    + [("symbols", C.c_uint64), ("table_reads_with_pairs", C.c_uint64), ("blocks", C.c_uint64), ("dc_sum", C.c_int32 * 3), ("dc_last_segment", C.c_int32 * 3)])
    L = _native.lib()
    L.ifhip_jpeg_debug_scan_report.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Report)]
    sample = files[:min(len(files), 16)]                 # gradient and noise files alternate: an even sample has the run's mix

    def one(f):
        r = Report()
        _native.check(L.ifhip_jpeg_debug_scan_report(f, len(f), C.byref(r)))
        return int(r.symbols), int(r.table_reads_with_pairs), int(r.sub_sequences)
    with ThreadPoolExecutor(8) as ex:
        rows = list(ex.map(one, sample))
    scale = n_files / len(sample)
    symbols, reads, subs = (sum(r[k] for r in rows) * scale for k in range(3))
    model = json.load(open(os.path.join(ROOT, "profiles", "entropy_issue_model.json")))
    per = model["per_64_lane_steps"]
    steps = {"round": 2.0 * reads, "count": reads, "write": symbols}
    simds, clock = 256 * 4, 2.4e9
    valu_wave_instr = sum(per[k]["valu"] * steps[k] / 64.0 for k in steps)
    lane_steps = sum(steps.values())
    issue_s = valu_wave_instr * 2.0 / (simds * clock)                     # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles
    achieved = lane_steps / (ent_ms * 1e-3)
    return {"bound": "issue", "unit": "G lane-steps/s", "achieved": round(achieved / 1e9, 1), "peak": round(lane_steps / issue_s / 1e9, 1),
            "frac": round(issue_s / (ent_ms * 1e-3), 4), "kernel_ms": round(ent_ms, 4),
            "symbols": int(symbols), "sub_sequences": int(subs), "walks": 4, "table_reads_per_symbol_with_pair_entries": round(reads / symbols, 3),
            "lane_steps": {k: int(v) for k, v in steps.items()},
            "valu_wave_instructions_per_64_lane_steps": {k: per[k]["valu"] for k in steps},
            "instr_per_symbol_all_walks": round(sum((per[k]["valu"] + per[k]["salu"] + per[k]["lds"]) * steps[k] for k in steps) / symbols, 1),
            # the stage's other yardstick: its compulsory HBM traffic -- the coefficient planes written once (zeros included) and the
            # compressed scans read by each of the three kernels -- against 8 TB/s in the same measured time
            "hbm": (lambda b: {"bytes": int(b), "unit": "GB/s", "achieved": round(b / (ent_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                               "frac": round(b / 8e12 / (ent_ms * 1e-3), 4),
                               "what": "coefficient planes out (24 883 200 B per 4K 4:2:0 file, SURVEY 8a a14) + 3 x the un-stuffed scan bytes in"})(
                n_files * 24_883_200.0 * (CFG4["in_w"] * CFG4["in_h"] / (3840.0 * 2160.0)) + 3.0 * subs * 128.0),
            "sampled_files": len(sample), "model_source": "static: profiles/entropy_issue_model.json (" + model["source"] + ")",
            "what": "the entropy decodes of one step, batches one after the other (hipEvents): four dense bit-serial walks of every sub-sequence "
                    "(speculative, from the predecessor's exit, count, write; the first three read pair entries).  frac = VALU issue cycles the "
                    "walks need on 1 024 SIMD-32s at 2.4 GHz / the measured time: what is left is dependent-chain latency (one table read per step), "
                    "lanes idle in divergent waves (43 - 50 of 64 active) and the write pass's 2 waves per SIMD"}


def cfg4_parity(torch, files, out_all, which):
    """After the timed region: the oracle's chain on some of this rank's files -- file -> coefficients (oracle/jpeg_oracle.c)
    -> 4/8 decode with the reference's compiled spatial luma scaler -> CPU resize -- against the bytes the last step left."""
    import numpy as np
    from oracle import oracle as O
    from tests import util as U
    ow, oh = CFG4["out_w"], CFG4["out_h"]
    equal, checked = True, []
    got_all = out_all.to_numpy()
    for i in sorted(set(which)):
        j = O.jpeg_read_coefficients(files[i])
        dec = O.jpeg_idct_color_scaled(j, 4, 2)
        exp = np.zeros((1, oh, U.stride_for(ow)), np.uint8)
        U.oracle_render(dec[None], CFG4["dec_w"], CFG4["dec_h"], exp, ow, oh, 0, 0, ow, oh)
        ok = bool(np.array_equal(got_all[i][:, :4 * ow], exp[0][:, :4 * ow]))
        equal = equal and ok
        checked.append(i)
    return {"frames": len(checked), "which": checked, "equal": equal, "against": "oracle chain (jpeg_oracle.c entropy + scaled IDCT, if_oracle.c resize), rank 0's files"}


def cfg4_selfcheck(torch, gathered, total, world, dev, out_all):
    """--selfcheck: the first frame every rank sent must equal that frame decoded HERE (rank 0 makes the same file again)."""
    import numpy as np
    from imageflow_amd.codecs import mozjpeg_decoder as D
    from imageflow_amd.graphics.bitmaps import Bitmap
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams
    from imageflow_amd.sharding import shard_range
    ow, oh = CFG4["out_w"], CFG4["out_h"]
    bad = []
    for r in range(world):
        lo = shard_range(total, r, world)[0]
        f = cfg4_files(lo, 1)
        ent = D.JpegEntropyBatch(f, device=str(dev))
        coef = ent.read_coefficients()
        stage = D.JpegPixelStage(CFG4["in_w"], CFG4["in_h"], 3, ent.h_samp, ent.v_samp, 1, device=str(dev), scale_num=4, luma_spatial=True, luma_srgb=True)
        qt = torch.from_numpy(ent.qt.view(np.int16)).to(dev)
        one = Bitmap.create_u8(1, ow, oh, dev)
        stage.read_frames_into(coef, qt, one, ScaleAndRenderParams(0, 0, ow, oh))
        torch.cuda.synchronize()
        if not bool(torch.equal(gathered[r, 0].to(one.data.device), one.data[0])):
            bad.append(r)
    return {"ranks": world, "first_frame_of_every_rank_equal": not bad, "ranks_that_differ": bad}


OTHER_CONFIGS = [        # name, what `bench.py` is asked for (the workloads' own sizes, reduced step counts)
    ("cfg5", ["--workload", "cfg5", "--steps", "30", "--warmup", "5"]),
    ("cfg3_job", ["--workload", "cfg3", "--steps", "40", "--warmup", "5"]),        # (fewer steps: the kernel probe runs before the clocks are up)
    ("cfg4", ["--workload", "cfg4", "--steps", "20", "--warmup", "4"]),
]


def other_configs(budget_s):
    """BASELINE configs 5, 3 and 4 beside the headline: each is THIS file run as a child process with `--workload ...`
    (exactly what a user would type; a child that fails or stalls costs its own entry, not the line), its JSON line cut
    down to what the driver needs to witness: time per step, the kernel-timed roofline fraction on SURVEY 8d's bytes, the
    parity stamp against the CPU oracle, and the workload it ran."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT")}
    out, t_start = {}, time.perf_counter()
    for name, extra in OTHER_CONFIGS:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20:
            out[name] = {"skipped": f"time budget of {budget_s:.0f} s used up"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-strong-field", "--no-other-configs"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=min(left, 100))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {r.stderr.strip()[-300:]}", "command": " ".join(cmd[1:])}
                continue
            j = json.loads(lines[-1])
            rf, pc = j.get("roofline", {}), j.get("parity_checked") or {}
            e = {"value": j.get("value"), "unit": j.get("unit"), "metric": j.get("metric"), "steps": j.get("steps"), "ms_per_step": j.get("ms_per_step"),
                 "roofline": {k: rf.get(k) for k in ("bound", "frac", "frac_timed", "kernel_ms", "achieved", "peak", "unit", "algorithmic_bytes_per_launch")},
                 "parity_checked": {"equal": pc.get("equal"), "frames": pc.get("frames"), "against": pc.get("against")},
                 "config": {"workload": j.get("config", {}).get("workload"), "kernel": j.get("config", {}).get("kernel")},
                 "command": "bench.py " + " ".join(extra), "wall_s": round(time.perf_counter() - t0, 1)}
            for k in ("files_per_s", "roofline_entropy"):
                if k in j:
                    e[k] = j[k]
            out[name] = e
        except subprocess.TimeoutExpired:
            out[name] = {"error": "timed out", "command": " ".join(cmd[1:])}
        except Exception as ex:  # noqa: BLE001
            out[name] = {"error": f"{type(ex).__name__}: {ex}", "command": " ".join(cmd[1:])}
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 0.3 s of GPU time: long enough that the clock ramp after the
    # idle barrier (the first ~20 launches run 3-5 % slower) does not colour the average
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--frames", type=int, default=None, help="weak scaling: frames per GPU (default: the workload's, 256 for cfg2)")
    ap.add_argument("--total-frames", type=int, default=1024, help="strong scaling: frames of the whole job")
    ap.add_argument("--pattern", default="mixed", choices=["mixed", "gradient", "random"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default=None, choices=["every", "final", "none"],
                    help="N > 1 only.  every (default): ONE RCCL gather of the batch's outputs to rank 0 PER STEP (a step is one "
                         "batch: SURVEY 8e 'one per batch'), asynchronous and double buffered so that it overlaps the next "
                         "batch's kernel; final: a single gather at the end of the timed region (amortised over the steps -- "
                         "not the north_star batch time); none: results stay sharded")
    ap.add_argument("--rccl-channels", type=int, default=8,
                    help="N > 1: cap on RCCL's channels (= workgroups of its kernels; NCCL_MAX_NCHANNELS / NCCL_MAX_P2P_NCHANNELS, set "
                         "before the process group is made unless the environment already names them; 0: leave RCCL alone).  7 peers x 20 MB per batch "
                         "need one channel each; the default RCCL set-up would take tens of CUs from a grid that wants all 256")
    ap.add_argument("--reserve-cus", default="auto",
                    help="N > 1 with --gather every: CUs the resample launches leave to the gather's workgroups "
                         "(ifhip_set_cu_budget(256 - this)).  auto (default): the warm-up measures a few batches WITH their gathers at 0 "
                         "and at --rccl-channels reserved CUs and the timed region uses the faster (an RCCL workgroup cannot share a CU "
                         "with a resample workgroup -- LDS and registers -- so an unreserved grid of exactly 256 workgroups waits for the "
                         "gather's CUs; a reserved one runs finer bands: profiles/r6_gather_overlap_emulation.jsonl has both sides)")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS) + ["cfg4"])
    ap.add_argument("--batches-in-flight", type=int, default=2, help="--workload cfg4: host threads / HIP streams, each decoding its batches of files one after the other")
    ap.add_argument("--files-per-batch", type=int, default=None, help="--workload cfg4: files per entropy / pixel-stage batch (default 64)")
    ap.add_argument("--selfcheck", dest="selfcheck", action="store_true", default=None,
                    help="after the timed region rank 0 checks the first gathered frame of EVERY rank against one it renders itself "
                         "(default: on whenever N > 1 and the outputs are gathered)")
    ap.add_argument("--no-selfcheck", dest="selfcheck", action="store_false")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default run (N = 1, cfg2) also measures BASELINE configs 5, 3 and 4 at reduced step counts, each as "
                         "`bench.py --workload ...` in a child process, and reports them under `other_configs`; this skips them")
    ap.add_argument("--other-configs-budget-s", type=float, default=150.0, help="wall-clock cap of that addition")
    ap.add_argument("--outputs", default="files", choices=["files", "bgra"],
                    help="--workload cfg3: what the job leaves and its final gather ships -- the four JPEG files per image "
                         "(libjpeg_turbo q90, coded on the device; the reference's export_4_sizes) or the raw BGRA levels")
    ap.add_argument("--no-strong-field", action="store_true",
                    help="skip the extra `strong_1024` measurement (the --total-frames job cut into N blocks) of a weak cfg2 run")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one.  One process per GPU under torch.distributed.run, rendezvous on 127.0.0.1.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to "
                         f"measure a different job than the one asked for")
    # IFHIP_BENCH_ONE_RANK_RCCL=1: development aid -- a process group of ONE rank over RCCL, so that the N > 1 code path (per-batch
    # asynchronous gathers, their stream waits, the reserve tuning, the selfcheck) runs against the real backend on a box with one
    # GPU (tests/test_gpu_process_group.py; the gather of a single rank moves nothing between devices: numbers mean nothing)
    one_rank_rccl = world == 1 and os.environ.get("IFHIP_BENCH_ONE_RANK_RCCL") == "1" and "MASTER_PORT" in os.environ
    grouped = world > 1 or one_rank_rccl
    if args.gather is None:
        args.gather = "every" if grouped else "none"
    if args.no_gather or not grouped:
        args.gather = "none"
    if args.selfcheck is None:
        args.selfcheck = grouped and args.gather != "none"
    # cfg4's input files are written on the host BEFORE the device is touched (the writers are forked workers)
    cfg4_inputs = None
    if args.workload == "cfg4":
        from imageflow_amd.sharding import shard_range as _sr
        if args.scaling == "strong":
            lo4, hi4 = _sr(args.total_frames, rank, world)
        else:
            per4 = args.frames or CFG4["files_per_gpu"]
            lo4, hi4 = rank * per4, (rank + 1) * per4
        if hi4 - lo4 < 1:
            raise SystemExit(f"rank {rank} owns no files")
        cfg4_inputs = (lo4, cfg4_files(lo4, hi4 - lo4))

    import torch
    import torch.distributed as dist

    from imageflow_amd import _native
    from imageflow_amd.graphics.bitmaps import Bitmap, BitmapCompositing
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, plan_for, scale_and_render, time_scale_and_render
    from imageflow_amd.graphics.weights import Filter
    from imageflow_amd.sharding import gather_bytes_to_root, gather_to_root, max_over_ranks, shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: imageflow_amd has no CPU path")
    # IFHIP_BENCH_DRYRUN_ONE_GPU=1: development aid -- run N ranks on ONE GPU over gloo to exercise the multi-rank control
    # flow where only a single GPU is available (never used by the driver; numbers from such a run mean nothing)
    dryrun = os.environ.get("IFHIP_BENCH_DRYRUN_ONE_GPU") == "1"
    if dryrun:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPUs visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = grouped
    rccl_env = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group("gloo")
        else:
            # RCCL's kernels take one workgroup per channel; its default set-up on an xGMI node opens dozens.  The job's only
            # collective is a gather of 20 MB per peer and batch: a handful of channels move that, and every CU RCCL does not
            # take stays with the resample grid (one workgroup per CU).  Names the environment already sets are left alone.
            if args.rccl_channels > 0:                       # (--rccl-channels 0: RCCL's own choice)
                for k in ("NCCL_MAX_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS"):
                    os.environ.setdefault(k, str(args.rccl_channels))
                os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
                os.environ.setdefault("NCCL_MIN_P2P_NCHANNELS", "1")
            rccl_env = {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS", "NCCL_MIN_P2P_NCHANNELS")}
            dist.init_process_group("nccl", device_id=dev)
    reserve_cus, reserve_candidates = 0, []
    if distributed and args.gather == "every":
        if args.reserve_cus == "auto":
            reserve_candidates = [0, max(1, min(128, args.rccl_channels or 8))]
        else:
            reserve_cus = max(0, min(128, int(args.reserve_cus)))
    if args.workload == "cfg4":
        run_cfg4(args, torch, dist, world, rank, local_rank, dev, dryrun, cfg4_inputs, rccl_env, grouped)
        if distributed:
            dist.destroy_process_group()
        return

    wl = WORKLOADS[args.workload]
    in_w, in_h, out_w, out_h = wl[0], wl[1], wl[2], wl[3]
    pyramid = args.workload == "cfg3"
    # this rank's block of the job
    if args.scaling == "strong":
        total = args.total_frames
        lo, hi = shard_range(total, rank, world)
    else:
        per = args.frames or wl[9]
        total = per * world
        lo, hi = rank * per, (rank + 1) * per
    n = hi - lo
    if n < 1:
        raise SystemExit(f"rank {rank} owns no frames ({total} frames over {world} ranks)")
    n_max = -(-total // world)                           # gather slots are sized by the largest block
    # the extra north_star measurement of a default run: --total-frames images cut into `world` blocks
    strong_field = args.scaling == "weak" and args.workload in ("cfg2", "cfg3") and not args.no_strong_field
    files_out = pyramid and args.outputs == "files"
    s_lo, s_hi = shard_range(args.total_frames, rank, world) if strong_field else (0, 0)
    n_strong = s_hi - s_lo
    if strong_field and n_strong < 1:
        strong_field = False
    n_alloc = max(n, n_strong)

    ranks_info = [{"rank": rank, "device": torch.cuda.get_device_name(local_rank), "local_rank": local_rank}]
    if distributed:
        objs = [None] * world
        dist.all_gather_object(objs, ranks_info[0])
        ranks_info = objs

    inp_all = make_frames(torch, n_alloc, lo, rank, dev, args.pattern, in_w, in_h)
    inp_all.alpha_meaningful = wl[6]
    if wl[6]:
        inp_all.data.view(n_alloc, in_h, -1)[:, :, 3::4] = torch.randint(0, 256, (n_alloc, in_h, inp_all.stride // 4), dtype=torch.uint8, device=dev)

    def first_frames(b, k):
        return Bitmap(b.data[:k], b.w, b.h, b.stride, b.alpha_meaningful, b.compose, b.matte)
    inp = first_frames(inp_all, n)
    mode = "none" if (not distributed or args.no_gather or os.environ.get("IFHIP_BENCH_GATHER", "1") == "0") else args.gather
    overlapped = mode == "every" and not files_out       # (files: the message's size is read on the host first, so that gather runs in line)
    gather_note = {"none": "none",
                   "every": ("rccl gather of the batch's outputs to rank 0 EVERY step (one per batch), asynchronous and double buffered: "
                             "the gather of batch k runs beside the kernel of batch k + 1" if overlapped else
                             "rccl gather of the batch's files to rank 0 EVERY step (one per batch), in line with the step (sizes first, then the bytes)"),
                   "final": "rccl gather of the outputs to rank 0 once, at the end of the timed region (the same gather ran once "
                            "during warm-up); amortised over the steps: not the per-batch figure"}[mode]
    gather_calls = {"warmup": 0, "timed": 0}

    class Job:
        """One measured job: `nj` frames on this rank out of `total_j`; owns its canvases and gather buffers."""

        def __init__(self, nj, total_j):
            self.n, self.total = nj, total_j
            self.n_max = -(-total_j // world)
            self.inp = first_frames(inp_all, nj)
            if pyramid:
                sizes = {name: (w, h) for _, name, w, h in PYRAMID}
                self.levels = {name: Bitmap.create_u8(self.n_max, w, h, dev) for name, (w, h) in sizes.items()}
                for b in self.levels.values():
                    b.data = b.data[:nj]
                self.chain = []
                for a, b, w, h in PYRAMID:
                    s = self.inp if a == "src" else self.levels[a]
                    self.chain.append((s, self.levels[b], ScaleAndRenderParams(0, 0, w, h), plan_for(s.w, s.h, w, h, Filter.Robidoux, 0.0, dev)))
                self.out_bytes_per_frame = sum(b.image_bytes for b in self.levels.values())
                self.enc = {}
                if files_out:
                    # MozjpegEncoder::write_frame behind every output (codecs/mozjpeg.rs:78-160, preset libjpeg_turbo q90,
                    # self_test.rs:186): 4:2:0, forward pixel stage + baseline entropy coder, files stay in HBM
                    import numpy as np
                    from imageflow_amd.codecs import mozjpeg as M
                    self.M = M
                    hs, vs = M.sampling_factors((2, 2), (2, 2))
                    self.qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(90)] * nj).view(np.int16)).to(dev)
                    for name, (w, h) in sizes.items():
                        fwd = M.JpegForwardStage(w, h, hs, vs, nj, dev)
                        coef = fwd.write_frames(self.levels[name], self.qt)
                        coder = M.JpegEntropyStage(w, h, hs, vs, fwd.blocks_w, fwd.blocks_h, nj, dev)
                        pitch = (3 * w * h + 4095) // 4096 * 4096 + 4096                 # three bytes per pixel: ten times a q90 photograph, room for noise
                        self.enc[name] = [fwd, coef, coder, torch.empty((nj, pitch), dtype=torch.uint8, device=dev), None, None,
                                          torch.empty(nj * pitch, dtype=torch.uint8, device=dev)]
                    self.file_bytes = None
                else:
                    # the four outputs of a frame, side by side; two of them when gathers overlap the next step
                    self.packed = [torch.empty((self.n_max, self.out_bytes_per_frame), dtype=torch.uint8, device=dev) for _ in range(2 if overlapped else 1)]
            else:
                self.canv = [Bitmap.create_u8(self.n_max, out_w, out_h, dev, compose=BitmapCompositing[wl[7]], matte=wl[8]) for _ in range(2)]
                self.views = [first_frames(c, nj) for c in self.canv]
                self.info = ScaleAndRenderParams(0, 0, out_w, out_h, wl[5], Filter[wl[4]])
                self.plan = plan_for(in_w, in_h, out_w, out_h, self.info.interpolation_filter, wl[5], dev)
                self.out_bytes_per_frame = self.canv[0].image_bytes
            self.gathered = None
            if mode != "none" and rank == 0 and not files_out:
                self.gathered = [torch.empty((world, self.n_max, self.out_bytes_per_frame), dtype=torch.uint8, device="cpu" if dryrun else dev)
                                 for _ in range(2 if overlapped else 1)]
            self.gathered_sizes = None
            self.pending = [None, None]
            self.host_copy = [None, None]                  # (dry run: the payload on the host while gloo gathers it)
            self.note = gather_note
            self.reserve, self.tuned = reserve_cus, {}     # CUs left to the gather's workgroups; ms per batch measured at each candidate
            self.gather_ok = True
            self.last = 0                                   # parity of the last step run (which canvas / gather buffer holds its outputs)

        def payload(self, i):
            """What step i's gather ships from this rank (device tensor, gather-slot sized)."""
            if not pyramid:
                return self.canv[i & 1].data
            pk = self.packed[i & 1 if overlapped else 0]
            off = 0
            for b in self.levels.values():                             # device-side packing: one message per rank
                pk[:self.n, off:off + b.image_bytes] = b.data
                off += b.image_bytes
            return pk

        def wait_slot(self, k):
            if self.pending[k] is not None:
                self.pending[k].wait()                      # (RCCL: the current stream waits; gloo: the host does)
                self.pending[k] = None

        def gather_step(self, i, phase, blocking):
            """The batch's one exchange: its finished outputs to rank 0.  The frames never meet before this point (no
            data-path collective).  A failure is reported in the line; it does not cost the compute measurement."""
            if not self.gather_ok:
                return
            try:
                if files_out:
                    msg, meta = self.packed_files()
                    meta_pad = torch.zeros((len(self.enc), self.n_max + 1), dtype=torch.int64, device=meta.device)
                    meta_pad[:, :meta.shape[1]] = meta
                    gather_to_root(meta_pad.cpu() if dryrun else meta_pad, 0)
                    self.gathered_sizes, _ = gather_bytes_to_root(msg.cpu() if dryrun else msg, 0)
                else:
                    k = (i & 1) if overlapped else 0
                    pay = self.payload(i)
                    if dryrun:
                        self.host_copy[k] = pay.cpu()
                        pay = self.host_copy[k]
                    out = self.gathered[k] if rank == 0 else None
                    if blocking:
                        gather_to_root(pay, 0, async_op=False, out=out)
                    else:
                        self.pending[k], _ = gather_to_root(pay, 0, async_op=True, out=out)
                gather_calls[phase] += 1
            except Exception as e:  # noqa: BLE001
                self.gather_ok = False
                self.note = f"rccl gather failed: {type(e).__name__}: {e}"

        def step(self, i, encode=True, phase=None):
            """One batch: the rank's frames through the hot path; with `phase` (and --gather every) its gather behind it."""
            gather = phase is not None and mode == "every"
            if gather and overlapped:
                self.wait_slot(i & 1)                      # the buffers step i is about to overwrite have been gathered (step i - 2)
            if pyramid:
                for s, d, inf, pl in self.chain:
                    scale_and_render(s, d, inf, plan=pl)
                if encode:
                    for name, e in self.enc.items():                   # (outputs have no meaningful alpha: no matte pass)
                        e[0].write_frames(self.levels[name], self.qt, e[1])
                        _, e[4], e[5] = e[2].encode_device(e[1], 90, files=e[3])
            else:
                scale_and_render(self.inp, self.views[i & 1], self.info, plan=self.plan)
            self.last = i & 1
            if gather:
                self.gather_step(i, phase, blocking=not overlapped)

        def sync_all(self):
            for k in range(2):
                self.wait_slot(k)
            torch.cuda.synchronize()

        def packed_files(self):
            """The job's outputs as one message: every level's files back to back (16-byte aligned starts), and the
            table that finds them -- per level the n + 1 offsets inside the level's part."""
            parts, meta = [], []
            for name, e in self.enc.items():
                parts.append(self.M.pack_files_device(e[3], e[4], out=e[6]))
            used = [int(o[-1].item()) for _, o in parts]                 # (one look at the device: the message's size)
            for (_, o) in parts:
                meta.append(o)
            self.file_bytes = sum(used)
            return torch.cat([p[:u] for (p, _), u in zip(parts, used)]), torch.stack(meta)

        def timed(self, steps, phase):
            """`steps` batches between barriers -> seconds on this rank.  phase None: no gathers at all."""
            # while gathers run beside the kernels the launches plan for the CUs RCCL's workgroups leave them
            _native.set_cu_budget(256 - self.reserve if (phase is not None and overlapped and self.reserve) else 0)
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                self.step(i, phase=phase)
            self.sync_all()
            t_steps = time.perf_counter() - t0
            if phase is not None and mode == "final":
                self.gather_step(steps - 1, phase, blocking=True)
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            _native.set_cu_budget(0)
            return dt, t_steps

        def measure(self, steps, warmup):
            """-> (whole-job seconds with the job's gathers, seconds of the same steps without any gather), both max over ranks.
            With gathers: every step's batch is followed by ITS gather (mode every) -- the timed region ends when the last
            batch's outputs are on rank 0 -- or the steps by one final gather (mode final).  Without: a second timed loop."""
            for i in range(warmup):
                self.step(i, phase="warmup")                # RCCL sets its peer-to-peer channels up lazily on the first send / recv
            self.sync_all()                                 # between two ranks (tens of ms): the warm-up steps gather as the timed ones do
            if overlapped and reserve_candidates:           # --reserve-cus auto: a few batches with their gathers at each setting
                k = max(2, min(steps, 10))
                for r in reserve_candidates:
                    self.reserve = r
                    self.tuned[r] = round(max_over_ranks(self.timed(k, "warmup")[0], dev) / k * 1e3, 4)
                self.reserve = min(self.tuned, key=self.tuned.get)      # (the same on every rank: the times are maxima over the ranks)
            if mode == "final":
                self.gather_step(max(warmup, 1) - 1, "warmup", blocking=True)
            elapsed, t_steps = self.timed(steps, "timed")
            if mode == "none":
                return max_over_ranks(elapsed, dev), max_over_ranks(elapsed, dev)
            if mode == "final":                              # the steps of the same loop, before its gather
                return max_over_ranks(elapsed, dev), max_over_ranks(t_steps, dev)
            plain, _ = self.timed(steps, None)
            return max_over_ranks(elapsed, dev), max_over_ranks(plain, dev)

    job = Job(n, total)
    elapsed, compute_s = job.measure(args.steps, args.warmup)
    resize_only_ms = None
    if files_out:                                   # what a step costs without the encoders, and what the job's files weigh
        k = max(1, min(args.steps, 20))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            job.step(i, encode=False)
        torch.cuda.synchronize()
        resize_only_ms = (time.perf_counter() - t0) / k * 1e3
        if job.file_bytes is None:
            job.packed_files()
        job_dropped = int(sum(int((e[5] != 0).sum().item()) for e in job.enc.values()))
    job_file_bytes, job_gathered_sizes = (job.file_bytes, job.gathered_sizes) if files_out else (None, None)
    views, chain = (None, job.chain) if pyramid else (job.views, None)
    info, plan = (None, None) if pyramid else (job.info, job.plan)
    gather_note = job.note
    job_reserve, job_tuned = job.reserve, {str(k): v for k, v in job.tuned.items()}
    main_gather_calls = dict(gather_calls)

    strong = None
    if strong_field:
        gather_calls.update(warmup=0, timed=0)
        sj = Job(n_strong, args.total_frames)
        s_steps = max(1, min(args.steps, 20))
        s_elapsed, s_compute = sj.measure(s_steps, max(1, min(args.warmup, 3)))
        strong = {"total_frames": args.total_frames, "frames_per_gpu": n_strong, "steps": s_steps,
                  "ms_per_step": round(s_elapsed / s_steps * 1e3, 4),
                  "value": round(args.total_frames * in_w * in_h / 1e6 * s_steps / s_elapsed, 1),
                  "value_without_gather": round(args.total_frames * in_w * in_h / 1e6 * s_steps / s_compute, 1),
                  "gather_ms": round((s_elapsed - s_compute) / s_steps * 1e3, 4), "unit": "MP/s", "scaling": "strong", "gather": sj.note,
                  "gathers": dict(gather_calls), "reserved_cus_while_gathering": sj.reserve if overlapped else 0,
                  "reserve_cus_tried_ms_per_batch": {str(k): v for k, v in sj.tuned.items()} or None,
                  **({"gathered_bytes_per_rank": sj.gathered_sizes} if files_out else {}),
                  "what": f"north_star job: a batch of {args.total_frames} images cut into {world} contiguous blocks, same kernel, same timing "
                          f"rule: ms_per_step = one batch INCLUDING its gather to rank 0 (one gather per batch; `gathers.timed` = steps), "
                          f"value_without_gather = the same batches with no gather at all, gather_ms = what the gather adds per batch "
                          f"after overlap.  Speed-up of the batch over 1 GPU = this value at N / this value at N = 1 (no gather there)"}
        del sj

    # dominant-kernel duration: hipEvents on the launch stream around back-to-back launches of the same op
    launches = max(20, min(args.steps, 50))

    def probe_once():
        if pyramid:
            return [time_scale_and_render(s, d, inf, launches=launches, plan=pl) for s, d, inf, pl in chain]
        return [time_scale_and_render(inp, views[0], info, launches=launches, plan=plan)]
    # The probe follows the timed job; when that job was short (a reduced run: the `other_configs` children, tests) the clocks
    # are still ramping -- measured on cfg3: level 0, probed first, 1.18 ms behind a 150 ms job and 1.10 behind a 700 ms one -- so
    # untimed passes of the same launches first bring the work done before the probe to ~0.6 s of GPU time.
    warmed, probe_warm_passes = elapsed, 0
    while warmed < 0.6 and probe_warm_passes < 50:
        warmed += sum(probe_once()) * launches * 1e-3
        probe_warm_passes += 1
    if pyramid:
        level_ms = probe_once()
        kernel_ms = sum(level_ms)
        algo_bytes = n * PYRAMID_BYTES_PER_IMAGE
        kernel_name = "fused_resample_kernel x4 (levels %s ms)" % "/".join(f"{m:.3f}" for m in level_ms)
    else:
        kernel_ms = probe_once()[0]
        algo_bytes = n * (in_w * in_h * 4 + out_w * out_h * 4)
        kernel_name = "fused_resample_kernel" if plan.kernel_kind(wl[6]) == 0 else "two-pass (banded_resample_kernel where a band's rows fit the LDS, else the generic pair)"
    torch.cuda.synchronize()

    parity, selfcheck = None, None
    if rank == 0:
        try:
            # (the kernel-duration probe above rewrote views[0] / the levels with the same inputs: same bytes as the last step's)
            parity = pyramid_parity(torch, job.inp, job.levels) if pyramid else resample_parity(torch, wl, job.inp, job.views[(args.steps - 1) & 1], [0, n - 1])
        except Exception as e:  # noqa: BLE001
            parity = {"frames": 0, "equal": None, "error": f"{type(e).__name__}: {e}"}
        if args.selfcheck and job.gathered is not None and not pyramid:
            # the first frame every rank sent must equal that frame made and rendered HERE: a gather that lands at a wrong
            # offset, or a rank that rendered another block than shard_range gave it, shows the first time RCCL really runs
            bad = []
            for r in range(world):
                r_lo = shard_range(total, r, world)[0] if args.scaling == "strong" else r * n
                one_in = make_frames(torch, 1, r_lo, r, dev, "gradient", in_w, in_h)
                one_in.alpha_meaningful = wl[6]
                one_out = Bitmap.create_u8(1, out_w, out_h, dev, compose=BitmapCompositing[wl[7]], matte=wl[8])
                scale_and_render(one_in, one_out, info, plan=plan)
                torch.cuda.synchronize()
                if not bool(torch.equal(job.gathered[((args.steps - 1) & 1) if overlapped else 0][r, 0].to(dev), one_out.data[0])):
                    bad.append(r)
            selfcheck = {"ranks": world, "first_frame_of_every_rank_equal": not bad, "ranks_that_differ": bad,
                         "note": "needs --pattern gradient or mixed (frame 0 of a block is a gradient frame) and alpha not meaningful"}

    if rank == 0:
        mp_per_step = total * in_w * in_h / 1e6
        value = mp_per_step * args.steps / elapsed
        achieved = algo_bytes / (kernel_ms * 1e-3)
        # HBM bytes per launch: NOT measured in this run -- the figure of the separate rocprofv3 --pmc passes kept under
        # profiles/ (FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections), quoted for the default job only
        traffic, traffic_source = None, None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic_cfg2.json")))
            if n == FRAMES_PER_GPU and args.workload == "cfg2":
                traffic = t["traffic_bytes_per_launch"]
                traffic_source = "static: profiles/traffic_cfg2.json (" + t.get("source", "rocprofv3 --pmc passes") + ")"
        except Exception:  # noqa: BLE001
            pass
        measured_read = None    # same-process yardstick: a read-only streaming kernel over 4 GiB (16-byte nt loads)
        try:
            import ctypes
            from imageflow_amd import _native
            bps = ctypes.c_double(0.0)
            _native.check(_native.lib().ifhip_measure_read_bandwidth(4 << 30, 5, ctypes.byref(bps)))
            measured_read = bps.value
        except Exception:  # noqa: BLE001
            pass
        # ... and, for workloads whose canvas stores are a real share of the bytes (every shape but the thumbnails), the same
        # with writes mixed in at that share: reads + writes run at 5.3 - 5.5 TB/s on MI355X where reads alone reach 7
        measured_mix, write_share = None, None
        try:
            wbytes = (n * PYRAMID_WRITE_BYTES_PER_IMAGE) if pyramid else n * out_w * out_h * 4
            write_share = wbytes / algo_bytes
            if write_share >= 0.02 and hasattr(_native.lib(), "ifhip_measure_mixed_bandwidth"):
                every = max(1, round((algo_bytes - wbytes) / wbytes))
                bps = ctypes.c_double(0.0)
                _native.lib().ifhip_measure_mixed_bandwidth.argtypes = [ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
                _native.check(_native.lib().ifhip_measure_mixed_bandwidth(4 << 30, every, 5, ctypes.byref(bps)))
                measured_mix = bps.value
        except Exception:  # noqa: BLE001
            pass
        shape = (f"{in_w}x{in_h} -> 1600x900 -> {{1200x675 -> 400x225, 800x450}} (export_4_sizes), four chained launches"
                 if pyramid else f"{in_w}x{in_h} BGRA8 -> {out_w}x{out_h}")
        out = {
            "metric": "megapixels/sec resize (4K->200px Robidoux)", "value": round(value, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "gather_ms": round((elapsed - compute_s) / (args.steps if mode == "every" else 1) * 1e3, 4),
            "value_without_gather": round(mp_per_step * args.steps / compute_s, 1),
            "config": {"workload": f"BASELINE {args.workload}: {total} frames ({n} on rank 0) {shape} {wl[4]}"
                                   f"{' sharpen ' + str(wl[5]) if wl[5] else ''}, linear light, {wl[7]}, "
                                   f"alpha {'meaningful' if wl[6] else 'not meaningful'}, device resident, pattern={args.pattern}",
                       "frames_per_gpu": n, "total_frames": total, "kernel": kernel_name, "gather": gather_note,
                       "gathers": main_gather_calls, "gather_mode": mode,
                       "gather_ms_is": ("per step: (time of the steps with their gathers - time of the same steps without) / steps" if mode == "every"
                                        else "the one gather at the end of the timed region"),
                       "rccl_channels": rccl_env, "reserved_cus_while_gathering": job_reserve if overlapped else 0,
                       "reserve_cus_tried_ms_per_batch": job_tuned or None,
                       "rccl_ranks": dist.get_world_size() if distributed else 1,
                       "backend": (dist.get_backend() if distributed else "none"), "ranks": ranks_info},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4),
                         # the same bytes over the TIMED loop's step time (launch gaps, clock ramp and, at N > 1, the gather included)
                         "frac_timed": round(algo_bytes / (compute_s / args.steps) / HBM_PEAK, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                         "probe_launches": launches, "probe_warm_passes": probe_warm_passes,
                         "measured_read_GBps": round(measured_read / 1e9, 1) if measured_read else None,
                         "frac_of_measured_read": round(achieved / measured_read, 4) if measured_read else None,
                         "write_share": round(write_share, 4) if write_share is not None else None,
                         "measured_mix_GBps": round(measured_mix / 1e9, 1) if measured_mix else None,
                         "frac_of_measured_mix": round(achieved / measured_mix, 4) if measured_mix else None},
        }
        if pyramid:
            out["config"]["outputs"] = ("four JPEG files per image (libjpeg_turbo q90 4:2:0: forward pixel stage + entropy coder on the device), "
                                        "the final gather ships the files" if files_out else "four raw BGRA levels per image, gathered as they are")
            if files_out:
                out["config"]["file_bytes_per_image"] = int(job_file_bytes // max(n, 1)) if job_file_bytes else None
                out["config"]["gathered_bytes_per_rank"] = job_gathered_sizes
                out["config"]["dropped_files"] = job_dropped
                out["config"]["bgra_bytes_per_rank_if_raw"] = n * job.out_bytes_per_frame
                out["config"]["resize_only_ms_per_step"] = round(resize_only_ms, 4)
                out["config"]["note"] = ("value counts source megapixels through the WHOLE job (resizes + encoders); roofline is the four "
                                         "resample launches alone")
        out["parity_checked"] = parity
        if selfcheck is not None:
            out["selfcheck"] = selfcheck
        if strong is not None:
            out["strong_1024"] = strong
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg2":
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "MP/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
            try:
                out["host_dropin"] = host_dropin_rate(torch, seconds=2.0)
            except Exception as e:  # noqa: BLE001
                out["host_dropin"] = {"images_per_s": None, "what": f"failed: {e}"}
        if world == 1 and args.workload == "cfg2" and not args.no_other_configs and args.scaling == "weak" and args.frames is None:
            out["other_configs"] = other_configs(args.other_configs_budget_s)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
