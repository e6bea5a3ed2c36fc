#!/usr/bin/env python3
"""bench.py -- megapixels/s of the resample hot path (BASELINE.json metric) on N MI355X of one node.

Workload (config.workload): BASELINE config 2 -- a batch of 256 synthetic 3840x2160 BGRA8 frames per GPU, device
resident, each resized to 200x200 with Robidoux in linear light (ReplaceSelf canvas, alpha not meaningful, as after
a JPEG decode).  One "step" = one pass of the fused kernel over the rank's whole batch.  Weak scaling: every rank owns
its own 256 frames (independent images: no data-path collective); for N > 1 the 200x200 outputs of each step are
gathered to rank 0 over RCCL (one direct xGMI transfer per peer), asynchronously, overlapped with the next step.

Prints ONE JSON line (rank 0).  value = source megapixels resized per second over all ranks, inputs already in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IN_W, IN_H, OUT_W, OUT_H = 3840, 2160, 200, 200
FRAMES_PER_GPU = 256
ALGO_BYTES_PER_FRAME = IN_W * IN_H * 4 + OUT_W * OUT_H * 4        # 33,337,600 B
HBM_PEAK = 8.0e12                                                   # MI355X_MICROARCH.md: 8 TB/s spec

# Other BASELINE shapes, selectable with --workload (parity cases; the headline line stays cfg2)
WORKLOADS = {
    # name: (in_w, in_h, out_w, out_h, filter, sharpen, alpha_meaningful, compositing, matte, default frames)
    "cfg2": (3840, 2160, 200, 200, "Robidoux", 0.0, False, "ReplaceSelf", 0, 256),
    "cfg2-alpha": (3840, 2160, 200, 200, "Robidoux", 0.0, True, "ReplaceSelf", 0, 256),
    "cfg5": (7680, 4320, 400, 225, "Lanczos", 15.0, True, "BlendWithMatte", 0xFFFFFFFF, 64),
    "cfg3-l0": (3840, 2160, 1600, 900, "Robidoux", 0.0, False, "ReplaceSelf", 0, 256),
    # the remaining levels of the export_4_sizes pyramid (self_test.rs:185-198) and the cfg4 / cfg1 resizes
    "cfg3-l1": (1600, 900, 1200, 675, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg3-l2": (1600, 900, 800, 450, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg3-l3": (1200, 675, 400, 225, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg4-resize": (1920, 1080, 800, 450, "Robidoux", 0.0, False, "ReplaceSelf", 0, 1024),
    "cfg1-resize": (480, 270, 200, 113, "Robidoux", 0.0, False, "ReplaceSelf", 0, 4096),
}


def make_frames(torch, n, rank, device, pattern):
    """frame k pixel (x,y): B=(x+k)&255, G=(y+k)&255, R=(x+y+k)&255, A=255 (bench_graphics.rs:403-414 + offset);
    the random half is uniform bytes (worst case for LUT bank conflicts and rounding)."""
    from imageflow_amd.graphics.bitmaps import Bitmap, get_stride
    stride = get_stride(IN_W)
    data = torch.empty((n, IN_H, stride), dtype=torch.uint8, device=device)
    x = torch.arange(IN_W, device=device, dtype=torch.int32)[None, :]
    y = torch.arange(IN_H, device=device, dtype=torch.int32)[:, None]
    gen = torch.Generator(device=device)
    gen.manual_seed(1000 + rank)
    for i in range(n):
        k = rank * n + i
        use_random = pattern == "random" or (pattern == "mixed" and i >= n // 2)
        if use_random:
            data[i] = torch.randint(0, 256, (IN_H, stride), dtype=torch.uint8, device=device, generator=gen)
            data[i, :, 3::4] = 255
        else:
            px = data[i, :, : IN_W * 4].view(IN_H, IN_W, 4)
            px[..., 0] = ((x + k) & 255).to(torch.uint8)
            px[..., 1] = ((y + k) & 255).to(torch.uint8).expand(IN_H, IN_W)
            px[..., 2] = ((x + y + k) & 255).to(torch.uint8)
            px[..., 3] = 255
    return Bitmap(data.view(n, IN_H * stride), IN_W, IN_H, stride, alpha_meaningful=False)


def cpu_baseline(sample_seconds=15.0):
    """The oracle (our C port of the reference's CPU path; the Rust reference cannot be built here) timed on the
    host cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    from tests import util as U
    cores = os.cpu_count() or 1
    fr = U.gradient_frames(1, IN_W, IN_H)
    cst = U.stride_for(OUT_W)
    can = np.zeros((1, OUT_H, cst), np.uint8)
    t0 = time.perf_counter()
    O.scale_and_render_batch(fr.reshape(1, -1), can.reshape(1, -1), IN_W, IN_H, fr.shape[2], OUT_W, OUT_H, cst,
                             0, 0, OUT_W, OUT_H, n_threads=1)
    t1 = time.perf_counter() - t0
    n = min(4 * cores, 96)
    frames = np.concatenate([U.gradient_frames(n // 2, IN_W, IN_H), U.random_frames(n - n // 2, IN_W, IN_H, alpha=False)])
    cans = np.zeros((n, OUT_H * cst), np.uint8)
    flat = frames.reshape(n, -1)
    def one_pass():
        t0 = time.perf_counter()
        rc = O.scale_and_render_batch(flat, cans, IN_W, IN_H, frames.shape[2], OUT_W, OUT_H, cst, 0, 0, OUT_W, OUT_H,
                                      n_threads=cores)
        assert rc == 0
        return time.perf_counter() - t0
    first = one_pass()                                   # also warms the page cache / thread pool
    reps = int(max(1, min(200, sample_seconds / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    total = time.perf_counter() - t0
    mp = reps * n * IN_W * IN_H / 1e6
    return {"value": round(mp / total, 2), "unit": "MP/s", "cores": cores, "kind": "port",
            "single_thread_MPps": round(IN_W * IN_H / 1e6 / t1, 2),
            "sample": f"{reps} passes over {n} frames 3840x2160->200x200 Robidoux linear (half gradient, half random), "
                      f"{total:.1f} s wall, oracle/if_oracle.c -O3 x86-64-v3, {cores} OpenMP threads (one frame per thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # 0.3 s of GPU time: long enough that the clock ramp after the
    # idle barrier (the first ~20 launches run 3-5 % slower) does not colour the average
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (default 256 = BASELINE config 2)")
    ap.add_argument("--pattern", default="mixed", choices=["mixed", "gradient", "random"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="final", choices=["final", "every", "none"],
                    help="N > 1 only.  final: one RCCL gather of the outputs to rank 0 at the end of the timed region "
                         "(the job's final gather, BASELINE north_star); every: one per step, asynchronous and double "
                         "buffered against the next step; none: results stay sharded")
    ap.add_argument("--no-gather", action="store_true", help="same as --gather none")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    args = ap.parse_args()
    global IN_W, IN_H, OUT_W, OUT_H, ALGO_BYTES_PER_FRAME
    wl = WORKLOADS[args.workload]
    IN_W, IN_H, OUT_W, OUT_H = wl[0], wl[1], wl[2], wl[3]
    ALGO_BYTES_PER_FRAME = IN_W * IN_H * 4 + OUT_W * OUT_H * 4
    if args.workload != "cfg2" and args.frames == FRAMES_PER_GPU:
        args.frames = wl[9]

    import torch
    import torch.distributed as dist

    from imageflow_amd.graphics.bitmaps import Bitmap, BitmapCompositing
    from imageflow_amd.graphics.scaling import ScaleAndRenderParams, plan_for, scale_and_render, time_scale_and_render
    from imageflow_amd.graphics.weights import Filter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: imageflow_amd has no CPU path")
    # IFHIP_BENCH_DRYRUN_ONE_GPU=1: development aid -- run N ranks on ONE GPU over gloo to exercise the multi-rank control
    # flow where only a single GPU is available (never used by the driver; numbers from such a run mean nothing)
    dryrun = os.environ.get("IFHIP_BENCH_DRYRUN_ONE_GPU") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    n = args.frames
    inp = make_frames(torch, n, rank, dev, args.pattern)
    inp.alpha_meaningful = wl[6]
    if wl[6]:
        inp.data.view(n, IN_H, -1)[:, :, 3::4] = torch.randint(0, 256, (n, IN_H, inp.stride // 4), dtype=torch.uint8, device=dev)
    canv = [Bitmap.create_u8(n, OUT_W, OUT_H, dev, compose=BitmapCompositing[wl[7]], matte=wl[8]) for _ in range(2)]
    info = ScaleAndRenderParams(0, 0, OUT_W, OUT_H, wl[5], Filter[wl[4]])
    plan = plan_for(IN_W, IN_H, OUT_W, OUT_H, info.interpolation_filter, wl[5], dev)
    mode = "none" if (not distributed or args.no_gather or os.environ.get("IFHIP_BENCH_GATHER", "1") == "0") else args.gather
    gather = mode == "every"
    if dryrun and mode == "every":  # gloo cannot gather device tensors; the dry run only walks the final gather (via the host)
        raise SystemExit("dry run: pass --gather final or none")
    from imageflow_amd.sharding import gather_to_root, max_over_ranks
    gathered = [torch.empty((world,) + tuple(c.data.shape), dtype=torch.uint8, device="cpu" if dryrun else dev) if rank == 0 else None
                for c in canv] if mode != "none" else None
    gather_note = {"none": "none",
                   "every": "rccl gather of the outputs to rank 0 every step, asynchronous, double buffered",
                   "final": "rccl gather of the outputs to rank 0 once, at the end of the timed region"}[mode]

    def step(i, pending):
        c = canv[i & 1]
        if gather and pending[i & 1] is not None:
            pending[i & 1].wait()                  # the buffer we are about to overwrite has been gathered
            pending[i & 1] = None
        scale_and_render(inp, c, info, plan=plan)
        if gather:
            pending[i & 1], _ = gather_to_root(c.data, 0, async_op=True, out=gathered[i & 1])

    def sync_all(pending):
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()

    pending = [None, None]
    for i in range(args.warmup):
        step(i, pending)
    sync_all(pending)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, pending)
    sync_all(pending)
    if mode == "final":
        # The frames never meet before this point (no data-path collective).  A gather kernel running beside the
        # resample kernel would take CUs from a grid that is exactly one workgroup per CU, so the job's one exchange
        # happens after the last batch; a failure here is reported, it does not cost the measurement.
        try:
            last = canv[(args.steps - 1) & 1].data
            gather_to_root(last.cpu() if dryrun else last, 0, async_op=False, out=gathered[0])
        except Exception as e:  # noqa: BLE001
            gather_note = f"final rccl gather failed: {type(e).__name__}: {e}"
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dev)

    # dominant-kernel duration: hipEvents on the launch stream around back-to-back launches of the same op
    kernel_ms = time_scale_and_render(inp, canv[0], info, launches=max(5, min(args.steps, 50)), plan=plan)
    torch.cuda.synchronize()

    if rank == 0:
        mp_per_step = world * n * IN_W * IN_H / 1e6
        value = mp_per_step * args.steps / elapsed
        algo_bytes = n * ALGO_BYTES_PER_FRAME
        achieved = algo_bytes / (kernel_ms * 1e-3)
        traffic = None          # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/), if recorded
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic_cfg2.json")))
            if n == FRAMES_PER_GPU and args.workload == "cfg2":
                traffic = t["traffic_bytes_per_launch"]
        except Exception:
            pass
        measured_copy = None    # same-process calibration: device-to-device memcpy of 2 GiB, read + write bytes per second
        try:
            import ctypes
            from imageflow_amd import _native
            bps = ctypes.c_double(0.0)
            _native.check(_native.lib().ifhip_measure_copy_bandwidth(2 << 30, 5, ctypes.byref(bps)))
            measured_copy = bps.value
        except Exception:  # noqa: BLE001
            pass
        out = {
            "metric": "megapixels/sec resize (4K->200px Robidoux)", "value": round(value, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE {args.workload}: {n} x {IN_W}x{IN_H} BGRA8 frames per GPU -> {OUT_W}x{OUT_H} {wl[4]}"
                                   f"{' sharpen ' + str(wl[5]) if wl[5] else ''}, linear light, {wl[7]}, "
                                   f"alpha {'meaningful' if wl[6] else 'not meaningful'}, device resident, pattern={args.pattern}",
                       "frames_per_gpu": n, "kernel": "fused_resample_kernel" if plan.kernel_kind(wl[6]) == 0 else "generic",
                       "gather": gather_note},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic,
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                         "measured_copy_GBps": round(measured_copy / 1e9, 1) if measured_copy else None,
                         "frac_of_measured_copy": round(achieved / measured_copy, 4) if measured_copy else None},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "cfg2":
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "MP/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
