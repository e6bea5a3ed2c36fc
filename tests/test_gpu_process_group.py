"""The N > 1 path with the HIP kernel under a process group: 2 and 3 ranks on the test box's one GPU, each
rendering its shard_range block with the fused kernel, outputs gathered to rank 0, compared with the single-process HIP
result and with the oracle.  Plus bench.py's own launcher: `--gpus 2` with no torch.distributed environment must start
2 ranks by itself and say so in its JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from tests import util as U  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_render_equals_single_process(tmp_path, world):
    n_frames, in_w, in_h, ow, oh = 7, 960, 540, 50, 50
    out = str(tmp_path / "gathered.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "pg_worker.py"), out] + [str(v) for v in (n_frames, in_w, in_h, ow, oh)]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    frames = np.concatenate([U.random_frames(1, in_w, in_h, seed0=4000 + i, alpha=False) for i in range(n_frames)])
    inp = Bitmap.from_numpy(frames, in_w, in_h, frames.shape[2], "cuda:0")
    can = Bitmap.create_u8(n_frames, ow, oh, "cuda:0")
    scale_and_render(inp, can, ScaleAndRenderParams(0, 0, ow, oh))
    torch.cuda.synchronize()
    single = can.data.cpu().numpy()
    assert np.array_equal(got, single)
    exp = np.zeros((n_frames, oh, U.stride_for(ow)), np.uint8)
    U.oracle_render(frames, in_w, in_h, exp, ow, oh, 0, 0, ow, oh)
    assert np.array_equal(got.reshape(exp.shape), exp)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_export_job_gathers_the_same_files_as_one_process(tmp_path, world):
    """The cfg3 form of the final gather: every rank resizes its block to each size, codes every output as a JPEG on the
    device (libjpeg_turbo q90) and sends ONE packed message; rank 0 ends with exactly the files a single process writes --
    which are the files libjpeg-turbo reads back as the oracle's pixels + encoder."""
    import io
    import pickle
    from tests import pg_worker as W
    n_frames = 5
    out = str(tmp_path / "files.pkl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "pg_worker.py"), "files", out, str(n_frames)]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = pickle.load(open(out, "rb"))
    in_w, in_h, sizes = 640, 360, [(400, 225), (200, 113)]
    frames = np.concatenate([U.random_frames(1, in_w, in_h, seed0=7000 + i, alpha=False) for i in range(n_frames)])
    msg, metas, part_sizes = W.export_files(frames, in_w, in_h, sizes)
    single = W.unpack_files(msg, metas, part_sizes)
    assert len(got["message_bytes"]) == world and sum(got["message_bytes"]) >= sum(len(f) for fs in single for f in fs)
    PIL = pytest.importorskip("PIL.Image")
    for k, (w, h) in enumerate(sizes):
        assert len(got["files"][k]) == n_frames
        for i in range(n_frames):
            assert got["files"][k][i] == single[k][i], (k, i)
            im = PIL.open(io.BytesIO(got["files"][k][i]))
            assert im.size == (w, h) and im.format == "JPEG"
    # ... and each is the file libjpeg-turbo itself writes from the ORACLE's resize of that frame (quality 90, 4:2:0, Annex K tables)
    for i in (0, n_frames - 1):
        exp = np.zeros((1, 225, U.stride_for(400)), np.uint8)
        U.oracle_render(frames[i:i + 1], in_w, in_h, exp, 400, 225, 0, 0, 400, 225)
        rgb = np.ascontiguousarray(exp[0, :, :1600].reshape(225, 400, 4)[:, :, 2::-1])
        b = io.BytesIO()
        PIL.fromarray(rgb).save(b, "JPEG", quality=90, subsampling="4:2:0", optimize=False)
        assert got["files"][0][i] == b.getvalue(), i


@pytest.mark.parametrize("extra,scaling,total", [(["--total-frames", "9"], "weak", 16), (["--scaling", "strong", "--total-frames", "9"], "strong", 9),
                                                 (["--workload", "cfg3", "--frames", "3", "--total-frames", "5"], "weak", 6),
                                                 (["--total-frames", "9", "--gather", "final"], "weak", 16)])
def test_bench_launches_its_own_ranks(extra, scaling, total):
    """N > 1: ONE gather per batch (= per step) by default, SURVEY 8e -- `gathers.timed == steps` -- and the first gathered frame
    of every rank is checked on rank 0 without being asked for; `--gather final` keeps the single amortised gather."""
    env = _clean_env()
    env["IFHIP_BENCH_DRYRUN_ONE_GPU"] = "1"
    steps = 3
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline"]
    if "--frames" not in extra and scaling == "weak":
        cmd += ["--frames", "8"]
    r = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["config"]["total_frames"] == total, j
    assert "failed" not in j["config"]["gather"], j["config"]["gather"]
    assert j["value"] > 0 and j["roofline"]["frac"] > 0 and j["roofline"]["frac_timed"] > 0
    final = "final" in extra
    # every batch is followed by its gather (the warm-up steps too: RCCL's lazy channel set-up is not timed)
    tuned = not final and "--workload" not in extra                  # --reserve-cus auto: 3 more warm-up batches at each of its two settings
    want = {"warmup": 1 + (6 if tuned else 0), "timed": 1 if final else steps}
    assert j["config"]["gathers"] == want and j["config"]["gather_mode"] == ("final" if final else "every"), j["config"]
    if tuned:
        tried = j["config"]["reserve_cus_tried_ms_per_batch"]
        assert sorted(tried) == ["0", "8"] and j["config"]["reserved_cus_while_gathering"] == int(min(tried, key=tried.get)), j["config"]
    assert j["config"]["rccl_ranks"] == 2 and len(j["config"]["ranks"]) == 2 and j["config"]["ranks"][1]["rank"] == 1
    assert j["value_without_gather"] > 0 and j["steps"] == steps
    if "--workload" not in extra:                                    # checked without --selfcheck on the command line
        assert j["selfcheck"]["first_frame_of_every_rank_equal"] is True and j["selfcheck"]["ranks"] == 2, j.get("selfcheck")
    if scaling == "weak" and "--workload" not in extra:              # the north_star job rides along with the default run
        st = j["strong_1024"]
        assert st["total_frames"] == 9 and st["frames_per_gpu"] == 5 and st["gathers"] == want and st["value"] > 0
        assert "one gather per batch" in st["what"]
    elif "--workload" in extra:                                      # cfg3: the job's outputs are FILES, and so is what the gather ships
        c = j["config"]
        assert "JPEG files" in c["outputs"] and c["dropped_files"] == 0
        sent = c["gathered_bytes_per_rank"]
        assert len(sent) == 2 and all(0 < b < c["bgra_bytes_per_rank_if_raw"] // 2 for b in sent), c
        assert c["file_bytes_per_image"] > 0 and abs(sent[0] - 3 * c["file_bytes_per_image"]) <= 3 * 4 * 16 and c["resize_only_ms_per_step"] > 0
        st = j["strong_1024"]
        assert st["gathers"] == want and len(st["gathered_bytes_per_rank"]) == 2
    else:
        assert "strong_1024" not in j


def test_bench_refuses_a_mismatched_launcher():
    env = _clean_env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": str(_free_port())})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


@pytest.mark.parametrize("world,extra,total", [(2, ["--frames", "5"], 10), (3, ["--scaling", "strong", "--total-frames", "10"], 10)])
def test_bench_cfg4_shards_files_over_ranks_and_checks_itself(world, extra, total):
    """BASELINE config 4 as `bench.py --workload cfg4 --gpus N` (files -> GPU entropy decode -> 4/8 pixel stage -> 800x450, sharded
    by shard_range, the 800x450 outputs gathered once): 2 ranks weak and 3 ranks strong (blocks of 4 / 3 / 3 files) on the test
    box's one GPU.  The line must carry the chain's parity stamp against the oracle and, with --selfcheck, rank 0's check of the
    first gathered frame of EVERY rank against the same file decoded locally."""
    env = _clean_env()
    env["IFHIP_BENCH_DRYRUN_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg4", "--gpus", str(world), "--steps", "2", "--warmup", "1",
           "--batches-in-flight", "2", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == world and j["config"]["total_files"] == total and "cfg4" in j["config"]["workload"], j["config"]
    assert j["config"]["gathers"] == {"warmup": 1, "timed": 2} and "failed" not in j["config"]["gather"]      # one per batch
    assert j["config"]["one_call_chain"] is True and j["config"]["batches_in_flight"] == 2
    assert j["parity_checked"]["equal"] is True and j["parity_checked"]["frames"] == 2
    assert j["selfcheck"] == {"ranks": world, "first_frame_of_every_rank_equal": True, "ranks_that_differ": []}
    assert j["value"] > 0 and j["files_per_s"] > 0 and 0 < j["roofline"]["frac"] < 1
    assert j["roofline"]["algorithmic_bytes_per_launch"] == j["config"]["files_per_gpu"] * 26_323_584


def test_bench_selfcheck_and_parity_stamp_of_the_default_workload():
    env = _clean_env()
    env["IFHIP_BENCH_DRYRUN_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "6", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-strong-field", "--no-selfcheck"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["parity_checked"]["equal"] is True and j["parity_checked"]["which"] == [0, 5]
    assert "selfcheck" not in j                                      # (--no-selfcheck skips what is otherwise on for N > 1)


def test_rccl_itself_runs_the_gathers_on_a_one_rank_group():
    """One GPU per test box: a group of ONE rank is the only way RCCL (backend "nccl") executes here at all.  The library loads,
    a communicator is made under the channel caps bench.py sets, and the three calls of the N > 1 path -- the rooted gather of a
    device canvas (blocking and asynchronous), the byte gather of a packed file message, the max over ranks -- run on device
    tensors and deliver the bytes."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["IFHIP_ROOT"])
from imageflow_amd.sharding import gather_bytes_to_root, gather_to_root, max_over_ranks
for k in ("NCCL_MAX_NCHANNELS", "NCCL_MAX_P2P_NCHANNELS"):
    os.environ.setdefault(k, "8")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
local = torch.arange(3 * 166400, dtype=torch.int64, device=dev).remainder(251).to(torch.uint8).view(3, 166400)
_, out = gather_to_root(local, 0)
assert out.shape == (1, 3, 166400) and torch.equal(out[0], local)
buf = torch.zeros((1, 3, 166400), dtype=torch.uint8, device=dev)
work, out2 = gather_to_root(local, 0, async_op=True, out=buf)
work.wait()
torch.cuda.synchronize()
assert out2 is buf and torch.equal(buf[0], local)
msg = local.view(-1)[:100003]
sizes, parts = gather_bytes_to_root(msg, 0)
assert sizes == [100003] and torch.equal(parts[0], msg)
assert abs(max_over_ranks(1.25, dev) - 1.25) < 1e-12
dist.barrier()
dist.destroy_process_group()
print("rccl one-rank ok")
'''
    env = _clean_env()
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": str(_free_port()), "IFHIP_ROOT": ROOT})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("extra", [["--frames", "12", "--total-frames", "16"], ["--workload", "cfg3", "--frames", "3", "--total-frames", "4"]])
def test_bench_runs_its_per_batch_gathers_over_rccl_on_a_one_rank_group(extra):
    """bench.py's N > 1 code path against the REAL backend (IFHIP_BENCH_ONE_RANK_RCCL: a group of one rank over RCCL): every
    step's batch followed by its asynchronous gather, the stream waits that let step i + 2 reuse a canvas, the reserve tuning,
    the selfcheck on the gathered frame -- with device tensors and `dist.gather` on backend nccl."""
    env = _clean_env()
    env.update({"IFHIP_BENCH_ONE_RANK_RCCL": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": str(_free_port())})
    steps = 4
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline",
                        "--no-other-configs"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = j["config"]
    assert c["backend"] == "nccl" and c["rccl_ranks"] == 1 and c["gather_mode"] == "every" and "failed" not in c["gather"], c
    assert c["gathers"]["timed"] == steps and c["rccl_channels"]["NCCL_MAX_NCHANNELS"] == "8", c
    assert j["parity_checked"]["equal"] is True
    if "--workload" not in extra:
        assert j["selfcheck"]["first_frame_of_every_rank_equal"] is True
        assert sorted(c["reserve_cus_tried_ms_per_batch"]) == ["0", "8"]
        assert j["strong_1024"]["gathers"]["timed"] == j["strong_1024"]["steps"]
    else:
        assert len(c["gathered_bytes_per_rank"]) == 1 and c["gathered_bytes_per_rank"][0] > 0
