"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise the frame sharding, the gather of outputs
and bench.py's barrier / max-over-ranks timing rule.  The per-shard "compute" here is the CPU oracle (test
infrastructure standing in for the GPU kernel, which needs a device); what is under test is the distributed layout."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from imageflow_amd.sharding import gather_outputs, gather_to_root, max_over_ranks, owner_of, shard_range  # noqa: E402


def test_shard_range_partitions_every_batch():
    for n in (1, 2, 7, 8, 255, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            if world > n:
                continue
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert hi - lo in (n // world, n // world + 1) or n % world == 0
                seen += list(range(lo, hi))
                for i in range(lo, hi):
                    assert owner_of(i, n, world) == r
            assert seen == list(range(n))
    assert shard_range(1024, 3, 8) == (384, 512)        # BASELINE config 3: 128 frames per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from tests import util as U
        in_w, in_h, ow, oh = 192, 108, 20, 11
        lo, hi = shard_range(n_frames, rank, world)
        frames = np.concatenate([U.gradient_frames(1, in_w, in_h, k0=i) for i in range(lo, hi)]) if hi > lo else \
            np.zeros((0, in_h, U.stride_for(in_w)), np.uint8)
        cst = U.stride_for(ow)
        canv = np.zeros((hi - lo, oh, cst), np.uint8)
        U.oracle_render(frames, in_w, in_h, canv, ow, oh, 0, 0, ow, oh)
        local = torch.from_numpy(canv.reshape(hi - lo, -1))
        dist.barrier()
        full = gather_outputs(local, n_frames)
        work, finish = gather_outputs(local, n_frames, async_op=True)
        work.wait()
        full2 = finish()
        t = max_over_ranks(0.5 + rank, torch.device("cpu"))
        # gather-to-root with equal shards (what bench.py does): pad the local shard to the common size first
        biggest = -(-n_frames // world)
        padded = torch.zeros((biggest, local.shape[1]), dtype=local.dtype)
        padded[: local.shape[0]] = local
        work, out = gather_to_root(padded, 0, async_op=True)
        work.wait()
        root_ok = True
        if rank == 0:
            parts = [out[r, : shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0]] for r in range(world)]
            root_ok = bool(torch.equal(torch.cat(parts, 0), full))
        else:
            root_ok = out is None
        q.put((rank, full.numpy().copy(), bool(torch.equal(full, full2)) and root_ok, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 6), (2, 5), (3, 7)])
def test_two_rank_gloo_gather_equals_single_process(world, n_frames):
    from oracle import oracle as O  # noqa: F401  (build the checker before forking)
    from tests import util as U
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    in_w, in_h, ow, oh = 192, 108, 20, 11
    frames = U.gradient_frames(n_frames, in_w, in_h)
    exp = np.zeros((n_frames, oh, U.stride_for(ow)), np.uint8)
    U.oracle_render(frames, in_w, in_h, exp, ow, oh, 0, 0, ow, oh)
    for rank, full, same, t in results:
        assert np.array_equal(full.reshape(exp.shape), exp), rank     # every rank holds the whole job's output
        assert same
        assert abs(t - (0.5 + world - 1)) < 1e-9                      # max over ranks


def _bytes_worker(rank, world, port, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imageflow_amd.sharding import gather_bytes_to_root
        mine = torch.from_numpy(np.random.default_rng(900 + rank).integers(0, 256, sizes[rank], dtype=np.uint8))
        got_sizes, parts = gather_bytes_to_root(mine, 0)
        out = torch.empty(sum(sizes) + 7, dtype=torch.uint8)             # ... and into a caller's buffer
        got_sizes2, parts2 = gather_bytes_to_root(mine, 0, out=out if rank == 0 else None)
        ok = got_sizes == list(sizes) == got_sizes2
        if rank == 0:
            for r in range(world):
                exp = np.random.default_rng(900 + r).integers(0, 256, sizes[r], dtype=np.uint8)
                ok = ok and np.array_equal(parts[r].numpy(), exp) and np.array_equal(parts2[r].numpy(), exp)
            ok = ok and (sizes[0] == 0 or parts2[0].data_ptr() == out.data_ptr())
        else:
            ok = ok and parts is None and parts2 is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(1000, 37), (0, 4096), (16, 0, 100001), (5, 5, 5)])
def test_gather_of_variable_length_messages_to_the_root(sizes):
    """The cfg3 job's final gather ships FILES: every rank one message of its own length (also an empty one)."""
    world = len(sizes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bytes_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def _cfg4_gather_worker(rank, world, port, total, q):
    """The cfg4 job's message protocol without a GPU: every rank holds its shard_range block of fixed-size 800x450 outputs
    (here 8 x 6 stand-ins whose bytes name the file they belong to), padded to the largest block; ONE rooted gather."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, rank, world)
        n_max = -(-total // world)
        image_bytes = 8 * 6 * 4
        out_all = torch.zeros((n_max, image_bytes), dtype=torch.uint8)
        for i in range(lo, hi):
            out_all[i - lo] = torch.full((image_bytes,), i % 251, dtype=torch.uint8)
        gathered = torch.empty((world, n_max, image_bytes), dtype=torch.uint8) if rank == 0 else None
        _, got = gather_to_root(out_all, 0, out=gathered)
        ok = True
        if rank == 0:
            ok = got.data_ptr() == gathered.data_ptr()
            for r in range(world):
                a, b = shard_range(total, r, world)
                for i in range(a, b):
                    ok = ok and bool((got[r, i - a] == i % 251).all())                   # file i sits in rank owner_of(i)'s slot i - lo
                    ok = ok and owner_of(i, total, world) == r
        else:
            ok = got is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 10), (3, 10), (3, 4)])
def test_cfg4_final_gather_protocol(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cfg4_gather_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def _subgroup_worker(rank, world, port, q):
    """Gathers inside a SUB-group whose ranks are not the global ones (global ranks 1 and 2 of 3): the group's rank 0 is global
    rank 1 -- the rooted gather and the point-to-point file gather must translate group ranks to global ones."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imageflow_amd.sharding import gather_bytes_to_root
        group = dist.new_group([1, 2])
        ok = True
        if rank in (1, 2):
            g_rank = dist.get_rank(group)
            local = torch.full((2, 5), 10 + rank, dtype=torch.uint8)
            _, got = gather_to_root(local, 0, group=group)
            if g_rank == 0:
                ok = ok and bool((got[0] == 11).all()) and bool((got[1] == 12).all())
            else:
                ok = ok and got is None
            mine = torch.arange(3 + 4 * rank, dtype=torch.uint8) + rank
            sizes, parts = gather_bytes_to_root(mine, 0, group=group)
            ok = ok and sizes == [7, 11]
            if g_rank == 0:
                ok = ok and bool((parts[0] == torch.arange(7, dtype=torch.uint8) + 1).all()) and bool((parts[1] == torch.arange(11, dtype=torch.uint8) + 2).all())
        q.put((rank, bool(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gathers_inside_a_sub_group_use_global_ranks():
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results
