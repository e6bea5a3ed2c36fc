"""Multi-GPU layout of a batch job: frames are independent, so a batch is cut into contiguous
blocks of frames, one block per rank (one process per GPU); there is no collective on the data path.  The only
exchange is the optional gather of the finished (small) output frames, one all_gather per batch over RCCL
(`backend="nccl"` on ROCm) -- or gloo on CPU tensors in the tests."""
import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int):
    """Frames [lo, hi) owned by `rank`: frame i lives on rank floor(i * world / n_frames) (contiguous blocks whose
    sizes differ by at most one)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = -(-rank * n_frames // world)          # ceil(rank * n / world)
    hi = -(-(rank + 1) * n_frames // world)
    return lo, hi


def owner_of(frame: int, n_frames: int, world: int) -> int:
    return frame * world // n_frames


def gather_outputs(local: torch.Tensor, n_frames: int, group=None, async_op=False, out=None):
    """All-gather equally shaped per-frame outputs.  `local` is [n_local, ...]; ranks may hold n_local differing by one,
    so shards are padded to the largest shard for the collective and trimmed afterwards.
    Returns the full [n_frames, ...] tensor (or (work, finish) when async_op, where finish() returns it)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    biggest = -(-n_frames // world)
    lo, hi = shard_range(n_frames, rank, world)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    if local.shape[0] < biggest:
        pad = torch.zeros((biggest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if out is None:
        out = torch.empty((world, biggest) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out.view(-1), local.view(-1), group=group, async_op=async_op)

    def finish():
        parts = []
        for r in range(world):
            a, b = shard_range(n_frames, r, world)
            parts.append(out[r, : b - a])
        return torch.cat(parts, 0)

    if async_op:
        return work, finish
    return finish()


def _global_rank(group, group_rank: int) -> int:
    """torch.distributed's point-to-point and rooted calls take GLOBAL ranks; inside a sub-group the two numberings differ."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def gather_to_root(local: torch.Tensor, root: int = 0, group=None, async_op=False, out=None):
    """The job's final gather (BASELINE north_star): equally shaped shards [n_local, ...] -> the group's rank `root`
    receives [world, n_local, ...]; every other rank only sends.  Over RCCL this is one direct xGMI transfer per peer (7
    links into the root in parallel), 1/8 of an all_gather's traffic.  `root` is a rank OF THE GROUP (0 = its first
    member).  Returns (work_or_None, out_or_None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local.contiguous()
    gather_list = None
    if rank == root:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        gather_list = [out[r] for r in range(world)]
    work = dist.gather(local, gather_list, dst=_global_rank(group, root), group=group, async_op=async_op)
    return (work if async_op else None), (out if rank == root else None)


def gather_bytes_to_root(local: torch.Tensor, root: int = 0, group=None, out=None):
    """The final gather of a job whose outputs are FILES (export_4_sizes ends in four JPEGs per image,
    imageflow_tool/src/self_test.rs:185-198): every rank holds one 1-D uint8 message of its own length (its files packed
    back to back, codecs.mozjpeg.pack_files_device).  The lengths travel first (one small all_gather, read with ONE
    device-to-host copy), then every peer sends exactly its bytes to the group's rank `root` -- grouped point-to-point
    transfers (batch_isend_irecv on both sides), over RCCL one direct xGMI link per peer.
    Returns (sizes [world] as a list, received) where received is, on the root, the list of the ranks' messages (views
    into `out` when given: a uint8 buffer of at least the sum of the sizes), elsewhere None."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local.contiguous().view(-1)
    mine = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    each = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(each, mine, group=group)
    sizes = [int(v) for v in each.tolist()]
    if rank != root:
        if sizes[rank]:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, _global_rank(group, root), group)]):
                w.wait()
        return sizes, None
    total = sum(sizes)
    if out is None:
        out = torch.empty(total, dtype=torch.uint8, device=local.device)
    if out.numel() < total:
        raise ValueError(f"gather buffer of {out.numel()} bytes for {total} bytes of files")
    parts, ops, at = [], [], 0
    for r in range(world):
        view = out[at:at + sizes[r]]
        at += sizes[r]
        parts.append(view)
        if r == rank:
            view.copy_(local)
        elif sizes[r]:
            ops.append(dist.P2POp(dist.irecv, view, _global_rank(group, r), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return sizes, parts


def max_over_ranks(seconds: float, device) -> float:
    """bench.py's timing rule: the job takes as long as its slowest rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
