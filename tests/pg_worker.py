"""Worker of tests/test_gpu_process_group.py: one rank of a torch.distributed job that renders ITS block of a batch with
the HIP fused kernel and sends the outputs to rank 0 (imageflow_amd.sharding).  Both ranks share cuda:0 here (one GPU per
test box), so the process group is gloo and the gather goes through host tensors; the compute is the product path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from imageflow_amd.graphics.bitmaps import Bitmap  # noqa: E402
from imageflow_amd.graphics.scaling import ScaleAndRenderParams, scale_and_render  # noqa: E402
from imageflow_amd.sharding import gather_to_root, shard_range  # noqa: E402
from tests import util as U  # noqa: E402


def export_files(frames_np, in_w, in_h, sizes, dev="cuda:0"):
    """A miniature export job on this rank's frames: resample to every size, each output through the classic JPEG encoder
    (quality 90, 4:2:0, pixel stage + entropy coder on the device), the files packed into ONE message.
    -> (message uint8 cuda tensor, offsets int64 [len(sizes)][n + 1] cuda tensor)."""
    from imageflow_amd.codecs import mozjpeg as M
    n = frames_np.shape[0]
    inp = Bitmap.from_numpy(frames_np, in_w, in_h, frames_np.shape[2], dev)
    hs, vs = M.sampling_factors((2, 2), (2, 2))
    qt = torch.from_numpy(np.stack([M.quant_tables_for_quality(90)] * n).view(np.int16)).to(dev)
    parts, metas = [], []
    for (w, h) in sizes:
        out = Bitmap.create_u8(n, w, h, dev)
        scale_and_render(inp, out, ScaleAndRenderParams(0, 0, w, h))
        fwd = M.JpegForwardStage(w, h, hs, vs, n, dev)
        coder = M.JpegEntropyStage(w, h, hs, vs, fwd.blocks_w, fwd.blocks_h, n, dev)
        files, lengths, status = coder.encode_device(fwd.write_frames(out, qt), 90)
        assert int(status.abs().sum().item()) == 0
        packed, offsets = M.pack_files_device(files, lengths)
        parts.append(packed[: int(offsets[-1].item())])
        metas.append(torch.cat([offsets[:-1], lengths.to(torch.int64)]))     # where each file starts, and how long it is
    return torch.cat(parts), torch.stack(metas), [int(p.numel()) for p in parts]


def unpack_files(message, metas, part_sizes):
    """The files of one rank's message, [size][frame] -> bytes."""
    msg = message.cpu().numpy()
    out, base = [], 0
    for k, size in enumerate(part_sizes):
        m = metas[k].cpu().numpy()
        n = m.shape[0] // 2
        out.append([msg[base + int(m[i]): base + int(m[i]) + int(m[n + i])].tobytes() for i in range(n)])
        base += size
    return out


def main_files():
    """`pg_worker.py files out.pkl n_frames`: the cfg3 form of the job's exchange -- every rank's FILES to rank 0."""
    import pickle
    from imageflow_amd.sharding import gather_bytes_to_root
    out_path, n_frames = sys.argv[2], int(sys.argv[3])
    in_w, in_h, sizes = 640, 360, [(400, 225), (200, 113)]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    lo, hi = shard_range(n_frames, rank, world)
    frames = np.concatenate([U.random_frames(1, in_w, in_h, seed0=7000 + i, alpha=False) for i in range(lo, hi)])
    msg, metas, part_sizes = export_files(frames, in_w, in_h, sizes)
    torch.cuda.synchronize()
    info = [None] * world
    dist.all_gather_object(info, (metas.cpu(), part_sizes))
    sizes_seen, parts = gather_bytes_to_root(msg.cpu(), 0)
    if rank == 0:
        files = [[] for _ in sizes]
        for r in range(world):
            per_size = unpack_files(parts[r], info[r][0], info[r][1])
            for k in range(len(sizes)):
                files[k] += per_size[k]
        pickle.dump({"files": files, "message_bytes": sizes_seen}, open(out_path, "wb"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    if sys.argv[1] == "files":
        return main_files()
    out_path, n_frames, in_w, in_h, ow, oh = sys.argv[1], *map(int, sys.argv[2:7])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    lo, hi = shard_range(n_frames, rank, world)
    n_max = -(-n_frames // world)
    frames = np.concatenate([U.random_frames(1, in_w, in_h, seed0=4000 + i, alpha=False) for i in range(lo, hi)])
    inp = Bitmap.from_numpy(frames, in_w, in_h, frames.shape[2], "cuda:0")
    can = Bitmap.create_u8(n_max, ow, oh, "cuda:0")
    view = Bitmap(can.data[: hi - lo], ow, oh, can.stride)
    plan = scale_and_render(inp, view, ScaleAndRenderParams(0, 0, ow, oh))
    torch.cuda.synchronize()
    assert plan.kernel_kind() == 0, "expected the fused kernel"
    dist.barrier()
    _, full = gather_to_root(can.data.cpu(), 0)
    if rank == 0:
        parts = []
        for r in range(world):
            a, b = shard_range(n_frames, r, world)
            parts.append(full[r, : b - a].numpy())
        np.save(out_path, np.concatenate(parts))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
