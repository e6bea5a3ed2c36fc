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


def main():
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
