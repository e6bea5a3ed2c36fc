#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_bitmap_ops.py -x -q 2>&1 | tail -3
for WL in cfg2 cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize; do
  python tools/exp_variants.py --reps 1 --workload $WL IFHIP_ONE_FRAME_PER_WG=1
done
