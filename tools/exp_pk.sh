#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py -x -q 2>&1 | tail -2
for WL in cfg2 cfg2-alpha cfg5 cfg3-l0; do
  python tools/exp_variants.py --reps 2 --workload $WL _pd4nd8
done
