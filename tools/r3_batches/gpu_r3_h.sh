#!/bin/bash
# round-3 GPU batch H: rows in flight on the narrow shape (12, 16 with the loop fully unrolled)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/exp_variants.py --reps 3 --workload cfg5 _d12 _d16 > gpurun_out/r3_h_variants.txt 2>&1
IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_d12.so python -m pytest tests/test_gpu_resample.py -x -q -m gpu -k "cfg5 or 7680 or matte or full" 2>&1 | tail -3 >> gpurun_out/r3_h_variants.txt
cat gpurun_out/r3_h_variants.txt
