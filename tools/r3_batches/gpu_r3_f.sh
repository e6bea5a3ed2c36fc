#!/bin/bash
# round-3 GPU batch F: what bounds cfg5 -- rows in flight (12, 16), no horizontal chains, loads only
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/exp_variants.py --reps 2 --workload cfg5 _d12 _d16 _nochain _loadonly > gpurun_out/r3_f_variants.txt 2>&1
IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_d16.so python -m pytest tests/test_gpu_resample.py -x -q -m gpu -k "cfg5 or 7680 or matte or full" 2>&1 | tail -3 >> gpurun_out/r3_f_variants.txt
cat gpurun_out/r3_f_variants.txt
