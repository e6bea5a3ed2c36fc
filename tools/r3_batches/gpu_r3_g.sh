#!/bin/bash
# round-3 GPU batch G: gathers requested before / used after the multiply-adds (pipelined shapes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r3_g_variants.txt
python tools/exp_variants.py --reps 3 --workload cfg5 _split6 >> gpurun_out/r3_g_variants.txt 2>&1
python tools/exp_variants.py --reps 3 --workload cfg2 _split4 >> gpurun_out/r3_g_variants.txt 2>&1
python tools/exp_variants.py --reps 2 --workload cfg3-l0 _split4 >> gpurun_out/r3_g_variants.txt 2>&1
IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_split6.so python -m pytest tests/test_gpu_resample.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/r3_g_variants.txt
cat gpurun_out/r3_g_variants.txt
