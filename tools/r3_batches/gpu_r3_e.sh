#!/bin/bash
# round-3 GPU batch E: shim tests, 2-pixels-per-lane narrow shape A/B, extended issue-rate probe
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_abi_shim.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3_e_tests.log
IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_npx2.so python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3_e_tests_npx2.log
python tools/exp_variants.py --reps 3 --workload cfg5 _npx2 > gpurun_out/r3_e_variants.txt 2>&1
tools/probes/issue_rate_probe > gpurun_out/r3_e_issue_rate.txt 2>&1
tail -n 5 gpurun_out/r3_e_tests.log gpurun_out/r3_e_tests_npx2.log; cat gpurun_out/r3_e_variants.txt; grep "1024 lanes" gpurun_out/r3_e_issue_rate.txt | head -20
