#!/bin/bash
# round-3 GPU batch C: full GPU suite on the product library, parity of the matrix-pipe vertical pass (variant builds),
# A/B of the variants on the instruction-bound shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_c_tests.log
for v in mfma1 mfma2; do
  IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_$v.so python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3_c_tests_$v.log
done
: > gpurun_out/r3_c_variants.txt
for wl in cfg5 cfg2-alpha cfg2 cfg3-l0 cfg3-l1 cfg4-resize; do
  python tools/exp_variants.py --reps 2 --workload $wl _mfma1 _mfma2 >> gpurun_out/r3_c_variants.txt 2>&1
done
tail -4 gpurun_out/r3_c_tests.log gpurun_out/r3_c_tests_mfma1.log gpurun_out/r3_c_tests_mfma2.log; cat gpurun_out/r3_c_variants.txt
