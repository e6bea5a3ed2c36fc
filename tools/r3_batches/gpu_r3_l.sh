#!/bin/bash
# round-3 GPU batch L: the horizontal wave -- parity (whole resample suite) and A/B against IFHIP_NO_H_WAVE=1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_process_group.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_l_tests.log
: > gpurun_out/r3_l_variants.txt
for wl in cfg2 cfg2-alpha cfg5; do
  python tools/exp_variants.py --reps 3 --workload $wl IFHIP_NO_H_WAVE=1 >> gpurun_out/r3_l_variants.txt 2>&1
done
python tools/exp_variants.py --reps 2 --workload cfg2 --pattern mixed IFHIP_NO_H_WAVE=1 >> gpurun_out/r3_l_variants.txt 2>&1
cat gpurun_out/r3_l_tests.log gpurun_out/r3_l_variants.txt
