#!/bin/bash
# round-3 GPU batch B: parity after the dot4 gather addresses, shim tests, MFMA exactness probe, A/B of the gather form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_abi_shim.py tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_process_group.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3_b_tests.log
tools/probes/mfma_fma_probe > gpurun_out/r3_b_mfma_probe.txt 2>&1
for wl in cfg5 cfg2-alpha cfg2 cfg3-l1; do
  python tools/exp_variants.py --reps 2 --workload $wl __nodot4 >> gpurun_out/r3_b_variants.txt 2>&1
done
tail -5 gpurun_out/r3_b_tests.log; cat gpurun_out/r3_b_mfma_probe.txt | tail -12; cat gpurun_out/r3_b_variants.txt
