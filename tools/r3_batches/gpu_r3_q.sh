#!/bin/bash
# round-3 GPU batch Q: row hand-over through counters + rotating, deferred horizontal pass: parity and A/B (IFHIP_NO_H_DEFER=1)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 150 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r3_q_tests.log
: > gpurun_out/r3_q_variants.txt
for wl in cfg2 cfg2-alpha cfg5; do
  timeout 200 python tools/exp_variants.py --reps 3 --workload $wl IFHIP_NO_H_DEFER=1 >> gpurun_out/r3_q_variants.txt 2>&1
done
timeout 100 python tools/exp_variants.py --reps 2 --workload cfg2 --pattern mixed IFHIP_NO_H_DEFER=1 >> gpurun_out/r3_q_variants.txt 2>&1
cat gpurun_out/r3_q_tests.log gpurun_out/r3_q_variants.txt
