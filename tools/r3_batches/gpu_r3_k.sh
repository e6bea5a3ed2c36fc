#!/bin/bash
# round-3 GPU batch K: horizontal per-pixel loop unrolled (LDS latency of group q+1 under the chains of group q)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/exp_variants.py --reps 2 --workload cfg5 _hp2_6 _hp4_6 > gpurun_out/r3_k_variants.txt 2>&1
python tools/exp_variants.py --reps 2 --workload cfg2-alpha _hp2_4 _nochain4 >> gpurun_out/r3_k_variants.txt 2>&1
python tools/exp_variants.py --reps 2 --workload cfg2 _hp2_4 _nochain4 >> gpurun_out/r3_k_variants.txt 2>&1
cat gpurun_out/r3_k_variants.txt
