#!/bin/bash
# round-3 GPU batch I: cfg5 -- strip access vs arithmetic (one strip as its own frame), no table gathers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/exp_variants.py --reps 2 --workload cfg5 _noconv _noconv2 _loadonly _nochain > gpurun_out/r3_i_variants.txt 2>&1
python tools/exp_variants.py --reps 2 --workload cfg5-quarter _noconv _loadonly >> gpurun_out/r3_i_variants.txt 2>&1
cat gpurun_out/r3_i_variants.txt
