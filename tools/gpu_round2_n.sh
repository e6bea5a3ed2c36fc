#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "jpeg or scaler or pipelines or abi" 2>&1 | tail -5
bash tools/gpu_round2_m.sh 2>&1 | head -12
grep -A12 scale_4_8 gpurun_out/r2m/bench_jpeg.json | head -14
