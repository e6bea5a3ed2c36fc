#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 120 bash tools/trace_entropy.sh 16 2>&1 | tail -3
head -8 gpurun_out/trace_entropy/kernel_stats.csv | cut -d, -f1-4 | cut -c1-100
