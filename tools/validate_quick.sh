#!/bin/bash
# quick end-of-session check on a GPU box: whole GPU suite, entropy chain bench + per-kernel statistics
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 bash tools/trace_entropy.sh 16 2>&1 | tail -2
head -8 gpurun_out/trace_entropy/kernel_stats.csv | cut -d, -f1-4 | cut -c1-100
