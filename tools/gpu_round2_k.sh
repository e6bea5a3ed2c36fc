#!/bin/bash
# whole GPU suite + entropy kernel trace
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
timeout 120 bash tools/trace_entropy.sh 16 2>&1 | tail -9
