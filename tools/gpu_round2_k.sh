#!/bin/bash
# JPEG GPU tests + entropy kernel trace
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q -k "jpeg or entropy or pipelines or abi" 2>&1 | tail -6
timeout 120 bash tools/trace_entropy.sh 16 2>&1 | tail -9
