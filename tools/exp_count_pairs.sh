#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_cpairs.so timeout 200 python -m pytest tests -m gpu -x -q -k "entropy" 2>&1 | tail -2
timeout 60 python tools/exp_entropy_variants.py gen
IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_cpairs.so timeout 60 python tools/exp_entropy_variants.py run 2>&1 | tail -1
timeout 60 python tools/exp_entropy_variants.py run 2>&1 | tail -1
