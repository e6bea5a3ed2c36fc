#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 60 python tools/exp_entropy_variants.py gen
for v in ww16 ww32 ww96 fl8 fl24; do
  IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so timeout 60 python tools/exp_entropy_variants.py run 2>&1 | tail -1
done
timeout 60 python tools/exp_entropy_variants.py run 2>&1 | tail -1
timeout 60 python tools/exp_entropy_variants.py run 2>&1 | tail -1
