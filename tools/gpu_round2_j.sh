#!/bin/bash
# entropy stage: per-workgroup trace of the round and write kernels (trace build prints from the device)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 60 python tools/exp_entropy_variants.py gen
IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_trace.so timeout 90 python tools/exp_entropy_variants.py run 2>&1 | grep "^WG\|^WR\|^wg" > gpurun_out/r2j/all.txt
grep "^WG" gpurun_out/r2j/all.txt | head -214 > gpurun_out/r2j/round_wg.txt
grep "^wg" gpurun_out/r2j/all.txt | head -600 > gpurun_out/r2j/round_its.txt
rm gpurun_out/r2j/all.txt
wc -l gpurun_out/r2j/*.txt
