#!/bin/bash
# entropy stage: tests, kernel trace, write-pass variants -- every step under its own timeout
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
timeout 120 python -m pytest tests/test_gpu_jpeg_entropy.py -x -q 2>&1 | tail -3
timeout 120 bash tools/trace_entropy.sh 16 2>&1 | tail -9
timeout 60 python tools/exp_entropy_variants.py gen
for v in "$@"; do
timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2i/$v -- env IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so python tools/exp_entropy_variants.py run > /dev/null 2>&1
f=$(find gpurun_out/r2i/$v -name '*kernel_stats.csv' | head -1); echo $v; head -4 "$f" | cut -d, -f1-4 | cut -c1-120; rm -rf gpurun_out/r2i/$v
done
