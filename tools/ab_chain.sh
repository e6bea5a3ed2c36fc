#!/bin/bash
# A/B of two library builds on one box: IFHIP_LIB picks the library (imageflow_amd/_native.py)
cd "$(dirname "$0")/.."
P=$PWD/imageflow_amd/lib/libimageflow_hip_prev.so
for i in 1 2 3; do
  echo "new  $(python tools/bench_jpeg.py 128 --chain 40 2>/dev/null)"
  echo "prev $(IFHIP_LIB=$P python tools/bench_jpeg.py 128 --chain 40 2>/dev/null)"
done
