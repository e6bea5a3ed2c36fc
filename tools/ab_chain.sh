#!/bin/bash
# A/B of two library builds on one box: IFHIP_LIB picks the library (imageflow_amd/_native.py).
# usage: tools/ab_chain.sh [command ...]   (default: the cfg4 chain at 128 frames); the previous build is lib/libimageflow_hip_prev.so
cd "$(dirname "$0")/.."
P=$PWD/imageflow_amd/lib/libimageflow_hip_prev.so
if [ $# -eq 0 ]; then set -- python tools/bench_jpeg.py 128 --chain 40; fi
for i in 1 2 3; do
  echo "new  $("$@" 2>/dev/null | tail -1 | cut -c1-300)"
  echo "prev $(IFHIP_LIB=$P "$@" 2>/dev/null | tail -1 | cut -c1-300)"
done
