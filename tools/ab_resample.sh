#!/bin/bash
# A/B of two library builds on the resample workloads, interleaved on one box (kernel ms from bench.py's own hipEvents):
# usage: tools/ab_resample.sh [workload ...]   (default cfg2 cfg2-alpha cfg5); the previous build is lib/libimageflow_hip_prev.so
cd "$(dirname "$0")/.."
P=$PWD/imageflow_amd/lib/libimageflow_hip_prev.so
if [ $# -eq 0 ]; then set -- cfg2 cfg2-alpha cfg5; fi
ms() { python bench.py --workload "$1" --steps 60 --warmup 10 --no-cpu-baseline --no-strong-field --gather none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print(j['ms_per_step'], j.get('roofline',{}).get('kernel_ms'))"; }
for w in "$@"; do
  for i in 1 2; do
    echo "$w new  $(ms $w)"
    echo "$w prev $(IFHIP_LIB=$P ms $w)"
  done
done
