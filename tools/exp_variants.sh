#!/bin/bash
# experiment driver: bench each lib variant (IFHIP_LIB) 3x interleaved, report min/median kernel ms -- NOT part of the product
cd "$(dirname "$0")/.."
PATTERN=${PATTERN:-random}
VARIANTS=("" "$@")
declare -A RES
for rep in 1 2 3; do
  for v in "${VARIANTS[@]}"; do
    lib=imageflow_amd/lib/libimageflow_hip$v.so
    [ -f $lib ] || continue
    ms=$(IFHIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --pattern $PATTERN 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
    RES[$v]="${RES[$v]} $ms"
  done
done
for v in "${VARIANTS[@]}"; do echo "variant \"$v\" pattern $PATTERN kernel_ms:${RES[$v]}"; done
