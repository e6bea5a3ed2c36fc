#!/bin/bash
# experiment driver: bench each lib variant (IFHIP_LIB) -- NOT part of the product
cd "$(dirname "$0")/.."
for v in "" _d2 _d6 _d8 _noh _loadonly _loadonly8 "$@"; do
  lib=imageflow_amd/lib/libimageflow_hip$v.so
  [ -f $lib ] || continue
  IFHIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pattern random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant \"$v\"', d['value'], 'ms', d['roofline']['kernel_ms'], 'GB/s', d['roofline']['achieved'])"
done
python - <<'PY'
import ctypes as C, torch
from imageflow_amd import _native
L=_native.lib(); torch.cuda.init(); torch.zeros(1,device='cuda')
bw=C.c_double()
for mb in (256, 1024, 4096):
    rc=L.ifhip_measure_copy_bandwidth(mb<<20, 10, C.byref(bw)); print("copy", mb, "MiB", rc, round(bw.value/1e9,1), "GB/s (r+w)")
PY
