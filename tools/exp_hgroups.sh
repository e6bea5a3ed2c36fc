#!/bin/bash
cd "$(dirname "$0")/.."
for g in "" 0 1 2 3 4 6; do
  IFHIP_HGROUPS=$g timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --pattern random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hgroups \"$g\"', d['value'], 'ms', d['roofline']['kernel_ms'], 'GB/s', d['roofline']['achieved'])"
done
