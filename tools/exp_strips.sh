#!/bin/bash
# Experiment: narrower strips / capped LDS per workgroup (co-resident workgroups) on the moderate-ratio shapes.
cd "$(dirname "$0")/.."
run() {  # name, env...
  local name=$1; shift
  local out
  out=$(env "$@" timeout 120 python bench.py --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1)
  echo "$WL $name $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step", d["ms_per_step"], "kernel_GBps", d["roofline"]["achieved"], "frac", round(d["roofline"]["frac"],3))' 2>/dev/null || echo "$out" | tail -c 300)"
}
for WL in cfg3-l0 cfg5 cfg2-alpha cfg2; do
  export WL
  run default A=1
  run lanes512 IFHIP_MAX_LANES=512
  run lanes512_lds80 IFHIP_MAX_LANES=512 IFHIP_LDS_LIMIT=81920
  run lanes256 IFHIP_MAX_LANES=256
  run lanes256_lds80 IFHIP_MAX_LANES=256 IFHIP_LDS_LIMIT=81920
  run lanes256_lds53 IFHIP_MAX_LANES=256 IFHIP_LDS_LIMIT=54272
  run lanes256_lds40 IFHIP_MAX_LANES=256 IFHIP_LDS_LIMIT=40960
  run lanes128_lds40 IFHIP_MAX_LANES=128 IFHIP_LDS_LIMIT=40960
done
