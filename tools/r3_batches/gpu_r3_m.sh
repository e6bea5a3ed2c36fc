#!/bin/bash
# round-3 GPU batch M: full GPU suite + the shapes the narrow-shape tuning touches
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3_m_tests.log
: > gpurun_out/r3_m_bench.txt
for wl in cfg5 cfg2 cfg2-alpha; do
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --pattern random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:30], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> gpurun_out/r3_m_bench.txt
done
cat gpurun_out/r3_m_tests.log gpurun_out/r3_m_bench.txt
