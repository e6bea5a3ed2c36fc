#!/bin/bash
# end-of-round validation: whole GPU suite, smoke, default bench line, entropy chain bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2o
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/r2o/bench_cfg2.json 2> gpurun_out/r2o/bench_err.txt; tail -c 1500 gpurun_out/r2o/bench_cfg2.json
timeout 200 python tools/bench_entropy.py 16 > gpurun_out/r2o/bench_entropy.json 2>/dev/null; head -16 gpurun_out/r2o/bench_entropy.json
