#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
timeout 300 python bench.py > gpurun_out/r2p/bench_a.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r2p/bench_a.json').read().strip().splitlines()[-1]);print('A', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['measured_read_GBps'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
timeout 300 python bench.py --steps 100 --warmup 20 > gpurun_out/r2p/bench_b.json 2>/dev/null; python -c "
import json;d=json.loads(open('gpurun_out/r2p/bench_b.json').read().strip().splitlines()[-1]);print('B', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
