#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
b() { timeout 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$1', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
b fresh
timeout 300 python -m pytest tests -m gpu -x -q -k "entropy" 2>&1 | tail -1
b after_entropy_tests
timeout 300 python -m pytest tests -m gpu -x -q -k "not entropy" 2>&1 | tail -1
b after_other_tests
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|power (W)\|junction" | head -4
sleep 10
b after_sleep
