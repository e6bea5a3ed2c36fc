#!/bin/bash
# round-3 GPU batch N: decode + resample as one call (planar source in the fused resampler): parity + timing
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_resample.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3_n_tests.log
python tools/bench_jpeg.py 32 > gpurun_out/r3_n_bench_jpeg.json 2> gpurun_out/r3_n_bench_jpeg.err
tail -6 gpurun_out/r3_n_tests.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_n_bench_jpeg.json'))
for k,v in d.items():
    if k.startswith('scale'): print(k, {a:b for a,b in v.items() if 'ms' in a or 'fused' in a or 'one_call' in a})
PY
tail -3 gpurun_out/r3_n_bench_jpeg.err
