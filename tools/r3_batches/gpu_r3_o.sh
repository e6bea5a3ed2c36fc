#!/bin/bash
# round-3 GPU batch O: one lane per block IDCT -- parity on every JPEG test, A/B against the eight-lanes form
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_jpeg_scaled.py tests/test_gpu_jpeg_random.py tests/test_gpu_jpeg_entropy.py tests/test_gpu_pipelines.py tests/test_gpu_abi_shim.py tests/test_gpu_jpeg_forward.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r3_o_tests.log
python tools/bench_jpeg.py 32 > gpurun_out/r3_o_bench_jpeg.json 2> /dev/null
IFHIP_JPEG_IDCT8=1 python tools/bench_jpeg.py 32 > gpurun_out/r3_o_bench_jpeg_idct8.json 2> /dev/null
tail -5 gpurun_out/r3_o_tests.log; python - <<'PY'
import json
for f in ('gpurun_out/r3_o_bench_jpeg.json','gpurun_out/r3_o_bench_jpeg_idct8.json'):
    d=json.load(open(f)); print(f)
    for k,v in d.items():
        if k.startswith('scale'): print(' ',k, {a:b for a,b in v.items() if 'ms' in a})
PY
