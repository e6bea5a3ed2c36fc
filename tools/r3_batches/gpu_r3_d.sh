#!/bin/bash
# round-3 GPU batch D: instruction issue-rate probe, the shim tests after the draw_image_exact fix, up-scale shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tools/probes/issue_rate_probe > gpurun_out/r3_d_issue_rate.txt 2>&1
python -m pytest tests/test_gpu_abi_shim.py tests/test_gpu_resample.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3_d_tests.log
for wl in up2-hermite up3-robidoux; do
  python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3_d_bench_$wl.json 2> gpurun_out/r3_d_bench_$wl.err
done
cat gpurun_out/r3_d_issue_rate.txt; tail -3 gpurun_out/r3_d_tests.log; cat gpurun_out/r3_d_bench_up*.json | cut -c1-1500; tail -3 gpurun_out/r3_d_bench_up*.err
