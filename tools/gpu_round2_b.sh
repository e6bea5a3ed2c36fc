set -x
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py -x -q 2>&1 | tail -8
for w in cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize cfg2 cfg2-alpha cfg5; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2b/bench_$w.json 2>&1; tail -1 gpurun_out/r2b/bench_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['roofline']['kernel_ms'], j['roofline']['frac'])"
  IFHIP_NO_FAST_H=1 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w nofast', j['roofline']['kernel_ms'], j['roofline']['frac'])"
done
