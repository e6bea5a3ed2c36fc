set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench_cfg2.json 2> gpurun_out/r2a/bench_cfg2.err; tail -1 gpurun_out/r2a/bench_cfg2.json
python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_cfg3.json 2>&1; tail -1 gpurun_out/r2a/bench_cfg3.json
python bench.py --scaling strong --total-frames 1024 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_strong1.json 2>&1; tail -1 gpurun_out/r2a/bench_strong1.json
for w in cfg2-alpha cfg5 cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_$w.json 2>&1; tail -1 gpurun_out/r2a/bench_$w.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline'].get('measured_read_GBps'))"
done
