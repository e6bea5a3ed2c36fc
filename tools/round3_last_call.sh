#!/bin/bash
# The round's last GPU call (a few minutes of box time left): whole GPU suite on the product build, then an A/B of the
# step-order / premultiply variant builds (tools/ab_variants.py; lib/libimageflow_hip_v{A,B,C,D}.so built beforehand with
# imageflow_amd.build.build_variant), their resample suites, the driver's bench line and the device coder's bench.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
L=$PWD/imageflow_amd/lib
WL="cfg2 cfg2-alpha cfg5 cfg3"
date +%s > $O/t0
timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite_base.log 2>&1; echo "suite_base rc=$?" | tee -a $O/steps.log; tail -3 $O/suite_base.log
ab() { for v in "$@"; do
    if [ "$v" = base ]; then timeout 60 python tools/ab_variants.py base $WL >> $O/ab.jsonl 2>> $O/ab_err.log
    else IFHIP_LIB=$L/libimageflow_hip_$v.so timeout 60 python tools/ab_variants.py $v $WL >> $O/ab.jsonl 2>> $O/ab_err.log; fi
    echo "ab $v rc=$? t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log; done; }
ab base vD vB vA vC
RT="tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py"
for v in vD vA vC; do
  IFHIP_LIB=$L/libimageflow_hip_$v.so timeout 90 python -m pytest $RT -m gpu -q -p no:cacheprovider > $O/suite_$v.log 2>&1
  echo "suite_$v rc=$? t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log; tail -2 $O/suite_$v.log
done
timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_like.json 2> $O/bench_err.log; echo "bench rc=$? t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log
timeout 60 python tools/bench_jpeg_encode.py > $O/bench_jpeg_encode.json 2>> $O/bench_err.log; echo "jpeg_encode rc=$? t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log
ab vD base vB
cat $O/ab.jsonl | tail -45
