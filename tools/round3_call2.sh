#!/bin/bash
# second-to-last GPU call of round 3: the banded two-pass kernel -- its own tests, the resample / shim suites with it switched
# on in auto mode, and the up-scale workloads with and without it
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3y; mkdir -p $O
date +%s > $O/t0
el() { echo "$1 rc=$2 t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log; }
timeout 120 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -k "banded" > $O/banded_tests.log 2>&1; el banded_tests $?; tail -15 $O/banded_tests.log
RT="tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_abi_shim.py tests/test_gpu_bitmap_ops.py"
IFHIP_BANDED=2 timeout 150 python -m pytest $RT -m gpu -q -p no:cacheprovider > $O/suite_banded2.log 2>&1; el suite_banded2 $?; tail -6 $O/suite_banded2.log
WL="up3-robidoux up2-hermite"
timeout 60 python tools/ab_variants.py generic_or_fused $WL >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_base $?
IFHIP_BANDED=2 timeout 60 python tools/ab_variants.py banded $WL >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_banded $?
IFHIP_BANDED=2 IFHIP_BANDED_FLAGS=0 timeout 60 python tools/ab_variants.py banded_taploop $WL >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_banded_taploop $?
IFHIP_BANDED=2 IFHIP_TRACE_LAUNCH=1 timeout 60 python bench.py --workload up3-robidoux --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "banded launch" | head -2 | tee $O/trace_launch.txt
cat $O/ab.jsonl
