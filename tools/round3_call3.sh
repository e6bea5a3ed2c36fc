#!/bin/bash
# round 3, GPU call 3: banded kernel, second form (frame loop, tables once per workgroup, prefetch), default in auto mode
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3x; mkdir -p $O
date +%s > $O/t0
el() { echo "$1 rc=$2 t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log; }
timeout 120 python -m pytest tests/test_gpu_resample.py -m gpu -q -p no:cacheprovider -k "banded" > $O/banded_tests.log 2>&1; el banded_tests $?; tail -12 $O/banded_tests.log
timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite_default.log 2>&1; el suite_default $?; tail -5 $O/suite_default.log
RT="tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_abi_shim.py tests/test_gpu_bitmap_ops.py"
IFHIP_BANDED=2 timeout 150 python -m pytest $RT -m gpu -q -p no:cacheprovider > $O/suite_banded2.log 2>&1; el suite_banded2 $?; tail -4 $O/suite_banded2.log
IFHIP_BANDED=2 IFHIP_BANDED_FLAGS=0 timeout 150 python -m pytest $RT -m gpu -q -p no:cacheprovider > $O/suite_banded2_flags0.log 2>&1; el suite_banded2_flags0 $?; tail -4 $O/suite_banded2_flags0.log
WL="up3-robidoux up2-hermite"
IFHIP_BANDED=0 timeout 60 python tools/ab_variants.py generic_or_fused $WL >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_base $?
IFHIP_BANDED=2 timeout 60 python tools/ab_variants.py banded $WL >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_banded $?
for F in 6 5 3; do IFHIP_BANDED=2 IFHIP_BANDED_FLAGS=$F timeout 60 python tools/ab_variants.py banded_flags$F up3-robidoux >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_flags$F $?; done
for W in 512 1024 4096 8192; do IFHIP_BANDED=2 IFHIP_BANDED_WGS=$W timeout 60 python tools/ab_variants.py banded_wgs$W up3-robidoux >> $O/ab.jsonl 2>> $O/ab_err.log; el ab_wgs$W $?; done
IFHIP_BANDED=2 IFHIP_TRACE_LAUNCH=1 timeout 60 python bench.py --workload up3-robidoux --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "banded launch" | head -1 | tee $O/trace_launch.txt
cat $O/ab.jsonl
