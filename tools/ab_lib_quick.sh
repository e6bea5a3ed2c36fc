#!/bin/bash
# ~20 s of box time: a variant build of the library (lib/libimageflow_hip_<tag>.so, imageflow_amd.build.build_variant) on every
# GPU test file that reaches the resample kernels, then the given workloads on the variant and on the product build.
# usage: tools/ab_lib_quick.sh <tag> workload ...      (round 3: `vE cfg3-l1` for IFHIP_ENCODE_STATIC, profiles/r3_ab_encode_static.jsonl)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/ab_$TAG; mkdir -p $O
L=$PWD/imageflow_amd/lib/libimageflow_hip_$TAG.so
IFHIP_LIB=$L timeout 60 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_abi_shim.py tests/test_gpu_bitmap_ops.py tests/test_gpu_jpeg.py -m gpu -q -p no:cacheprovider > $O/suite_$TAG.log 2>&1
echo "suite_$TAG rc=$?" | tee $O/steps.log; tail -2 $O/suite_$TAG.log
AB_REPS=${AB_REPS:-1} IFHIP_LIB=$L timeout 60 python tools/ab_variants.py $TAG "$@" >> $O/ab.jsonl 2>> $O/ab_err.log; echo "ab $TAG rc=$?" | tee -a $O/steps.log
AB_REPS=${AB_REPS:-1} timeout 60 python tools/ab_variants.py base "$@" >> $O/ab.jsonl 2>> $O/ab_err.log; echo "ab base rc=$?" | tee -a $O/steps.log
cat $O/ab.jsonl
