#!/bin/bash
# ~20 s: the IFHIP_ENCODE_STATIC variant build (lib/libimageflow_hip_vE.so) -- every GPU test file that reaches the resample
# kernels, then the worst moderate-ratio shape on both builds
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
L=$PWD/imageflow_amd/lib/libimageflow_hip_vE.so
IFHIP_LIB=$L timeout 20 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_abi_shim.py tests/test_gpu_bitmap_ops.py tests/test_gpu_jpeg.py -m gpu -q -p no:cacheprovider > $O/suite_vE.log 2>&1
echo "suite_vE rc=$?" | tee $O/steps.log; tail -2 $O/suite_vE.log
AB_REPS=1 IFHIP_LIB=$L timeout 12 python tools/ab_variants.py vE cfg3-l1 >> $O/ab.jsonl 2>> $O/ab_err.log; echo "ab vE rc=$?" | tee -a $O/steps.log
AB_REPS=1 timeout 12 python tools/ab_variants.py base cfg3-l1 >> $O/ab.jsonl 2>> $O/ab_err.log; echo "ab base rc=$?" | tee -a $O/steps.log
cat $O/ab.jsonl
