set -u
export TMPDIR=/tmp
O=gpurun_out/r4_s2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_jpeg.py -m gpu -q -x 2>&1 | tail -3 > $O/tests.log
cat $O/tests.log
for rep in 1 2; do
AB_REPS=1 IFHIP_LIB=$PWD/imageflow_amd/lib/libimageflow_hip_permoff.so timeout 200 python tools/ab_variants.py permoff cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize cfg3 >> $O/ab.jsonl 2>> $O/ab_err.log
AB_REPS=1 timeout 200 python tools/ab_variants.py perm cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize cfg3 >> $O/ab.jsonl 2>> $O/ab_err.log
done
cat $O/ab.jsonl
timeout 400 tools/profile_pmc.sh cfg3-l0 > $O/pmc_cfg3-l0_perm.txt 2>&1
grep -E "LDS|VALU|GRBM" $O/pmc_cfg3-l0_perm.txt
