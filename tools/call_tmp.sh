set -u
export TMPDIR=/tmp
O=gpurun_out/r4_s3; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_resample.py tests/test_gpu_random_shapes.py tests/test_gpu_pipelines.py tests/test_gpu_reference_checksums.py tests/test_gpu_jpeg.py tests/test_gpu_abi_shim.py -m gpu -q 2>&1 | tail -15 > $O/tests.log
cat $O/tests.log
for rep in 1 2; do
AB_REPS=1 IFHIP_NO_TWO_COL=1 timeout 200 python tools/ab_variants.py four cfg3-l1 cfg3-l3 cfg3 >> $O/ab.jsonl 2>> $O/ab_err.log
AB_REPS=1 timeout 200 python tools/ab_variants.py two cfg3-l1 cfg3-l3 cfg3 >> $O/ab.jsonl 2>> $O/ab_err.log
done
cat $O/ab.jsonl; tail -3 $O/ab_err.log
timeout 400 tools/profile_pmc.sh cfg3-l1 > $O/pmc_cfg3-l1_two.txt 2>&1
grep -E "LDS|VALU|GRBM" $O/pmc_cfg3-l1_two.txt
