cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/exp_entropy_variants.py gen
python tools/exp_entropy_variants.py run
for v in lut10 lut11 inner1 sub512 sub2048; do IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so python tools/exp_entropy_variants.py run 2>&1 | tail -1; done
for v in inner1 lut11; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2e/$v -- env IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so python tools/exp_entropy_variants.py run > /dev/null 2>&1
f=$(find gpurun_out/r2e/$v -name '*kernel_stats.csv' | head -1); echo $v; head -4 "$f" | cut -c1-120
done
