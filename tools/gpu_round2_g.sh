cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python tools/exp_entropy_variants.py gen
python tools/exp_entropy_variants.py run
for v in warm512 warm1024 warm2048; do IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so python tools/exp_entropy_variants.py run 2>&1 | tail -1; done
for v in warm1024; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2g/$v -- env IFHIP_LIB=$GRAFT_REPO_ROOT/imageflow_amd/lib/libimageflow_hip_$v.so python tools/exp_entropy_variants.py run > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2g/warm1024/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'entropy' in r['Kernel_Name']]
for r in rows[-5:]:
    print(r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
done
