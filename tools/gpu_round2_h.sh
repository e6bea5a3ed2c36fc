cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_jpeg_entropy.py tests/test_gpu_jpeg_random.py tests/test_gpu_abi_shim.py -x -q 2>&1 | tail -3
python tools/exp_entropy_variants.py gen
python tools/exp_entropy_variants.py run
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h/trace -- python tools/exp_entropy_variants.py run > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2h/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'entropy' in r['Kernel_Name']]
for r in rows[-5:]:
    print(r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
