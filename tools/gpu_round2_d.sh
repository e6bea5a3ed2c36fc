cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
python tools/bench_entropy.py 16 > gpurun_out/r2d/bench_entropy_base.json 2>&1; cat gpurun_out/r2d/bench_entropy_base.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d/trace -- python tools/bench_entropy.py 16 > /dev/null 2>&1
f=$(find gpurun_out/r2d/trace -name '*kernel_stats.csv' | head -1); head -12 "$f" | cut -c1-200
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2d/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'entropy' in r['Kernel_Name']]
# print last decode's kernel sequence
for r in rows[-8:]:
    print(r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us', 'gap to prev start', '')
PY
