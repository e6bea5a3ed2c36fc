#!/bin/bash
# Where the GPU's time goes while jobs run through the libimageflow ABI: rocprofv3 --kernel-trace of tools/bench_abi_jobs.cpp
# (T threads, one job kind), then per kernel: launches, summed duration, and how much of the wall span any kernel was running.
#   usage (on the GPU box): tools/trace_abi_jobs.sh <job kind> <threads> [seconds]      -> gpurun_out/trace_abi/<kind>_<threads>.txt
# Keep the run SHORT (0.5 s): at this build's 9 000 jobs/s a 1.0 s run under --kernel-trace died inside librocprofiler-sdk
# (SIGSEGV below an HSA hook, or a hang; the harness prints the frames) three times out of three, 0.5 s runs four out of four
# fine, with any wait mode and slot count -- profiles/NOTEBOOK.md, round 5.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
KIND=${1:-cfg1}; T=${2:-32}; SEC=${3:-0.5}
OUT=gpurun_out/trace_abi; mkdir -p $OUT
W=$(mktemp -d)
python - "$W" "$KIND" <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_abi_jobs as B
w, kind = sys.argv[1], sys.argv[2]
B.build_harness(w)
open(os.path.join(w, "in.jpg"), "wb").write(B.make_file())
open(os.path.join(w, "job.json"), "w").write(json.dumps(B.JOBS[kind]))
PY
timeout 60 rocprofv3 --kernel-trace --output-format csv -d $W/t -- $W/bench_abi_jobs $PWD/imageflow_amd/lib/libimageflow_hip.so $W/in.jpg $W/job.json $T $SEC > $W/run.json 2> $W/err.txt
grep "crash frame" $W/err.txt | head -40
f=$(find $W/t -name '*kernel_trace.csv' | head -1)
{ echo "# $KIND, $T threads, $SEC s under rocprofv3 --kernel-trace"; grep -o '"jobs_per_s": [0-9.]*' $W/run.json | head -1
python - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in rows:
    a = agg[k.split("(")[0][-60:]]; a[0] += 1; a[1] += e - s
print(f"launches {len(rows)}  span {span/1e6:.1f} ms  some kernel running {busy/1e6:.1f} ms = {busy/span:.3f}  summed kernel time {sum(v[1] for v in agg.values())/1e6:.1f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:62s} {n:7d} launches {t/1e6:9.2f} ms  avg {t/n/1e3:8.1f} us")
PY
} > $OUT/${KIND}_$T.txt
cat $OUT/${KIND}_$T.txt
rm -rf $W
