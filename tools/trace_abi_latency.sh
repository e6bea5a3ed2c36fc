#!/bin/bash
# Where a JOB's wall time goes between the host and the device while jobs run through the libimageflow ABI: rocprofv3 with the
# HIP API trace, the kernel trace and the memory-copy trace of tools/bench_abi_jobs.cpp, joined by correlation id:
#   dispatch latency   = kernel start - start of the launch call that submitted it
#   completion latency = end of a synchronising call - end of the last device activity its thread had submitted before it
#   queue delay        = kernel start - max(launch call, end of the same thread's previous kernel)
# plus the time each thread spent inside each HIP entry point, and how many hardware queues the kernels ran on.
#   usage (on the GPU box): [IFHIP_<SWITCH>=v ...] tools/trace_abi_latency.sh <job kind> <threads> [seconds]
#   -> gpurun_out/trace_abi/latency_<kind>_<threads>.txt
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
KIND=${1:-cfg4}; T=${2:-64}; SEC=${3:-0.6}
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
timeout 90 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --output-format csv -d $W/t -- \
    $W/bench_abi_jobs $PWD/imageflow_amd/lib/libimageflow_hip.so $W/in.jpg $W/job.json $T $SEC > $W/run.json 2> $W/err.txt
{ echo "# $KIND, $T threads, $SEC s under rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace"
  grep -o '"jobs_per_s": [0-9.]*' $W/run.json | head -1
python - "$W/t" <<'PY'
import bisect, collections, csv, glob, os, sys
root = sys.argv[1]
def find(suffix):
    f = [p for p in glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True)]
    return f[0] if f else None
api_f, ker_f, cpy_f = find("hip_api_trace.csv"), find("kernel_trace.csv"), find("memory_copy_trace.csv")
if not api_f or not ker_f:
    print("no traces:", api_f, ker_f); sys.exit(0)
api = []                                                # (thread, start, end, name, corr)
for r in csv.DictReader(open(api_f)):
    api.append((int(r["Thread_Id"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], int(r["Correlation_Id"])))
by_corr = {a[4]: a for a in api}
kern = []                                               # (thread, start, end, name, queue, launch_start)
queues = collections.Counter()
for r in csv.DictReader(open(ker_f)):
    a = by_corr.get(int(r["Correlation_Id"]))
    if not a: continue
    kern.append((a[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?"), a[1]))
    queues[r.get("Queue_Id", "?")] += 1
copies = []
if cpy_f:
    for r in csv.DictReader(open(cpy_f)):
        a = by_corr.get(int(r["Correlation_Id"]))
        if a: copies.append((a[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy"), "-", a[1]))
def pct(v, p):
    v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))] / 1e3 if v else 0.0
def line(title, v):
    print(f"{title:58s} n {len(v):7d}  p50 {pct(v, .5):9.1f} us  p90 {pct(v, .9):9.1f}  p99 {pct(v, .99):9.1f}  mean {sum(v) / max(1, len(v)) / 1e3:9.1f}")
span = max(a[2] for a in api) - min(a[1] for a in api)
print(f"span {span / 1e6:.1f} ms, {len(api)} HIP calls, {len(kern)} kernels on {len(queues)} hardware queues {dict(queues)}, {len(copies)} copies")
line("dispatch latency (kernel start - launch call start)", [k[1] - k[5] for k in kern])
line("copy latency (copy start - call start)", [c[1] - c[5] for c in copies])
per_thread = collections.defaultdict(list)
for k in kern + copies: per_thread[k[0]].append(k)
qd = []
for t, ks in per_thread.items():
    ks.sort(key=lambda k: k[5])
    prev_end = 0
    for k in ks:
        qd.append(k[1] - max(k[5], prev_end)); prev_end = max(prev_end, k[2])
line("queue delay (start - max(launch, thread's previous end))", qd)
# completion latency of synchronising calls
sync_names = ("hipStreamSynchronize", "hipEventSynchronize", "hipDeviceSynchronize", "hipMemcpy")
comp, waits = [], []
for t, ks in per_thread.items():
    ends = sorted(k[2] for k in ks)
    for a in api:
        if a[0] != t or a[3] not in sync_names: continue
        i = bisect.bisect_right(ends, a[2]) - 1
        if i >= 0 and ends[i] >= a[1]:                  # something of this thread finished while it waited
            comp.append(a[2] - ends[i]); waits.append(a[2] - a[1])
line("synchronising calls that waited: their duration", waits)
line("  completion latency (call end - thread's last device end)", comp)
tot = collections.defaultdict(lambda: [0, 0])
for a in api:
    tot[a[3]][0] += 1; tot[a[3]][1] += a[2] - a[1]
nthreads = len({a[0] for a in api})
print(f"-- time inside HIP entry points, summed over {nthreads} threads (span x threads = {span * nthreads / 1e6:.0f} ms)")
for name, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:34s} {n:7d} calls {ns / 1e6:10.1f} ms  mean {ns / n / 1e3:9.1f} us")
PY
} > $OUT/latency_${KIND}_$T.txt 2>&1
cat $OUT/latency_${KIND}_$T.txt
tail -3 $W/err.txt
rm -rf $W
