#!/bin/bash
set -u
cd /root/repo 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
W=$(mktemp -d)
python - "$W" <<'PY'
import json, os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench_abi_jobs as B
w = sys.argv[1]
B.build_harness(w)
open(os.path.join(w, "in.jpg"), "wb").write(B.make_file())
for k in ("cfg4", "cfg1", "cfg4h"):
    open(os.path.join(w, k + ".json"), "w").write(json.dumps(B.JOBS[k]))
PY
LIB=$PWD/imageflow_amd/lib/libimageflow_hip.so
run() { tag=$1; job=$2; thr=$3; reps=$4; shift 4
  for rep in $(seq 1 $reps); do
    r=$(env "$@" $W/bench_abi_jobs $LIB $W/in.jpg $W/$job.json $thr 1.5 3 2>/dev/null | grep -o '"jobs_per_s": [0-9.]*\|"host": {[^}]*}' | tr '\n' ' ')
    echo "$tag job=$job threads=$thr rep=$rep $r"
  done; }
mkdir -p gpurun_out/samples
samp() { tag=$1; job=$2; thr=$3; shift 3
  env "$@" IFHIP_BENCH_SAMPLE=$W/$tag.samples $W/bench_abi_jobs $LIB $W/in.jpg $W/$job.json $thr 2.0 3 2>/dev/null | grep -o '"jobs_per_s": [0-9.]*\|"host": {[^}]*}' | tr '\n' ' ' > gpurun_out/samples/$tag.txt
  echo >> gpurun_out/samples/$tag.txt
  python tools/sample_stacks.py $W/$tag.samples 16 >> gpurun_out/samples/$tag.txt 2>&1
  echo "=== $tag"; cat gpurun_out/samples/$tag.txt; }
for job in cfg1 cfg4h cfg4; do
  run base $job 64 2 A=1
  run spin8 $job 64 2 IFHIP_WAIT_SPINNERS=8
  run spin12 $job 64 1 IFHIP_WAIT_SPINNERS=12
  run sleep5 $job 64 1 IFHIP_WAIT_SLEEP_US=5
  run sleep50 $job 64 1 IFHIP_WAIT_SLEEP_US=50
  run spin8_sleep5 $job 64 1 IFHIP_WAIT_SPINNERS=8 IFHIP_WAIT_SLEEP_US=5
  run slots64_spin8 $job 64 1 IFHIP_WAIT_SPINNERS=8 IFHIP_JOB_SLOTS=64
  run slots96_t128 $job 128 1 IFHIP_JOB_SLOTS=96
  run inflight3 $job 64 1 IFHIP_COALESCE_DECODES_IN_FLIGHT=3
  run runtime $job 64 1 IFHIP_WAIT=runtime
done
rm -rf $W
