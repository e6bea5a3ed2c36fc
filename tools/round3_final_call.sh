#!/bin/bash
# round 3, last GPU call: whole GPU suite + smoke on the final build, then rocprofv3 kernel statistics of the bench line,
# of the 3x up-scale (banded kernel) and of the device entropy coder
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r3w; rm -rf $O; mkdir -p $O
date +%s > $O/t0
el() { echo "$1 rc=$2 t=$(( $(date +%s) - $(cat $O/t0) ))" | tee -a $O/steps.log; }
timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite.log 2>&1; el suite $?; tail -4 $O/suite.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; el smoke $?; tail -1 $O/smoke.log
stats() {   # stats <tag> <grep pattern> -- cmd...
  local tag=$1 pat=$2; shift 3
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- "$@" > $O/bench_${tag}_under_trace.json 2> $O/trace_$tag.err
  local f=$(find $O/trace_$tag -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then head -1 $f > $O/${tag}_kernel_stats.csv; grep -E "$pat" $f >> $O/${tag}_kernel_stats.csv; fi
  rm -rf $O/trace_$tag
  el stats_$tag $?
}
stats cfg2 "fused_resample|read_probe" -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strong-field
stats up3-robidoux "banded|generic|fused_resample" -- python bench.py --workload up3-robidoux --steps 20 --warmup 3 --no-cpu-baseline
stats jpeg_encode "jpeg|enc_|forward|scan|stuff|count" -- python tools/bench_jpeg_encode.py 32
timeout 60 python bench.py --workload up3-robidoux --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_up3-robidoux.json 2>/dev/null; el bench_up3 $?
timeout 60 python bench.py --workload up2-hermite --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_up2-hermite.json 2>/dev/null; el bench_up2 $?
head -5 $O/*_kernel_stats.csv
