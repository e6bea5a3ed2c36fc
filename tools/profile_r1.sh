#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Writes raw output under gpurun_out/prof and the
# summaries we keep under gpurun_out/prof_summary (copied to profiles/ afterwards).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
SUM=gpurun_out/prof_summary
mkdir -p $OUT $SUM
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
TRACE_CMD="python bench.py --steps 50 --warmup 5 --no-cpu-baseline"   # long enough that the cold first launches do not move the average

# 1. kernel trace + stats (per-kernel average duration; must agree with bench.py's hipEvent number)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $TRACE_CMD > $SUM/bench_under_trace.json 2> $OUT/trace.err
find $OUT/trace -name '*kernel_stats.csv' -exec cp {} $SUM/kernel_stats.csv \;
find $OUT/trace -name '*kernel_trace.csv' | head -1 | xargs -I{} sh -c "head -1 {} > $SUM/kernel_trace_head.csv; grep fused_resample {} | head -20 >> $SUM/kernel_trace_head.csv"

# 2. PMC passes, each in its own run (HBM bytes; LDS; issue/wait breakdown)
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > /dev/null 2> $OUT/pmc$i.err
  f=$(find $OUT/pmc$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$C" >> $SUM/pmc_summary.txt <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if "fused_resample" not in k:
        continue
    agg[r["Counter_Name"]][0] += float(r["Counter_Value"])
    agg[r["Counter_Name"]][1] += 1
for name, (tot, n) in sorted(agg.items()):
    print(f"{name:28s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
  else
    echo "pass $i ($C): no counter csv; see $OUT/pmc$i.err" >> $SUM/pmc_summary.txt
    tail -3 $OUT/pmc$i.err >> $SUM/pmc_summary.txt
  fi
done
cat $SUM/kernel_stats.csv | head -8
cat $SUM/pmc_summary.txt
cat $SUM/bench_under_trace.json
