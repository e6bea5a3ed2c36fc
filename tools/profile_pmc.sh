#!/bin/bash
# PMC passes for one bench workload (each counter group in its own run, kernel-trace only): tools/profile_pmc.sh <workload>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
WL=${1:-cfg3-l0}
OUT=gpurun_out/pmc_$WL
mkdir -p $OUT
CMD="python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline"
: > $OUT/summary.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
         "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "fused_resample" not in r.get("Kernel_Name", ""):
        continue
    agg[r["Counter_Name"]][0] += float(r["Counter_Value"])
    agg[r["Counter_Name"]][1] += 1
for name, (tot, n) in sorted(agg.items()):
    print(f"{name:28s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
  else
    echo "pass $i ($C): no csv" >> $OUT/summary.txt; tail -2 $OUT/p$i.err >> $OUT/summary.txt
  fi
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
