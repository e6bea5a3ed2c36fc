#!/bin/bash
# round-3 GPU batch J: counters on cfg5 and cfg2-alpha with the current product library (split conversion)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/profile_pmc.sh cfg5 > /dev/null 2>&1
tools/profile_pmc.sh cfg2-alpha > /dev/null 2>&1
for wl in cfg5 cfg2-alpha; do
  CMD="python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline"
  for C in "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES"; do
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmcx -- $CMD > /dev/null 2> gpurun_out/pmcx.err
    f=$(find gpurun_out/pmcx -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then python - "$f" >> gpurun_out/pmc_$wl/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "fused_resample" not in r.get("Kernel_Name", ""): continue
    agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
for name, (tot, n) in sorted(agg.items()):
    print(f"{name:28s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
    else echo "($C): no csv: $(tail -1 gpurun_out/pmcx.err)" >> gpurun_out/pmc_$wl/summary.txt; fi
    rm -rf gpurun_out/pmcx
  done
done
cat gpurun_out/pmc_cfg5/summary.txt; echo; cat gpurun_out/pmc_cfg2-alpha/summary.txt
