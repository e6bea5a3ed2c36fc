#!/bin/bash
# Three counter passes (issue / wait cycles, instruction mix, LDS) for any command, summed per kernel:
# tools/pmc_quick.sh <out-tag> <kernel-substring> <command...>   (each group in its own run with --kernel-trace only)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; KSUB=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
: > $OUT/summary.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$KSUB" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k:
        continue
    key = (k.split("(")[0][-32:], r["Counter_Name"])
    agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:34s} {c:24s} {v / n:16.0f} per launch ({n} launches)")
PY
  fi
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
