#!/bin/bash
# PMC passes for any command: tools/profile_pmc_cmd.sh <out-tag> <kernel-substring> <command...>
# (each counter group in its own run with --kernel-trace only, as the guide prescribes)
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
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
         "SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- "$@" > /dev/null 2> $OUT/p$i.err
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$KSUB" >> $OUT/summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k:
        continue
    key = (k.split("(")[0][-40:], r["Counter_Name"])
    agg[key][0] += float(r["Counter_Value"])
    agg[key][1] += 1
for (k, name), (tot, n) in sorted(agg.items()):
    print(f"{k:42s} {name:26s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
  else
    echo "pass $i ($C): no csv" >> $OUT/summary.txt; tail -2 $OUT/p$i.err >> $OUT/summary.txt
  fi
  rm -rf $OUT/p$i
done
cat $OUT/summary.txt
