#!/bin/bash
# Round-2 evidence run (on the GPU box through gpurun).  Raw output under gpurun_out/prof2, the summaries kept under
# gpurun_out/prof2_summary (copied to profiles/r2_* afterwards).  Counter passes are separate runs with --kernel-trace only.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof2
SUM=gpurun_out/prof2_summary
rm -rf $OUT $SUM; mkdir -p $OUT $SUM

pmc_pass() {   # pmc_pass <tag> <kernel substring> <counters> -- cmd...
  local tag=$1 ksub=$2 ctr=$3; shift 4
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$tag -- "$@" > /dev/null 2> $OUT/pmc_$tag.err
  local f=$(find $OUT/pmc_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$ksub" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r.get("Kernel_Name", ""):
        continue
    agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
for name, (tot, n) in sorted(agg.items()):
    print(f"{name:28s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
  else echo "($ctr): no counter csv"; tail -2 $OUT/pmc_$tag.err; fi
  rm -rf $OUT/pmc_$tag
}

# 1. headline: bench line, the same command under kernel-trace, HBM traffic counters
python bench.py > $SUM/bench_cfg2.json 2> $SUM/bench_cfg2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg2 -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $SUM/bench_cfg2_under_trace.json 2> $OUT/trace_cfg2.err
find $OUT/trace_cfg2 -name '*kernel_stats.csv' -exec cp {} $SUM/cfg2_kernel_stats.csv \;
{ echo "# rocprofv3 --pmc passes (one counter group per run), python bench.py --steps 10 --warmup 2 --no-cpu-baseline, kernel fused_resample"
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    pmc_pass cfg2 fused_resample "$C" -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  done; } > $SUM/cfg2_pmc.txt

# 2. the other resample workloads: bench line + kernel stats + counters each
for W in cfg2-alpha cfg5 cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > $SUM/bench_$W.json 2> $OUT/trace_$W.err
  find $OUT/trace_$W -name '*kernel_stats.csv' -exec sh -c "head -1 {} > $SUM/${W}_kernel_stats.csv; grep fused_resample {} >> $SUM/${W}_kernel_stats.csv" \;
  { echo "# rocprofv3 --pmc passes, python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline, kernel fused_resample"
    for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
      pmc_pass $W fused_resample "$C" -- python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline
    done; } > $SUM/${W}_pmc.txt
  rm -rf $OUT/trace_$W
done

# 3. jobs: export_4_sizes, 1024-frame strong-scaling job on one GPU
python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline > $SUM/bench_cfg3_job.json 2>&1
python bench.py --scaling strong --total-frames 1024 --steps 30 --warmup 5 --no-cpu-baseline > $SUM/bench_strong_1024_1gpu.json 2>&1

# 4. JPEG: entropy + pixel stage chain (cfg4), per-kernel statistics; pixel stage alone with counters
python tools/bench_entropy.py 16 > $SUM/bench_entropy.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_jpeg -- python tools/bench_entropy.py 16 > /dev/null 2> $OUT/trace_jpeg.err
find $OUT/trace_jpeg -name '*kernel_stats.csv' -exec sh -c "head -8 {} > $SUM/jpeg_chain_kernel_stats.csv" \;
python tools/bench_jpeg.py 32 > $SUM/bench_jpeg.json 2> /dev/null
{ echo "# rocprofv3 --pmc passes, python tools/bench_jpeg.py 32, kernel jpeg_color_kernel<true> (fused luma IDCT + colour, full-size 4:2:0)"
  for C in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    pmc_pass jpeg "jpeg_color_kernel<true>" "$C" -- python tools/bench_jpeg.py 32
  done; } > $SUM/jpeg_color_pmc.txt
rm -rf $OUT
ls -la $SUM
