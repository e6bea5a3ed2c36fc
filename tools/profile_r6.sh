#!/bin/bash
# Round-6 evidence run (on the GPU box through gpurun): every figure DESIGN.md section 6 quotes for this round, from one build on one box.
# Raw output under gpurun_out/prof6, the summaries kept under gpurun_out/prof6_summary (copied to profiles/r6_* afterwards).  Counter passes are separate runs with --kernel-trace only.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof6
SUM=gpurun_out/prof6_summary
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
# SECTIONS="5" tools/profile_r6.sh re-measures one part only (default: all)
SECTIONS=${SECTIONS:-"0 1 2 3 4 5 7 8"}     # (6: jobs through the libimageflow ABI -- not re-measured in round 6)
want() { case " $SECTIONS " in *" $1 "*) return 0;; esac; return 1; }

pmc_pass() {   # pmc_pass <tag> <kernel substring> <counters> -- cmd...
  local tag=$1 ksub=$2 ctr=$3; shift 4
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$tag -- "$@" > /dev/null 2> $OUT/pmc_$tag.err
  local f=$(find $OUT/pmc_$tag -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$ksub" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k:
        continue
    key = (k.split("(")[0][-52:], r["Counter_Name"])
    agg[key][0] += float(r["Counter_Value"]); agg[key][1] += 1
for (k, name), (tot, n) in sorted(agg.items()):
    print(f"{k:54s} {name:24s} per-dispatch avg {tot / max(n, 1):.6g}  (dispatches {n})")
PY
  else echo "($ctr): no counter csv: $(tail -1 $OUT/pmc_$tag.err)"; fi
  rm -rf $OUT/pmc_$tag
}

# 0. the whole GPU suite and smoke() on this build
if want 0; then
{ timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; } > $SUM/gpu_tests_and_smoke.log
fi

# 1. headline: bench line, the same command under kernel-trace, HBM traffic + SQ counters
if want 1; then
python bench.py > $SUM/bench_cfg2.json 2> $SUM/bench_cfg2.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg2 -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-strong-field --no-other-configs > $SUM/bench_cfg2_under_trace.json 2> $OUT/trace_cfg2.err
find $OUT/trace_cfg2 -name '*kernel_stats.csv' -exec cp {} $SUM/cfg2_kernel_stats.csv \;
{ echo "# rocprofv3 --pmc passes (one counter group per run), python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong-field --no-other-configs, kernel fused_resample"
  for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
    pmc_pass cfg2 fused_resample "$C" -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-strong-field --no-other-configs
  done; } > $SUM/cfg2_pmc.txt
fi

# 2. cfg5 (BASELINE's HBM-roofline config) and cfg2 with alpha: bench line + kernel stats + counters
if want 2; then
for W in cfg5 cfg2-alpha; do
  python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > $SUM/bench_$W.json 2> /dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> $OUT/trace_$W.err
  find $OUT/trace_$W -name '*kernel_stats.csv' -exec sh -c "head -1 {} > $SUM/${W}_kernel_stats.csv; grep fused_resample {} >> $SUM/${W}_kernel_stats.csv" \;
  { echo "# rocprofv3 --pmc passes, python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline, kernel fused_resample"
    for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
      pmc_pass $W fused_resample "$C" -- python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline
    done; } > $SUM/${W}_pmc.txt
  rm -rf $OUT/trace_$W
done
fi

# 3. the other resample shapes (this round: output columns dealt to the lane groups of ds_read_b128, two-column groups): bench line + kernel stats
if want 3; then
for W in cfg3-l0 cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize up2-hermite up3-robidoux; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -- python bench.py --workload $W --steps 60 --warmup 5 --no-cpu-baseline > $SUM/bench_$W.json 2> $OUT/trace_$W.err
  find $OUT/trace_$W -name '*kernel_stats.csv' -exec sh -c "head -1 {} > $SUM/${W}_kernel_stats.csv; grep -E 'fused_resample|generic' {} >> $SUM/${W}_kernel_stats.csv" \;
  rm -rf $OUT/trace_$W
done
# counters of the two moderate-ratio shapes this round's changes aim at (LDS busy / bank conflicts / VALU)
for W in cfg3-l0 cfg3-l1; do
  { echo "# rocprofv3 --pmc passes, python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline, kernel fused_resample"
    for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
      pmc_pass $W fused_resample "$C" -- python bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline
    done; } > $SUM/${W}_pmc.txt
done
fi

# 4. jobs: export_4_sizes, 1024-frame strong-scaling job on one GPU
if want 4; then
python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline > $SUM/bench_cfg3_job.json 2>/dev/null
python bench.py --workload cfg3 --outputs bgra --steps 50 --warmup 5 --no-cpu-baseline > $SUM/bench_cfg3_job_bgra.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3job -- python bench.py --workload cfg3 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/trace_cfg3job.err
find $OUT/trace_cfg3job -name '*kernel_stats.csv' -exec sh -c "head -14 {} | cut -c1-220 > $SUM/cfg3_job_kernel_stats.csv" \;
rm -rf $OUT/trace_cfg3job
python bench.py --scaling strong --total-frames 1024 --steps 30 --warmup 5 --no-cpu-baseline > $SUM/bench_strong_1024_1gpu.json 2>/dev/null
fi

# 5. JPEG: pixel stage (cfg4 chain as one call / two calls), per-kernel statistics, HBM traffic of the chain; entropy chain
if want 5; then
python tools/bench_jpeg.py 32 > $SUM/bench_jpeg.json 2> /dev/null
tools/profile_jpeg_kernels.sh > /dev/null 2>&1; cp gpurun_out/jpeg_kernels/kernels.txt $SUM/bench_jpeg_kernels.txt
{ echo "# cfg4 chain (32 frames 3840x2160 4:2:0 -> 4/8 decode, spatial sRGB luma -> 800x450), HBM traffic per chain call, summed over its kernels"
  echo "# rocprofv3 --pmc <counter> --kernel-trace -- python tools/bench_jpeg.py 32 --chain 6 [--two-call]; FETCH_SIZE / WRITE_SIZE in KB per dispatch"
  for MODE in "" "--two-call"; do
    echo "## one call (planes -> resampler)${MODE:+ -- NO: two calls (BGRA bitmap in HBM)}"
    for C in "FETCH_SIZE" "WRITE_SIZE"; do
      pmc_pass chain "" "$C" -- python tools/bench_jpeg.py 32 --chain 6 $MODE
    done
  done; } > $SUM/cfg4_chain_traffic.txt
python tools/bench_entropy.py 16 > $SUM/bench_entropy.json 2> /dev/null
timeout 300 python tools/bench_entropy.py 16 --streams 1,2,4 > $SUM/bench_entropy_streams.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_jpeg -- python tools/bench_entropy.py 16 > /dev/null 2> $OUT/trace_jpeg.err
find $OUT/trace_jpeg -name '*kernel_stats.csv' -exec sh -c "head -9 {} > $SUM/jpeg_chain_kernel_stats.csv" \;
{ echo "# rocprofv3 --pmc passes on the entropy stage, python tools/bench_entropy.py 1 (one file)"
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    pmc_pass ent "entropy_" "$C" -- python tools/bench_entropy.py 1
  done; } > $SUM/entropy_pmc.txt
fi

# 6. jobs through the libimageflow ABI (file in, file out): T threads x one context per job, cfg1 / cfg4 / cfg4h jobs
if want 6; then
timeout 900 python tools/bench_abi_jobs.py --threads 1,8,16,64,128 --seconds 2.5 --spread > $SUM/abi_jobs.json 2> $SUM/abi_jobs.err
fi

# 7. BASELINE config 4 as a bench.py workload (files -> entropy decode -> 4/8 pixel stage -> 800x450), line + kernel statistics
if want 7; then
python bench.py --workload cfg4 --steps 20 --warmup 5 > $SUM/bench_cfg4.json 2> $SUM/bench_cfg4.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $SUM/bench_cfg4_under_trace.json 2> $OUT/trace_cfg4.err
find $OUT/trace_cfg4 -name '*kernel_stats.csv' -exec sh -c "head -14 {} | cut -c1-220 > $SUM/cfg4_kernel_stats.csv" \;
rm -rf $OUT/trace_cfg4
fi
# 8. round 6: entropy stage of cfg4 -- rounds per batch, one batch alone under kernel-trace, SQ counters on a two-file batch; gather-overlap emulation
if want 8; then
{ for F in 0 64; do timeout 300 python tools/exp_entropy_rounds.py 64 24,0 $F 1; done; } > $SUM/entropy_rounds.txt 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ent1 -- python tools/exp_entropy_rounds.py 64 0 0 1 > /dev/null 2> $OUT/trace_ent1.err
find $OUT/trace_ent1 -name '*kernel_stats.csv' -exec sh -c "head -5 {} > $SUM/entropy_one_batch_kernel_stats.csv" \;
timeout 600 python tools/exp_gather_overlap.py > $SUM/gather_overlap_emulation.jsonl 2> /dev/null
fi
rm -rf $OUT
ls -la $SUM
