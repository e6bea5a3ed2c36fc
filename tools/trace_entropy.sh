#!/bin/bash
# Per-launch durations of the entropy chain (rocprofv3 kernel trace of tools/bench_entropy.py); run through gpurun.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/trace_entropy
rm -rf $OUT; mkdir -p $OUT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python tools/bench_entropy.py ${1:-16} > $OUT/bench.json 2> $OUT/err.txt
f=$(find $OUT/t -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -9 "$f" | cut -c1-160 > $OUT/kernel_stats.csv
t=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && timeout 60 python - "$t" > $OUT/last_launches.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "entropy" in r["Kernel_Name"] or "jpeg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-16:]:
    print(f'{r["Kernel_Name"][:48]:48s} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0:9.1f} us')
PY
rm -rf $OUT/t
cat $OUT/kernel_stats.csv $OUT/last_launches.txt
grep entropy_decode_ms $OUT/bench.json
