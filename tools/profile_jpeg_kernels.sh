#!/bin/bash
# kernel statistics of the JPEG pixel stage bench (8/8, 4/8 spatial sRGB, 1/8) -- per-kernel averages
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/jpeg_kernels
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python tools/bench_jpeg.py 32 > $OUT/bench_jpeg.json 2> $OUT/err.txt
t=$(find $OUT/t -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && timeout 60 python - "$t" > $OUT/kernels.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group by (kernel name, grid) and report count / avg
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
for k, (n, tot) in agg.items():
    if "jpeg" in k[0] or "fused" in k[0] or "scale" in k[0]:
        print(f"{k[0]:60s} grid {k[1]:>9s} {k[2]:>5s} {k[3]:>3s}  calls {n:4d}  avg {tot / n:9.1f} us")
PY
rm -rf $OUT/t
cat $OUT/kernels.txt; head -40 $OUT/bench_jpeg.json
