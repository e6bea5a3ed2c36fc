#!/bin/bash
# Host code of the JPEG stages under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU): the file writer on random
# coefficient planes with every option, and the header parser + un-stuffing + decode-table builders + host walkers
# (ifhip_jpeg_debug_scan_report) on mutated copies of committed files.  Builds into /tmp; prints one summary line each.
set -eu
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
W=/tmp/ifhip_sanitize; rm -rf $W; mkdir -p $W; cd $W
INC="-I$ROOT/imageflow_amd/csrc -I$ROOT/include"
SAN="-O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17"
g++ $SAN $INC -o writer_fuzz "$ROOT/tools/sanitize/writer_fuzz.cpp" "$ROOT/imageflow_amd/csrc/jpeg_write.cpp" "$ROOT/tools/sanitize/stubs.cpp"
./writer_fuzz
# the entropy file's host half: host-only compile of the .hip source; the device blob it would embed is an empty stand-in
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only $SAN $INC -c "$ROOT/imageflow_amd/csrc/jpeg_entropy.hip" -o entropy_host.o 2>/dev/null
SYM=$(nm entropy_host.o | awk '/__hip_fatbin_/ {print $2; exit}')
printf '__attribute__((section(".hip_fatbin"))) const char %s[64] = {0};\n' "$SYM" > fatbin_stub.c
gcc -c fatbin_stub.c -o fatbin_stub.o
g++ $SAN $INC -c "$ROOT/tools/sanitize/parser_fuzz.cpp" -o parser_fuzz.o
g++ $SAN -c "$ROOT/tools/sanitize/stubs.cpp" -o stubs.o
# (the handles' memory comes from the library's cache: devmem.cpp, host-only like the rest; debug_switch lives in weights.cpp)
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only $SAN $INC -c "$ROOT/imageflow_amd/csrc/devmem.cpp" -o devmem_host.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address,undefined -o parser_fuzz parser_fuzz.o entropy_host.o devmem_host.o stubs.o fatbin_stub.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
python3 - "$ROOT" <<'PY'
import sys, numpy as np
z = np.load(sys.argv[1] + "/tests/golden/jpeg_entropy_cases.npz")
for i in (0, 5, 17, 33, 48, 60):
    open(f"/tmp/ifhip_sanitize/case_{i}.jpg", "wb").write(z[f"jpg_{i}"].tobytes())
PY
ASAN_OPTIONS=detect_leaks=0 ./parser_fuzz case_*.jpg
# the whole library host-only (every translation unit, device blobs replaced by empty stand-ins) behind the libimageflow
# ABI subset: damaged JSON jobs
mkdir -p lib && : > fatbin_stubs.c
OBJS=""
for src in "$ROOT"/imageflow_amd/csrc/*.cpp "$ROOT"/imageflow_amd/csrc/*.hip; do
  base=$(basename "$src")
  if [ "$base" = "resample_fused.hip" ]; then
    for k in 1 2 3 4 5 6 7 8; do
      /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only $SAN $INC -DIFHIP_FUSED_K=$k -c "$src" -o lib/fused_$k.o 2>/dev/null &
      OBJS="$OBJS lib/fused_$k.o"
    done
  elif [ "$base" = "resample_ws.hip" ]; then
    for k in 1 2 3 4 5; do
      /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only $SAN $INC -DIFHIP_FUSED_K=$k -c "$src" -o lib/ws_$k.o 2>/dev/null &
      OBJS="$OBJS lib/ws_$k.o"
    done
  else
    /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 --cuda-host-only $SAN $INC -c "$src" -o lib/$base.o 2>/dev/null &
    OBJS="$OBJS lib/$base.o"
  fi
done
wait
for o in $OBJS; do
  for sym in $(nm $o | awk '/ U __hip_fatbin_/ {print $2}'); do printf '__attribute__((section(".hip_fatbin"))) const char %s[64] = {0};\n' "$sym" >> fatbin_stubs.c; done
done
sort -u fatbin_stubs.c -o fatbin_stubs.c
gcc -c fatbin_stubs.c -o fatbin_stubs.o
g++ $SAN $INC -c "$ROOT/tools/sanitize/json_fuzz.cpp" -o json_fuzz.o
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address,undefined -o json_fuzz json_fuzz.o $OBJS fatbin_stubs.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lpthread
ASAN_OPTIONS=detect_leaks=0 ./json_fuzz
