#!/bin/bash
# The wave-specialised / decoupled resample kernels against the one-role kernel: their tests, then tools/ab_switches.py on the
# moderate-ratio shapes (settings interleaved on one box, medians of three).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ab_ws; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_resample_ws.py -x -q -m gpu > $OUT/ws_tests.log 2>&1
echo "ws tests rc=$?" >> $OUT/ws_tests.log
timeout 900 python tools/ab_switches.py --reps 3 --launches 30 --workloads cfg3-l0,cfg3-l1,cfg3-l2,cfg3-l3,cfg4-resize,cfg1-resize \
  --settings base:ws=0 spec:ws=1 dec:ws=2 dec_r2:ws=2,ws_ring=2 dec_r1:ws=2,ws_ring=1 \
  > $OUT/ab.jsonl 2> $OUT/ab.err
tail -4 $OUT/ws_tests.log; grep median $OUT/ab.jsonl; tail -3 $OUT/ab.err
