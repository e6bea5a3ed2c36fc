#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ab_ws; rm -rf $OUT; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_resample_ws.py -x -q -m gpu > $OUT/ws_tests.log 2>&1
echo "ws tests rc=$?" >> $OUT/ws_tests.log
timeout 900 python tools/ab_switches.py --reps 3 --launches 30 --workloads cfg3-l0,cfg3-l1,cfg3-l2,cfg4-resize \
  --settings base:ws=0 ws0:ws=1 ws_r2:ws=1,ws_ring=2 \
     ws0h6:ws=1,ws_h_waves=6 \
  > $OUT/ab.jsonl 2> $OUT/ab.err
tail -3 $OUT/ws_tests.log; grep median $OUT/ab.jsonl; tail -3 $OUT/ab.err
