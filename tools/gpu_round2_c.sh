run() { tag=$1; shift; w=$1; shift; env "$@" python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', '$tag', j['roofline']['kernel_ms'], j['roofline']['frac'])"; }
for w in cfg3-l1 cfg3-l2 cfg3-l3 cfg4-resize cfg1-resize; do
  run base $w A=1
  run oneframe $w IFHIP_ONE_FRAME_PER_WG=1
  run oneframe_lds80 $w IFHIP_ONE_FRAME_PER_WG=1 IFHIP_LDS_LIMIT=81920
  run lds80 $w IFHIP_LDS_LIMIT=81920
  run oneframe_lds53 $w IFHIP_ONE_FRAME_PER_WG=1 IFHIP_LDS_LIMIT=54000
done
for w in cfg3-l0 cfg5 cfg2-alpha; do
  run base $w A=1
  run lanes512 $w IFHIP_MAX_LANES=512
  run lanes512_lds80 $w IFHIP_MAX_LANES=512 IFHIP_LDS_LIMIT=81920
  run lanes256_lds53 $w IFHIP_MAX_LANES=256 IFHIP_LDS_LIMIT=54000
  run bands2 $w IFHIP_BANDS=2
done
