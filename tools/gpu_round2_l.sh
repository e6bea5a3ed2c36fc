#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 60 python tools/exp_entropy_variants.py gen
nproc
IFHIP_ENT_TIMING=1 timeout 120 python tools/exp_entropy_prepare.py 2>&1 | grep -v Warning | tail -12
