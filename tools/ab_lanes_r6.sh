#!/bin/bash
# lanes 1 / 2 / 3 / 4 of the headline loop, unfused and fused, on ONE box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
OUT=gpurun_out/ab_lanes_r6.log
: > $OUT
F="--steps 100 --warmup 5 --no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs"
for rep in 1 2; do
 for cfg in "2 0" "3 0" "4 0" "1 1" "2 0"; do
  set -- $cfg
  GCDM_FUSE_NODE=$2 GCDM_FUSE_TILE=32 timeout 200 python bench.py --lanes $1 $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BENCH qm9 lanes=$1 fuse=$2 ms_per_step=%.4f median=%.4f sclk=%s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline'].get('sclk_mhz')))" | tee -a $OUT
 done
done
