#!/bin/bash
# stream priorities for the two slices of the timed loop, with and without the fused layer launch on the slice handles (ONE box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
OUT=gpurun_out/ab_prio.log
: > $OUT
F="--steps 100 --warmup 5 --no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs"
for rep in 1 2; do
 for cfg in "0 none" "0 -1,0" "1 -1,0" "1 none" "1 -1,-1" "0 none"; do
  set -- $cfg
  if [ "$2" = "none" ]; then unset GCDM_LANE_PRIO; else export GCDM_LANE_PRIO=$2; fi
  GCDM_LANE_FUSE=$1 timeout 200 python bench.py --workload ${WL:-qm9} --lanes 2 $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BENCH ${WL:-qm9} lanes=2 lane_fuse=$1 prio=$2 ms_per_step=%.4f median=%.4f sclk=%s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline'].get('sclk_mhz')))" | tee -a $OUT
 done
done
