#!/bin/bash
# bench-level A/B of the fused layer launch on ONE box (alternating): the headline loop (2 slices) and one handle, QM9 and GEOM
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
OUT=gpurun_out/ab_fuse_bench.log
: > $OUT
F="--steps 100 --warmup 5 --no-cpu-baseline --no-fp32-timing --no-extras --no-full-sample --no-other-configs"
for wl in ${1:-qm9 geom}; do
 for lanes in 2 1; do
  for rep in 1 2; do
   for v in "0 0" "1 32"; do
    set -- $v
    GCDM_FUSE_NODE=$1 GCDM_FUSE_TILE=$2 timeout 200 python bench.py --workload $wl --lanes $lanes $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BENCH $wl lanes=$lanes fuse=$1 tile=$2 ms_per_step=%.4f median=%.4f edge_ms=%.4f node_ms=%s tile_cycles=%s sclk=%s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['roofline']['node_kernel']['avg_launch_ms'], d['roofline']['tile_cycles'], d['roofline'].get('sclk_mhz')))" | tee -a $OUT
   done
  done
 done
done
