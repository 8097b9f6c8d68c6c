#!/bin/bash
# A/B of the fused layer launch (node tiles as a tail role of the persistent edge workgroups) on ONE box: GCDM_FUSE_NODE=0|1, GCDM_FUSE_TILE=32|64
#   tools/ab_fuse.sh "qm9 geom"      -> gpurun_out/ab_fuse.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
: > gpurun_out/ab_fuse.log
for case in ${1:-qm9 geom}; do
  for rep in 1 2; do
    GCDM_FUSE_NODE=0 timeout 170 python tools/ab_variant.py unfused $case unfused 2>&1 | grep "^AB" | tee -a gpurun_out/ab_fuse.log
    GCDM_FUSE_NODE=1 GCDM_FUSE_TILE=32 timeout 170 python tools/ab_variant.py fused32 $case unfused 2>&1 | grep -E "^AB|rror" | tee -a gpurun_out/ab_fuse.log

  done
done
