#!/bin/bash
# A/B of pre-built kernel variants on ONE box:  tools/ab_run.sh "<case>:<tile>:<variant> ..."   (variants = build/ab/libgcdm_<variant>.so)
# restores the default library at the end.  Output: gpurun_out/ab_run.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
cp bio-diffusion_amd/libgcdm_hip.so /tmp/libgcdm_keep.so
: > gpurun_out/ab_run.log
for spec in $1; do
    IFS=: read -r case tile v <<< "$spec"
    cp build/ab/libgcdm_$v.so bio-diffusion_amd/libgcdm_hip.so
    GCDM_EDGE_TILE=$tile timeout 170 python tools/ab_variant.py $v $case base 2>&1 | grep "^AB" | tee -a gpurun_out/ab_run.log
done
cp /tmp/libgcdm_keep.so bio-diffusion_amd/libgcdm_hip.so
