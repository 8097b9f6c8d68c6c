#!/bin/bash
# Round profile of bench.py on the GPU box (run through gpurun): kernel-trace stats + separate PMC passes.
#   tools/gpu_profile.sh <tag> [workload]      -> gpurun_out/<tag>_{stats,pmc1..4}
set -u
TAG=${1:-prof}; WL=${2:-qm9}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
# --lanes 1: whole-batch launches, i.e. what bench.py's roofline object measures (the timed loop of the default run uses 2 slices)
# GCDM_MFMA=f32 in the environment profiles the exact-fp32 MFMA kernel family instead of the default split-precision one
# GCDM_FUSE_NODE=0: two launches per layer on the primary handle too (round 6: its default is the fused layer launch) -- the statistics are about the two kernels by
# themselves, as the roofline object is; the bench's own fused sections (roofline.fused_layer) still put the fused kernel's launches into the same trace
export GCDM_FUSE_NODE=${GCDM_FUSE_NODE:-0}
B="python $ROOT/bench.py --workload $WL --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-timing --no-other-configs --no-extras --no-full-sample"
run() { local name=$1; shift; (timeout 280 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/${TAG}_$name -- $B > $OUT/${TAG}_$name.log 2>&1; echo "$name exit=$?"); }
run stats --stats
run pmc1 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU
if [ "${GCDM_PROFILE_SHORT:-0}" != "1" ]; then
run pmc2 --pmc FETCH_SIZE
run pmc3 --pmc WRITE_SIZE
run pmc4 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
fi
du -sh $OUT/${TAG}_*
