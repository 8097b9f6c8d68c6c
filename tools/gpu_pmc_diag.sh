#!/bin/bash
# Issue-slot diagnosis of the edge kernel (run through gpurun): three SQ counter passes over a short bench.py run.
#   tools/gpu_pmc_diag.sh <tag> [workload]      -> gpurun_out/<tag>_sq{1,2,3} ; summary printed by profiles/pmc_summarize.py
set -u
TAG=${1:-diag}; WL=${2:-qm9}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --workload $WL --lanes 1 --steps 4 --warmup 2 --no-cpu-baseline --no-fp32-timing --no-other-configs"
run() { local name=$1; shift; (timeout 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_$name -- $B > $OUT/${TAG}_$name.log 2>&1; echo "$name exit=$?"); }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT SQ_INST_CYCLES_VMEM_RD
python $ROOT/profiles/pmc_summarize.py $OUT/${TAG}_sq1 $OUT/${TAG}_sq2 $OUT/${TAG}_sq3 > $OUT/${TAG}_sq_summary.json
python - <<PY
import json
d = json.load(open("$OUT/${TAG}_sq_summary.json"))
for k, v in d.items():
    if "edge_msg" in k or "k_node" in k:
        print(k, v.get("launches_sampled"))
        for c, x in sorted(v["mean_per_launch"].items()):
            print(f"    {c:32s} {x:16.0f}")
PY
