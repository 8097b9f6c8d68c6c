#!/bin/bash
# The round's measurement set in ONE gpurun call (one box): rocprofv3 stats + PMC passes for QM9 and GEOM, the PMC summaries written
# into profiles/ of the box's copy (so that the bench lines that follow carry `traffic` of THIS build), then the three bench lines.
#   GCDM_GIT_HEAD=<short sha> tools/gpu_round.sh r02        -> gpurun_out/<tag>_*  (copy what is to be kept into profiles/)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
# gate: the configuration that exposed the stale-operand fault of k_edge_embed_x3, repeated, before anything is measured
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -k "edge_embedding_of_both_modes or ragged_geom_is_bit_reproducible" > $OUT/${TAG}_gate.log 2>&1 || { tail -5 $OUT/${TAG}_gate.log; echo "GATE FAILED"; exit 1; }
tools/gpu_profile.sh ${TAG} qm9 > $OUT/${TAG}_profile_qm9.log 2>&1
tools/gpu_profile.sh ${TAG}g geom > $OUT/${TAG}_profile_geom.log 2>&1
# one pass of the exact-fp32 MFMA kernel family (kernel stats + the SQ counters)
GCDM_MFMA=f32 GCDM_PROFILE_SHORT=1 tools/gpu_profile.sh ${TAG}f32 qm9 > $OUT/${TAG}_profile_qm9_f32.log 2>&1
python profiles/pmc_summarize.py $OUT/${TAG}f32_pmc1 > $OUT/${TAG}_pmc_summary_qm9_f32.json
python profiles/pmc_summarize.py $OUT/${TAG}_pmc1 $OUT/${TAG}_pmc2 $OUT/${TAG}_pmc3 $OUT/${TAG}_pmc4 > $OUT/${TAG}_pmc_summary_qm9_x3.json
python profiles/pmc_summarize.py $OUT/${TAG}g_pmc1 $OUT/${TAG}g_pmc2 $OUT/${TAG}g_pmc3 $OUT/${TAG}g_pmc4 > $OUT/${TAG}_pmc_summary_geom_x3.json
cp $OUT/${TAG}_pmc_summary_qm9_x3.json $OUT/${TAG}_pmc_summary_geom_x3.json profiles/
for d in ${TAG} ${TAG}g ${TAG}f32; do f=$(ls -t $OUT/${d}_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/${d}_kernel_stats.csv; done
# the bench lines below recompute roofline.frac from the committed kernel statistics: those of THIS box and build (profiles/<tag>_bench_{qm9,geom}_x3_kernel_stats.csv)
[ -f $OUT/${TAG}_kernel_stats.csv ] && cp $OUT/${TAG}_kernel_stats.csv profiles/${TAG}_bench_qm9_x3_kernel_stats.csv
[ -f $OUT/${TAG}g_kernel_stats.csv ] && cp $OUT/${TAG}g_kernel_stats.csv profiles/${TAG}_bench_geom_x3_kernel_stats.csv
[ -f $OUT/${TAG}f32_kernel_stats.csv ] && cp $OUT/${TAG}f32_kernel_stats.csv profiles/${TAG}_bench_qm9_f32_kernel_stats.csv
# raw counter dumps are large: keep the summaries and the per-kernel stats only
rm -rf $OUT/${TAG}_pmc[1-4] $OUT/${TAG}g_pmc[1-4] $OUT/${TAG}f32_pmc1 $OUT/${TAG}_stats $OUT/${TAG}g_stats $OUT/${TAG}f32_stats
timeout 400 python bench.py --steps 100 --warmup 5 > $OUT/${TAG}_bench_qm9.json 2> $OUT/${TAG}_bench_qm9.err
timeout 250 python bench.py --workload geom --steps 100 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_geom.json 2> $OUT/${TAG}_bench_geom.err
timeout 200 python bench.py --lanes 1 --steps 100 --warmup 5 --no-cpu-baseline --no-other-configs > $OUT/${TAG}_bench_qm9_lanes1.json 2> $OUT/${TAG}_bench_lanes1.err
# in-kernel phase stamps need the -DGCDM_STAMPS build (tools/build_variants.sh stamps:-DGCDM_STAMPS -> build/ab/libgcdm_stamps.so)
if [ -f build/ab/libgcdm_stamps.so ]; then
    cp bio-diffusion_amd/libgcdm_hip.so /tmp/libgcdm_keep.so; cp build/ab/libgcdm_stamps.so bio-diffusion_amd/libgcdm_hip.so
    GCDM_FUSE_NODE=0 timeout 100 python tests/gpu_time.py qm9 1024 > $OUT/${TAG}_phase_stamps_qm9.txt 2>&1
    GCDM_FUSE_NODE=0 timeout 100 python tests/gpu_time.py geom 256 > $OUT/${TAG}_phase_stamps_geom.txt 2>&1
    timeout 100 python tests/gpu_node_phases.py qm9 512 > $OUT/${TAG}_node_phase_stamps_qm9.txt 2>&1
    GCDM_FUSE_NODE=0 timeout 100 python tests/gpu_first_tile.py qm9 > $OUT/${TAG}_first_tile_qm9.txt 2>&1
    cp /tmp/libgcdm_keep.so bio-diffusion_amd/libgcdm_hip.so
fi
tail -c 600 $OUT/${TAG}_bench_qm9.json; echo; tail -3 $OUT/${TAG}_profile_qm9.log
