#!/bin/bash
# Builds kernel variants of libgcdm_hip.so for A/B runs (build container; hipcc cross-compiles gfx950 without a GPU):
#     tools/build_variants.sh base: stamps:-DGCDM_STAMPS "nogate:-DGCDM_STAMPS -DGCDM_ABLATIONS -DGCDM_ABL_NOGATE"
# -> build/ab/libgcdm_base.so, build/ab/libgcdm_v3.so   (build/ is git-ignored but travels to the GPU box with gpurun)
# then on the GPU box, same call, alternating:
#     for v in base v3 base v3; do cp build/ab/libgcdm_$v.so bio-diffusion_amd/libgcdm_hip.so; python tools/ab_variant.py $v qm9 base; done
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
for spec in "$@"; do
    name="${spec%%:*}"; flags="${spec#*:}"
    # (GCDM_BUILD_BASE=-fno-slp-vectorize replaces the packed-fp32 switch for the -DGCDM_X3_PK=1 variants: the assembler needs the feature for v_pk_*_f32)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC ${GCDM_BUILD_BASE:--Xclang -target-feature -Xclang -packed-fp32-ops} $flags \
        -o "build/ab/libgcdm_${name}.so" bio-diffusion_amd/csrc/gcdm_api.hip 2>&1 | grep -v "packed-fp32-ops" || true
    echo "built build/ab/libgcdm_${name}.so  [$flags]"
done
