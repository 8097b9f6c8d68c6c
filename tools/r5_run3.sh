# per-wave phase stamps of stamped variants: tools/r5_run3.sh "<variant> ..."  -> gpurun_out/r5_stamps_<variant>.txt
cp bio-diffusion_amd/libgcdm_hip.so /tmp/libgcdm_keep.so
for v in $1; do
    cp build/ab/libgcdm_$v.so bio-diffusion_amd/libgcdm_hip.so
    timeout 120 python tests/gpu_time.py qm9 1024 > gpurun_out/r5_stamps_$v.txt 2>&1
    grep -E "^ +(1|2|3|4|19|20) " gpurun_out/r5_stamps_$v.txt | head -8
done
cp /tmp/libgcdm_keep.so bio-diffusion_amd/libgcdm_hip.so
