// Mimics the x3 GEMM loop: per block 6 f16 MFMAs (+ optional software-pipelined 4x ds_read_b128, 2x global_load_dwordx4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS, bool GLB, bool VACC>
__global__ void k(float* out, const h8* __restrict__ w, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    h8 ah[3], al[3], bh[2][2], bl[2][2];
    for (int r = 0; r < 3; ++r) for (int i = 0; i < 8; ++i) { ah[r][i] = (_Float16)(lane * 0.001f + i + r); al[r][i] = (_Float16)(i * 0.01f + r); }
    for (int r = 0; r < 2; ++r) for (int n = 0; n < 2; ++n) for (int i = 0; i < 8; ++i) { bh[r][n][i] = (_Float16)(0.5f + i); bl[r][n][i] = (_Float16)(0.25f * i); }
    const h8* sh = (const h8*)smem + (lane >> 5) * 65 + (lane & 31);
    const h8* wl = w + lane;
    __syncthreads();
    for (int it = 0; it < iters; it += 6) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int kb = (it + r) & 15;
            if (GLB) { ah[(r + 2) % 3] = wl[kb * 64]; al[(r + 2) % 3] = wl[kb * 64 + 2048]; }
            if (LDS) {
#pragma unroll
                for (int n = 0; n < 2; ++n) { bh[(r + 1) & 1][n] = sh[(2 * kb) * 65 + n * 32]; bl[(r + 1) & 1][n] = sh[(2 * kb) * 65 + n * 32 + 36 * 65]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r % 3], bh[r & 1][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r % 3], bh[r & 1][1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r % 3], bl[r & 1][0], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r % 3], bl[r & 1][1], a3, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[r % 3], bh[r & 1][0], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[r % 3], bh[r & 1][1], a3, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool LDS, bool GLB>
void run(const char* name, int threads, int blocks, int iters, const h8* w) {
    float* out;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipFuncSetAttribute((const void*)k<LDS, GLB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDS, GLB, true>), dim3(blocks), dim3(threads), 80000, 0, out, w, 60);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDS, GLB, true>), dim3(blocks), dim3(threads), 80000, 0, out, w, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 6.0 * iters * (threads / 256.0);
    printf("%-40s threads=%d blocks=%d : wall %.3f ms -> %.2f ns per MFMA per SIMD (32 cyc @1.9GHz = 16.8 ns)\n", name, threads, blocks, ms, ms * 1e6 / mfma_per_simd);
    hipFree(out);
}

int main() {
    h8* w; hipMalloc(&w, 16 * 1024 * 1024); hipMemset(w, 0, 16 * 1024 * 1024);
    const int it = 60000;
    run<false, false>("MFMA only", 512, 256, it, w);
    run<true, false>("MFMA + pipelined 4x ds_read_b128", 512, 256, it, w);
    run<false, true>("MFMA + 2x global_load_dwordx4 (L1/L2)", 512, 256, it, w);
    run<true, true>("MFMA + both", 512, 256, it, w);
    run<true, true>("MFMA + both, 1 wave/SIMD", 256, 256, it, w);
    return 0;
}
