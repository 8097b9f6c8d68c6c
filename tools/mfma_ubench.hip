// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x2_f32 on gfx950, 1 or 2 waves per SIMD,
// with and without LDS operand reads; reports s_memtime ticks and wall time per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: f16 32x32x16 regs only, 1: f32 32x32x2 regs only, 2: f16 + LDS b128 reads per 6 MFMAs
__global__ void k(float* out, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(i * 0.01f); }
    float fx = lane * 0.001f, fy = 0.5f;
    const h8* lds = (const h8*)smem + lane;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            h8 b0 = lds[(it & 15) * 64], b1 = lds[(it & 15) * 64 + 1024], b2 = lds[(it & 7) * 64 + 2048], b3 = lds[(it & 7) * 64 + 3072];
            __builtin_amdgcn_sched_barrier(0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, b1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b2, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b3, a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, b1, a1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (MODE == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx, fy, a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fy, fx, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fy, fx, a1, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int blocks, int iters) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&ticks, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 70000, 0, out, ticks, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 70000, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), ticks, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double mfma_per_simd = 6.0 * iters * (threads / 256.0);
    printf("%-34s threads=%d blocks=%d : %.1f ticks/MFMA(per SIMD), wall %.3f ms -> %.1f ns/MFMA/SIMD => %.2f GHz-equivalent at 32cyc, %.2f at 64cyc\n",
           name, threads, blocks, mean / mfma_per_simd, ms, ms * 1e6 / mfma_per_simd, 32.0 / (ms * 1e6 / mfma_per_simd), 64.0 / (ms * 1e6 / mfma_per_simd));
    hipFree(out); hipFree(ticks);
}

int main() {
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 70000);
    const int it = 20000;
    run<0>("f16 32x32x16, 1 wave/SIMD, 1 CU-ish", 256, 1, it);
    run<0>("f16 32x32x16, 1 wave/SIMD, all CUs", 256, 256, it);
    run<0>("f16 32x32x16, 2 waves/SIMD, all CUs", 512, 256, it);
    run<2>("f16 + 4x ds_read_b128 / 6 MFMA, 2w", 512, 256, it);
    run<1>("f32 32x32x2, 1 wave/SIMD, all CUs", 256, 256, it);
    run<1>("f32 32x32x2, 2 waves/SIMD, all CUs", 512, 256, it);
    return 0;
}
