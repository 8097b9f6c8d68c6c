// Times the real tile_gemm_x3 (from the library header) in isolation: ticks per call per wave.
#include "../bio-diffusion_amd/csrc/gcdm_edge_x3.hip.h"
#include <cstdio>
#include <vector>

template <int VARIANT, int PD>
__global__ __launch_bounds__(512) void kb(float* out, const h8* wH, const h8* wL, int KB, int reps, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 36 * 65 * 8; i += 512) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 am[1][2], al[1][2];
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) { am[0][n][r] = 0.f; al[0][n][r] = 0.f; }
    X3Ring<1, PD> ring;
    const h8* xh8 = (const h8*)smem;
    const h8* xl8 = (const h8*)(smem + 36 * 65 * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        x3_prefetch<1, PD>(ring, wH + (size_t)wave * KB * 64, wL + (size_t)wave * KB * 64, KB, lane);
        tile_gemm_x3<1, 2, PD>(am, al, ring, wH + (size_t)wave * KB * 64, wL + (size_t)wave * KB * 64, KB, xh8, xl8, 65, lane);
        if (VARIANT == 1) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += am[0][n][r] + al[0][n][r];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VARIANT, int PD>
void run(const char* name, int KB, int reps, int blocks) {
    float* out; h8 *wH, *wL; unsigned long long* ticks;
    hipMalloc(&out, 4 * 512 * blocks); hipMalloc(&wH, 8 * KB * 64 * 16 + 65536); hipMalloc(&wL, 8 * KB * 64 * 16 + 65536); hipMalloc(&ticks, 8 * 8 * blocks);
    hipMemset(wH, 0, 8 * KB * 64 * 16); hipMemset(wL, 0, 8 * KB * 64 * 16);
    hipFuncSetAttribute((const void*)kb<VARIANT, PD>, hipFuncAttributeMaxDynamicSharedMemorySize, 152108);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kb<VARIANT, PD>), dim3(blocks), dim3(512), 152108, 0, out, wH, wL, KB, 2, ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kb<VARIANT, PD>), dim3(blocks), dim3(512), 152108, 0, out, wH, wL, KB, reps, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8 * blocks);
    hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double mfma_per_simd = 2.0 * 6 * KB * reps;
    printf("%-28s KB=%2d blocks=%3d: %.0f ticks/call/wave = %.1f cycles per MFMA per SIMD ; wall %.3f ms -> clock %.2f GHz\n", name, KB, blocks,
           mean / reps, mean / mfma_per_simd, ms, mean / (ms * 1e6));
}

int main() {
    run<0, 2>("PD=2", 18, 400, 256);
    run<0, 3>("PD=3", 18, 400, 256);
    run<0, 2>("PD=2 KB=72", 72, 100, 256);
    return 0;
}
