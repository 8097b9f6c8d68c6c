// Does padding a wave's MFMA stream with s_nop (the wave does not ask for the VALU port while its previous MFMA still occupies the matrix
// pipe) let the CO-RESIDENT wave's VALU work issue in the gaps?  (ubench6: with back-to-back MFMAs in wave A, wave B makes no progress.)
//   512 threads = 2 waves per SIMD: waves 0-3 run NM x { MFMA 32x32x16 f16 ; PAD wait states }, waves 4-7 run fp32 FMA chains (or nothing).
//   PAD is given in s_nop states (1 state = 1 cycle of the wave; the matrix pipe needs 32 cycles per MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PAD>
__device__ __forceinline__ void pad() {
    if constexpr (PAD >= 8) { asm volatile("s_nop 7"); pad<PAD - 8>(); }
    else if constexpr (PAD > 0) { asm volatile("s_nop %0" ::"n"(PAD - 1)); }
}

template <int PAD, int SLEEP>
__device__ __forceinline__ void do_mfma(int n, float* sink, int lane) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)(0.001f * lane); b[s] = (_Float16)(0.002f * s); }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (SLEEP) asm volatile("s_sleep %0" ::"n"(SLEEP)); else pad<PAD>();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    *sink = s;
}
__device__ __forceinline__ void do_valu(int n, float* sink, int lane) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    const float c = 1.0001f, d = 0.0003f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], c, d);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    *sink = s;
}

template <int PAD, int SLEEP>
__global__ __launch_bounds__(512) void kb(int with_valu, int n_mfma, int n_valu, float* out, unsigned long long* ticks) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0;
    if (wave < 4) do_mfma<PAD, SLEEP>(n_mfma, &r, lane);
    else if (with_valu) do_valu(n_valu, &r, lane);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = r;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAD, int SLEEP>
void run(const char* name) {
    const int blocks = 256, NM = 2000, NV = 2000;     // 8000 MFMAs per wave; 64000 FMAs per wave
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    for (int with = 0; with < 2; ++with) {
        hipLaunchKernelGGL((kb<PAD, SLEEP>), dim3(blocks), dim3(512), 0, 0, with, 10, 10, out, ticks);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL((kb<PAD, SLEEP>), dim3(blocks), dim3(512), 0, 0, with, NM, NV, out, ticks);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * blocks);
        (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w];
        a /= 4 * blocks; b /= 4 * blocks;
        printf("%-26s %-12s  MFMA wave %8.0f clk (%5.1f / MFMA)   FMA wave %8.0f clk (%5.2f / FMA)\n", name, with ? "| FMA wave" : "| idle", a, a / (4.0 * NM), b, b / (32.0 * NV));
    }
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<0, 0>("MFMA back to back");
    run<4, 0>("MFMA + 4 nop states");
    run<8, 0>("MFMA + 8 nop states");
    run<16, 0>("MFMA + 16 nop states");
    run<20, 0>("MFMA + 20 nop states");
    run<24, 0>("MFMA + 24 nop states");
    run<28, 0>("MFMA + 28 nop states");
    run<32, 0>("MFMA + 32 nop states");
    run<0, 1>("MFMA + s_sleep 1");
    return 0;
}
