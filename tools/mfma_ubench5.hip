// Same-wave interleave: does VALU work placed in the shadow of the wave's own MFMAs run for free when two such waves share a SIMD?
//   per iteration: 4 independent MFMAs (32x32x16 f16, 32 clk each) + NV independent fp32 FMAs (8 chains), scheduled MFMA,NV/4 x VALU,...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(512) void kb(int n, int active_waves, float* out, unsigned long long* ticks) {
    __shared__ float4 sm[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 512) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)(0.001f * lane); b[s] = (_Float16)(0.002f * s); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    const float c = 1.0001f, d = 0.0003f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < active_waves) {
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / 4; ++j) {
                    const int q = (i * (NV / 4) + j) & 7;
                    if (KIND == 0) x[q] = __builtin_fmaf(x[q], c, d);
                    else if (KIND == 1) x[q] = __builtin_amdgcn_exp2f(x[q]) * 0.5f;
                    else { const float4 v = sm[(lane + 64 * q + it) & 1023]; x[q] += v.x + v.w; }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // 1 MFMA
                if (NV > 0) __builtin_amdgcn_sched_group_barrier(KIND == 2 ? 0x100 | 0x002 : 0x002, KIND == 2 ? NV / 2 : (KIND == 1 ? NV / 2 : NV / 4), 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NV, int KIND>
void run(const char* name, int active) {
    const int blocks = 256, n = 2000;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    hipLaunchKernelGGL((kb<NV, KIND>), dim3(blocks), dim3(512), 0, 0, 10, active, out, ticks);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((kb<NV, KIND>), dim3(blocks), dim3(512), 0, 0, n, active, out, ticks);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * blocks);
    (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < active; ++w) m = m > h[i * 8 + w] ? m : h[i * 8 + w]; mx += m; }
    mx /= blocks;
    const int wps = active / 4;
    printf("%-34s waves/SIMD=%d  %8.1f clk per iteration per SIMD   (MFMA pipe %d, VALU issue ~%d)\n", name, wps, mx / n, wps * 128, wps * NV * 4);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<0, 0>("4 MFMA only", 4);
    run<0, 0>("4 MFMA only", 8);
    run<16, 0>("4 MFMA + 16 FMA", 4);
    run<16, 0>("4 MFMA + 16 FMA", 8);
    run<32, 0>("4 MFMA + 32 FMA", 4);
    run<32, 0>("4 MFMA + 32 FMA", 8);
    run<64, 0>("4 MFMA + 64 FMA", 8);
    run<16, 1>("4 MFMA + 16 (exp2+mul)", 8);
    run<16, 2>("4 MFMA + 16 (ds_read_b128+2 add)", 8);
    return 0;
}
