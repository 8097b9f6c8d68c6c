// Can a SIMD overlap one wave's MFMAs with another wave's VALU work?  512 threads = 2 waves per SIMD.
//   mode 0: all 8 waves MFMA           mode 1: all 8 waves VALU (fp32 FMA chains)     mode 2: waves 0-3 MFMA, 4-7 VALU (one of each per SIMD)
//   mode 3: waves 0-3 MFMA, 4-7 idle   mode 4: waves 0-3 idle, 4-7 VALU               mode 5: 2: but VALU = v_exp_f32 (transcendental)
//   mode 6: waves 0-3 MFMA, 4-7 LDS reads (ds_read_b128)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void do_mfma(int n, float* sink, int lane) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)(0.001f * lane); b[s] = (_Float16)(0.002f * s); }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
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
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], c, d);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    *sink = s;
}
__device__ __forceinline__ void do_exp(int n, float* sink, int lane) {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
    *sink = s;
}
__device__ __forceinline__ void do_lds(int n, float* sink, int lane, const float4* sm) {
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 v = sm[(lane + 64 * i + it) & 1023]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    *sink = acc.x + acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(512) void kb(int mode, int prio_valu, int swap, int n_mfma, int n_valu, float* out, unsigned long long* ticks) {
    __shared__ float4 sm[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 512) sm[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0;
    const bool first = swap ? wave >= 4 : wave < 4;
    if (!first && prio_valu) __builtin_amdgcn_s_setprio(3);
    if (first && prio_valu < 0) __builtin_amdgcn_s_setprio(3);
    if (mode == 0 || ((mode == 2 || mode == 3 || mode == 5 || mode == 6) && first)) do_mfma(n_mfma, &r, lane);
    else if (mode == 1 || ((mode == 2 || mode == 4) && !first)) do_valu(n_valu, &r, lane);
    else if (mode == 5 && !first) do_exp(n_valu, &r, lane);
    else if (mode == 6 && !first) do_lds(n_valu, &r, lane, sm);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + tid] = r;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int blocks = 256;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 4 * 512 * blocks); hipMalloc(&ticks, 8 * 8 * blocks);
    const int NM = 2000, NV = 8000;   // 8000 MFMAs (x32 clk = 256k) vs 64000 FMAs (x4 clk = 256k) per wave
    const char* names[] = {"all MFMA", "all VALU fma", "MFMA | VALU fma", "MFMA | idle", "idle | VALU fma", "MFMA | v_exp", "MFMA | ds_read_b128"};
    for (int cfg = 0; cfg < 6; ++cfg)
    for (int mode = 0; mode < 7; ++mode) {
        const int prio = cfg == 1 || cfg == 3 ? 1 : (cfg == 4 || cfg == 5 ? -1 : 0), swap = cfg == 2 || cfg == 3 || cfg == 5;
        if (cfg > 0 && mode != 2 && mode != 5 && mode != 6) continue;
        if (mode == 0) printf("--\n");
        if (mode == 2) printf("[VALU-side prio %d (-1: MFMA side), MFMA waves are the %s ones]\n", prio, swap ? "younger (4-7)" : "older (0-3)");
        hipLaunchKernelGGL(kb, dim3(blocks), dim3(512), 0, 0, mode, prio, swap, 10, 10, out, ticks);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kb, dim3(blocks), dim3(512), 0, 0, mode, prio, swap, NM, NV, out, ticks);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * blocks);
        hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int i = 0; i < blocks; ++i) for (int w = 0; w < 8; ++w) ((swap ? w >= 4 : w < 4) ? a : b) += h[i * 8 + w];
        a /= 4 * blocks; b /= 4 * blocks;
        printf("%-22s MFMA-side %9.0f ticks   other-side %9.0f ticks   (MFMA alone = %d, FMA alone = %d cycles nominal)\n", names[mode], a, b, NM * 4 * 32, NV * 8 * 4);
    }
    return 0;
}
