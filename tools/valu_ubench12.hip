// Does a transcendental (v_exp_f32 / v_rcp_f32) overlap with plain VALU instructions of the SAME wave or of the OTHER wave of its SIMD?  (round 5)
// The SiLU of the edge kernel costs what its instructions cost one after the other (removing it returns 7.9 % of the tile, profiles/r05_ab_log.txt run 19);
// if the transcendental pipe ran beside the main VALU, interleaving the two kinds -- within a wave, or by running one wave's SiLU against the other
// wave's splits -- would hide most of it.
//   T   : 16 independent chains of v_exp_f32                              (transcendental only)
//   F   : 16 x 3 independent v_fma_f32                                     (plain VALU only)
//   TF  : the same instructions interleaved 1 : 3 in ONE wave's stream
//   T|F : waves 0..k-1 of each SIMD run T, the others run F (two kinds on one SIMD at the same time; wall = slowest wave)
// 256 workgroups x (256 | 512) threads = 1 | 2 waves per SIMD; prints shader clocks per loop iteration (16 exp and / or 48 fma per wave).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_ubench12 tools/valu_ubench12.hip && tools/valu_ubench12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ __launch_bounds__(512) void kb(int n, float* out, unsigned long long* ticks) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float x[16], y[48];
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (lane + i);
    for (int i = 0; i < 48; ++i) y[i] = 0.002f * (lane + i);
    const float c = 1.0001f, d = 0.0003f;
    // K == 3: role by wave -- waves 0..3 (one per SIMD) do T, waves 4..7 (their SIMD partners) do F
    const bool doT = K == 0 || K == 2 || (K == 3 && wave < 4), doF = K == 1 || K == 2 || (K == 3 && wave >= 4);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (K == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[3 * i]) : "v"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[3 * i + 1]) : "v"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[3 * i + 2]) : "v"(c), "v"(d));
            }
        } else {
            if (doT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            }
            if (doF) {
#pragma unroll
                for (int i = 0; i < 48; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(c), "v"(d));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 48; ++i) s += y[i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int K>
void run(const char* name) {
    const int blocks = 256, N = 400;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    printf("%-64s", name);
    for (int wps = 1; wps <= 2; ++wps) {
        if (K == 3 && wps == 1) { printf("      -    "); continue; }
        const int threads = 256 * wps, nw = threads / 64;
        hipLaunchKernelGGL(kb<K>, dim3(blocks), dim3(threads), 0, 0, N, out, ticks);
        hipLaunchKernelGGL(kb<K>, dim3(blocks), dim3(threads), 0, 0, N, out, ticks);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(8 * blocks);
        (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
        double mx = 0, sumT = 0, sumF = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < nw; ++w) { const double v = (double)h[b * 8 + w] / N; mx = v > mx ? v : mx; (w < 4 ? sumT : sumF) += v; }
        if (K == 3) printf("  T waves %7.1f  F waves %7.1f", sumT / (blocks * 4), sumF / (blocks * 4));
        else printf("  %d wave(s)/SIMD: %7.1f clk/iter", wps, (sumT + sumF) / (blocks * nw));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<0>("T : 16 v_exp_f32");
    run<1>("F : 48 v_fma_f32");
    run<2>("TF: 16 v_exp_f32 interleaved 1:3 with 48 v_fma_f32 (one stream)");
    run<3>("T|F: one wave of each SIMD runs T, its partner F");
    return 0;
}
