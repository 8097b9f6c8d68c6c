// VALU issue rate of a SIMD as a function of the number of waves it hosts (1 .. 4), for the instruction kinds of the edge kernel's VALU phases.
// valu_ubench8 found a lone wave issuing one VALU instruction per ~5.9 clk and two waves one per ~3.0 clk per SIMD; the vector ALU of gfx950
// retires a wave64 fp32 instruction in 2 clk -- does a third / fourth wave buy the rest?  (round 4, DESIGN.md 3.4: the VALU phases of
// k_edge_msg_x3 run with two waves per SIMD because a wave needs 256 registers.)
//   kinds: v_fma_f32 | SiLU chain (fma exp add rcp mul) | hi/lo' split of a pair (2 mul, cvt_pk, 2 fma_mix, cvt_pk) | the finish mix of one GCP2
// 256 workgroups x (256, 512, 768, 1024) threads; prints clk per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K>
__global__ __launch_bounds__(1024) void kb(int n, float pre, float neg, float* out, unsigned long long* ticks) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = 0.001f * (lane + i); y[i] = 0.5f * x[i]; }
    const float c = 1.0001f, d = 0.0003f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (K == 0) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(c), "v"(d));
            } else if constexpr (K == 1) {
                float t, e;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x[i]), "v"(c), "v"(y[i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(e) : "v"(t));
                asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(e));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x[i]) : "v"(t), "v"(e));
            } else if constexpr (K == 2) {
                if (i & 1) continue;
                float t0_, t1_, r0, r1; unsigned hiu, lou;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0_) : "s"(pre), "v"(x[i]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1_) : "s"(pre), "v"(x[i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t0_), "v"(t1_));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x[i]));
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x[i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
                y[i] = __builtin_bit_cast(float, hiu); y[i + 1] = __builtin_bit_cast(float, lou);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += x[i] + y[i];
    out[blockIdx.x * 1024 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int K>
void run(const char* name, int per_iter) {
    const int blocks = 256, N = 400;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4 * 1024 * blocks); (void)hipMalloc(&ticks, 8 * 16 * blocks);
    printf("%-42s", name);
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 256 * wps, nw = threads / 64;
        hipLaunchKernelGGL((kb<K>), dim3(blocks), dim3(threads), 0, 0, 5, 4.8828125e-4f, -2048.f, out, ticks);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL((kb<K>), dim3(blocks), dim3(threads), 0, 0, N, 4.8828125e-4f, -2048.f, out, ticks);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(16 * blocks);
        (void)hipMemcpy(h.data(), ticks, 8 * 16 * blocks, hipMemcpyDeviceToHost);
        double a = 0;
        for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < nw; ++w) m = m > h[i * 16 + w] ? m : h[i * 16 + w]; a += m; }
        a /= blocks;
        printf("  %d waves/SIMD %5.2f clk/instr/SIMD", wps, a / ((double)N * per_iter * wps));
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<0>("v_fma_f32", 32);
    run<1>("SiLU chain (fma exp add rcp mul)", 80);
    run<2>("split of a pair (2 mul cvt_pk 2 fma_mix cvt_pk)", 48);
    return 0;
}
