// After how many wait states may a VALU instruction read the result of v_mfma_f32_32x32x16_f16 (gfx950)?  The compiler's hazard recognizer
// allows it after 11 (8 passes + 3).  One asm block, fixed registers: D = A.B (A all ones, B = per-lane integer), <N wait states>, copy D out.
// D is preset to a sentinel, so a result read too early shows the sentinel (or a partial sum) instead of the exact expectation.
//     hipcc --offload-arch=gfx950 -O3 -o tools/mfma_result_latency tools/mfma_result_latency.hip && tools/mfma_result_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define RUN(WAITSTR, FIRSTREG, LASTREG)                                                                                  \
    asm volatile("v_mov_b32 v96, 0x3c003c00\n v_mov_b32 v97, 0x3c003c00\n v_mov_b32 v98, 0x3c003c00\n v_mov_b32 v99, 0x3c003c00\n" \
                 "v_mov_b32 v100, %[b]\n v_mov_b32 v101, %[b]\n v_mov_b32 v102, %[b]\n v_mov_b32 v103, %[b]\n"            \
                 "v_mov_b32 " FIRSTREG ", -1.0\n v_mov_b32 " LASTREG ", -1.0\n"                                          \
                 "s_nop 7\n"                                                                                            \
                 "v_mfma_f32_32x32x16_f16 v[104:119], v[96:99], v[100:103], 0\n" WAITSTR                                 \
                 "v_mov_b32 %[d0], " FIRSTREG "\n v_mov_b32 %[d1], " LASTREG "\n"                                        \
                 "s_nop 15\n s_nop 15\n"                                                                                \
                 : [d0] "=v"(d0), [d1] "=v"(d1)                                                                         \
                 : [b] "v"(bpk)                                                                                         \
                 : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",   \
                   "v113", "v114", "v115", "v116", "v117", "v118", "v119")

template <int N>
__global__ void k(int iters, unsigned* bad) {
    const int lane = threadIdx.x & 63;
    unsigned b0 = 0, b1 = 0;
    for (int it = 0; it < iters; ++it) {
        const int vb = (lane + 3 * it + (int)blockIdx.x) % 61;
        _Float16 hb = (_Float16)(float)vb;
        uint16_t ub;
        __builtin_memcpy(&ub, &hb, 2);
        const uint32_t bpk = (uint32_t)ub * 0x10001u;
        float d0, d1;
        if (N == 4) RUN("s_nop 3\n", "v104", "v119");
        else if (N == 8) RUN("s_nop 7\n", "v104", "v119");
        else if (N == 10) RUN("s_nop 9\n", "v104", "v119");
        else if (N == 11) RUN("s_nop 10\n", "v104", "v119");
        else if (N == 12) RUN("s_nop 11\n", "v104", "v119");
        else if (N == 14) RUN("s_nop 13\n", "v104", "v119");
        else if (N == 16) RUN("s_nop 15\n", "v104", "v119");
        else RUN("s_nop 15\n s_nop 3\n", "v104", "v119");
        const float want = (float)(8 * vb + 8 * __shfl_xor(vb, 32));
        b0 += d0 != want;
        b1 += d1 != want;
    }
    if (b0) atomicAdd(bad, b0);
    if (b1) atomicAdd(bad + 1, b1);
}

template <int N>
void run(int grid, int block, int iters, unsigned* d) {
    (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k<N>, dim3(grid), dim3(block), 0, 0, iters, d);
    unsigned h[2];
    (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("  %2d wait states: first result register wrong %10u, last result register wrong %10u   of %lld\n", N, h[0], h[1], (long long)grid * (block / 64) * iters * 64);
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 8);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int grid = cfg ? 320 : 4096, block = cfg ? 256 : 64, iters = 2000;
        printf("grid %d x %d threads, %d iterations per wave:\n", grid, block, iters);
        run<4>(grid, block, iters, d);
        run<8>(grid, block, iters, d);
        run<10>(grid, block, iters, d);
        run<11>(grid, block, iters, d);
        run<12>(grid, block, iters, d);
        run<14>(grid, block, iters, d);
        run<16>(grid, block, iters, d);
        run<20>(grid, block, iters, d);
    }
    return 0;
}
