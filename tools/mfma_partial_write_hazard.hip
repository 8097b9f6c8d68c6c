// How many wait states does an MFMA need behind a 16-bit partial write (v_fma_mixhi_f16) of one of its source registers?  (gfx950)
//
// Each wave builds a B operand of v_mfma_f32_32x32x16_f16 with v_fma_mix{lo,hi}_f16 (the operand split of gcdm_edge_x3.hip.h), issues the MFMA
// WAIT wait states behind the last v_fma_mixhi_f16 -- everything in ONE asm block with fixed registers, so the compiler cannot move or pad
// anything -- and compares D with the exact expectation (A = all ones, inputs chosen so that every value is an integer).  The inputs change from
// iteration to iteration, so an operand read too early shows up as the previous iteration's value.
//     hipcc --offload-arch=gfx950 -O3 -o tools/mfma_partial_write_hazard tools/mfma_partial_write_hazard.hip && tools/mfma_partial_write_hazard
// Background: DESIGN.md 3.4 (x3_settle).  The compiler's hazard recognizer leaves ONE wait state between an inline-asm definition and its consumer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define SPLIT_AND_MFMA3(PRESTR, WAITSTR, POSTSTR)                                                                                          \
    asm volatile("v_mov_b32 v96, 0x3c003c00\n v_mov_b32 v97, 0x3c003c00\n v_mov_b32 v98, 0x3c003c00\n v_mov_b32 v99, 0x3c003c00\n" \
                 "v_fma_mixlo_f16 v100, %[x0], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixlo_f16 v101, %[x2], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixlo_f16 v102, %[x4], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixlo_f16 v103, %[x6], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixhi_f16 v100, %[x1], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixhi_f16 v101, %[x3], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 "v_fma_mixhi_f16 v102, %[x5], %[pre], 0 op_sel_hi:[0,0,0]\n"                                           \
                 PRESTR "v_fma_mixhi_f16 v103, %[x7], %[pre], 0 op_sel_hi:[0,0,0]\n" WAITSTR                                   \
                 "v_mfma_f32_32x32x16_f16 v[104:119], v[96:99], v[100:103], 0\n" POSTSTR                                \
                 "s_nop 15\n s_nop 15\n"                                                                                \
                 "v_mov_b32 %[d0], v104\n v_mov_b32 %[d1], v119\n"                                                      \
                 : [d0] "=v"(d0), [d1] "=v"(d1)                                                                         \
                 : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [pre] "s"(pre) \
                 : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",   \
                   "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",    \
                   "v130", "v131", "v132", "v133", "v134", "v135")

#define SPLIT_AND_MFMA(PRESTR, WAITSTR) SPLIT_AND_MFMA3(PRESTR, WAITSTR, "")
// BEFORE = 1: an independent MFMA is issued right in front of the last partial write (the situation inside a k-block loop)
#define PRE_MFMA "v_mfma_f32_32x32x16_f16 v[120:135], v[96:99], v[96:99], 0\n"
template <int WAIT, int BEFORE>
__global__ void k(int iters, unsigned* bad, unsigned* bad_lanes) {
    const int lane = threadIdx.x & 63;
    const float pre = 1.0f / 2048.0f;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float x[8];
        int m = 0;
        for (int j = 0; j < 8; ++j) {
            const int v = (lane * 8 + j + 37 * it + 11 * (int)blockIdx.x) % 997;
            x[j] = 2048.0f * (float)v;             // image = v exactly
            m += v;
        }
        float d0, d1;
        if (BEFORE == 0) {
            if (WAIT == 0) SPLIT_AND_MFMA("", "");
            else if (WAIT == 1) SPLIT_AND_MFMA("", "s_nop 0\n");
            else if (WAIT == 2) SPLIT_AND_MFMA("", "s_nop 1\n");
            else if (WAIT == 3) SPLIT_AND_MFMA("", "s_nop 2\n");
            else SPLIT_AND_MFMA("", "s_nop 4\n");
        } else if (BEFORE == 2) {            // write-after-read: a 16-bit partial write to a B register right behind the MFMA that reads it (2 wait states in front)
            if (WAIT == 0) SPLIT_AND_MFMA3("", "s_nop 1\n", "v_fma_mixlo_f16 v100, %[x7], %[pre], 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 v103, %[x0], %[pre], 0 op_sel_hi:[0,0,0]\n");
            else if (WAIT == 1) SPLIT_AND_MFMA3("", "s_nop 1\n", "s_nop 0\n v_fma_mixlo_f16 v100, %[x7], %[pre], 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 v103, %[x0], %[pre], 0 op_sel_hi:[0,0,0]\n");
            else if (WAIT == 2) SPLIT_AND_MFMA3("", "s_nop 1\n", "s_nop 3\n v_fma_mixlo_f16 v100, %[x7], %[pre], 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 v103, %[x0], %[pre], 0 op_sel_hi:[0,0,0]\n");
            else if (WAIT == 3) SPLIT_AND_MFMA3("", "s_nop 1\n", "v_mov_b32 v100, 0\n v_mov_b32 v103, 0\n");                     // the same with full 32-bit writes
            else SPLIT_AND_MFMA3("", "s_nop 1\n", "v_fma_mixlo_f16 v96, %[x7], %[pre], 0 op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 v99, %[x0], %[pre], 0 op_sel_hi:[0,0,0]\n");   // ... to an A register
        } else {
            if (WAIT == 0) SPLIT_AND_MFMA(PRE_MFMA, "");
            else if (WAIT == 1) SPLIT_AND_MFMA(PRE_MFMA, "s_nop 0\n");
            else if (WAIT == 2) SPLIT_AND_MFMA(PRE_MFMA, "s_nop 1\n");
            else if (WAIT == 3) SPLIT_AND_MFMA(PRE_MFMA, "s_nop 2\n");
            else SPLIT_AND_MFMA(PRE_MFMA, "s_nop 4\n");
        }
        const int want = m + __shfl_xor(m, 32);    // column (lane & 31): the k-slots of both half-waves
        if (d0 != (float)want || d1 != (float)want) { ++nbad; atomicOr(&bad_lanes[(lane & 31) >> 4], 1u); }
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int WAIT, int BEFORE>
void run(int grid, int block, int iters, unsigned* d) {
    (void)hipMemset(d, 0, 16);
    hipLaunchKernelGGL((k<WAIT, BEFORE>), dim3(grid), dim3(block), 0, 0, iters, d, d + 1);
    unsigned h[3];
    (void)hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("  %s wait states %d: %9u wrong results of %lld MFMAs  (columns 0-15 hit: %u, columns 16-31 hit: %u)\n", BEFORE == 2 ? "[write-after-read: 0/1/4 wait states, full writes, A register]" : BEFORE ? "[MFMA in front of the last partial write]" : "[plain]", WAIT, h[0], (long long)grid * (block / 64) * iters * 64, h[1], h[2]);
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 16);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int grid = cfg ? 320 : 4096, block = cfg ? 256 : 64, iters = 2000;
        printf("grid %d x %d threads, %d iterations per wave:\n", grid, block, iters);
        run<0, 0>(grid, block, iters, d);
        run<1, 0>(grid, block, iters, d);
        run<2, 0>(grid, block, iters, d);
        run<0, 1>(grid, block, iters, d);
        run<1, 1>(grid, block, iters, d);
        run<2, 1>(grid, block, iters, d);
        run<3, 1>(grid, block, iters, d);
        run<0, 2>(grid, block, iters, d);
        run<1, 2>(grid, block, iters, d);
        run<2, 2>(grid, block, iters, d);
        run<3, 2>(grid, block, iters, d);
        run<5, 2>(grid, block, iters, d);
    }
    return 0;
}
