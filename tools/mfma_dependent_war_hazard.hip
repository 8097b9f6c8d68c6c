// Write-after-read behind a DEPENDENT MFMA (gfx950): is it safe to overwrite a source register of an MFMA that cannot start yet because its
// C operand is the result of the MFMA issued just before it?  (The compiler does this: e.g. v_accvgpr_read into a register of the A operand
// one instruction behind such an MFMA.)  One asm block, fixed registers, exact integer expectation.
//     hipcc --offload-arch=gfx950 -O3 -o tools/mfma_dependent_war_hazard tools/mfma_dependent_war_hazard.hip && tools/mfma_dependent_war_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// v[96:99] = A (all ones), v[100:103] = B1, v[120:123] = B2 (both = per-lane integers as f16), D = v[104:119]
// D = A.B1 ; D += A.B2 (dependent) ; <POST: overwrite a source of the second MFMA> ; wait ; read D
#define CHAIN(POSTSTR)                                                                                                   \
    asm volatile("v_mov_b32 v96, 0x3c003c00\n v_mov_b32 v97, 0x3c003c00\n v_mov_b32 v98, 0x3c003c00\n v_mov_b32 v99, 0x3c003c00\n" \
                 "v_mov_b32 v100, %[b]\n v_mov_b32 v101, %[b]\n v_mov_b32 v102, %[b]\n v_mov_b32 v103, %[b]\n"            \
                 "v_mov_b32 v120, %[c]\n v_mov_b32 v121, %[c]\n v_mov_b32 v122, %[c]\n v_mov_b32 v123, %[c]\n"            \
                 "s_nop 4\n"                                                                                            \
                 "v_mfma_f32_32x32x16_f16 v[104:119], v[96:99], v[100:103], 0\n"                                        \
                 "v_mfma_f32_32x32x16_f16 v[104:119], v[96:99], v[120:123], v[104:119]\n" POSTSTR                        \
                 "s_nop 15\n s_nop 15\n s_nop 15\n"                                                                     \
                 "v_mov_b32 %[d0], v104\n v_mov_b32 %[d1], v119\n"                                                      \
                 : [d0] "=v"(d0), [d1] "=v"(d1)                                                                         \
                 : [b] "v"(bpk), [c] "v"(cpk), [z] "v"(zero)                                                             \
                 : "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",   \
                   "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123")

template <int MODE>
__global__ void k(int iters, unsigned* bad) {
    const int lane = threadIdx.x & 63;
    unsigned nbad = 0;
    const uint32_t zero = 0;
    for (int it = 0; it < iters; ++it) {
        const int vb = (lane + 3 * it + (int)blockIdx.x) % 61, vc = (5 * lane + 7 * it) % 53;
        _Float16 hb = (_Float16)(float)vb, hc = (_Float16)(float)vc;
        uint16_t ub, uc;
        __builtin_memcpy(&ub, &hb, 2);
        __builtin_memcpy(&uc, &hc, 2);
        const uint32_t bpk = (uint32_t)ub * 0x10001u, cpk = (uint32_t)uc * 0x10001u;
        float d0, d1;
        if (MODE == 0) CHAIN("");                                                      // control
        else if (MODE == 1) CHAIN("v_mov_b32 v120, %[z]\n v_mov_b32 v123, %[z]\n");    // full writes to B of the dependent MFMA, distance 1
        else if (MODE == 2) CHAIN("v_mov_b32 v96, %[z]\n v_mov_b32 v99, %[z]\n");      // full writes to A, distance 1
        else if (MODE == 3) CHAIN("s_nop 1\n v_mov_b32 v120, %[z]\n v_mov_b32 v96, %[z]\n");   // the same 2 wait states later
        else CHAIN("s_nop 7\n v_mov_b32 v120, %[z]\n v_mov_b32 v96, %[z]\n");          // ... 8 wait states later
        // column sum over k = 16 slots: 8 per half-wave, all equal to the lane's value
        const int want = 8 * (vb + vc) + 8 * (__shfl_xor(vb, 32) + __shfl_xor(vc, 32));
        if (d0 != (float)want || d1 != (float)want) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int MODE>
void run(const char* what, int grid, int block, int iters, unsigned* d) {
    (void)hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, iters, d);
    unsigned h;
    (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("  %-70s %10u wrong of %lld\n", what, h, (long long)grid * (block / 64) * iters * 64);
}

int main() {
    unsigned* d;
    (void)hipMalloc(&d, 4);
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int grid = cfg ? 320 : 4096, block = cfg ? 256 : 64, iters = 2000;
        printf("grid %d x %d threads, %d iterations per wave:\n", grid, block, iters);
        run<0>("control (nothing behind the dependent MFMA)", grid, block, iters, d);
        run<1>("full writes to its B registers, next instruction", grid, block, iters, d);
        run<2>("full writes to its A registers, next instruction", grid, block, iters, d);
        run<3>("writes to A and B, 2 wait states later", grid, block, iters, d);
        run<4>("writes to A and B, 8 wait states later", grid, block, iters, d);
    }
    return 0;
}
