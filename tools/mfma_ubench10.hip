// Lockstep vs enforced de-phasing of the two waves of a SIMD (round 4).
//
// One residual message GCP2 of k_edge_msg_x3 on a 64-edge tile is, per CU: a scalar GEMM (8 M-tiles x 2 N-tiles x 18 k-blocks x 3
// v_mfma_f32_32x32x16_f16, weights streamed from L2, activations' hi / lo' images read from LDS) and the "finish" work on its result
// (32 accumulator values per lane and wave: merge, SiLU, hi / lo' split for the gate contraction, 12 gate MFMAs, partial store, residual
// add, hi / lo' split of the new state, image stores: ~480 VALU + 24 LDS + 12 MFMA).
//   L  lockstep (the kernel today): all 8 waves run the GEMM (wave = one M-tile x both N-tiles: 2 KB of weights per k-block and wave),
//      barrier, all 8 waves run the finish, barrier.
//   D  de-phased halves: waves 0-3 (one per SIMD) own edges 0-31, waves 4-7 edges 32-63, each wave two M-tiles x one N-tile (4 KB of
//      weights per k-block and wave).  Slot 1: waves 0-3 GEMM, waves 4-7 finish; barrier; slot 2: roles swapped; barrier.
//   S  same-wave skew: every wave keeps its M-tile; slot 1: GEMM of edges 0-31 with the finish work of edges 32-63 (of the previous GCP2) cut into
//      18 stages issued between the MFMAs of its k-blocks; barrier; slot 2: GEMM of edges 32-63 with the finish work of edges 0-31; barrier.
// Same MFMAs and the same finish work per GCP2 in all three; D and S stream every weight byte twice per 64 edges.  Prints shader cycles per GCP2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void split16x2(float x0, float x1, h2& hi, h2& lo, float pre, float neg) {
    uint32_t hiu, lou;
    const float t0 = x0 * pre, t1 = x1 * pre;
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t0), "v"(t1));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    __builtin_memcpy(&hi, &hiu, 4);
    __builtin_memcpy(&lo, &lou, 4);
}

constexpr int KB = 18, TP = 65, PD = 2;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WPool {                         // weight stream through buffer loads, scalar block offsets (as the kernel)
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    __device__ __forceinline__ h8 ld(uint32_t soff) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        h8 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    }
};

// tile GEMM: MT M-tiles x NT N-tiles, weights streamed with a ring of PD + 1 register sets, B operands one block ahead (the kernel's tile_gemm_x3s)
template <int MT, int NT, int PER = 0, class Hook = void (*)(int), int MIDBAR = 0>
__device__ __forceinline__ void gemm(f32x16 (&am)[MT][NT], f32x16 (&al)[MT][NT], const WPool& wp, uint32_t wH, uint32_t wL, const h8* xh, const h8* xl, int lane,
                                     Hook hook = [](int) {}) {
    constexpr int R = PD + 1;
    h8 ah[R][MT], alo[R][MT], bh[2][NT], bl[2][NT];
    const int boff = (lane >> 5) * TP + (lane & 31);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < PD; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) { ah[r][m] = wp.ld(wH + (m * KB + r) * 1024); alo[r][m] = wp.ld(wL + (m * KB + r) * 1024); }
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = xh[boff + 32 * n]; bl[0][n] = xl[boff + 32 * n]; }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        if (MIDBAR > 0 && k == MIDBAR) __syncthreads();          // schedule M: the workgroup barrier between the two halves of a GEMM
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[(k + PD) % R][m] = wp.ld(wH + (m * KB + (k + PD < KB ? k + PD : KB - 1)) * 1024);
            alo[(k + PD) % R][m] = wp.ld(wL + (m * KB + (k + PD < KB ? k + PD : KB - 1)) * 1024);
        }
        const int kn = k + 1 < KB ? k + 1 : k;
#pragma unroll
        for (int n = 0; n < NT; ++n) { bh[(k + 1) & 1][n] = xh[boff + kn * 2 * TP + 32 * n]; bl[(k + 1) & 1][n] = xl[boff + kn * 2 * TP + 32 * n]; }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(ah[k % R][m], bh[k & 1][n], k == 0 ? zero : am[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ah[k % R][m], bl[k & 1][n], k == 0 ? zero : al[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(alo[k % R][m], bh[k & 1][n], al[m][n]);
        if constexpr (PER > 0) {                     // S: a stage of the OTHER half's finish work between this block's MFMAs
            hook(k);
#pragma unroll
            for (int i = 0; i < 3 * MT * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x400 | 0x200 | 0x100 | 0x020, PER, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // (as in the kernel: the scheduler works on one k-block at a time)
    }
}

// Schedule Z (round 6): ONE accumulator per N-tile.  The three products land at the same scale when W_hi . X_lo' reads a THIRD, natural-scale image of W_hi
// (streamed from wN: +50 % weight bytes), so there is no `am + al * inv` merge and the wave holds 32 accumulator registers less.
template <int MT, int NT>
__device__ __forceinline__ void gemm1(f32x16 (&am)[MT][NT], const WPool& wp, uint32_t wH, uint32_t wL, uint32_t wN, const h8* xh, const h8* xl, int lane) {
    constexpr int R = PD + 1;
    h8 ah[R][MT], alo[R][MT], an[R][MT], bh[2][NT], bl[2][NT];
    const int boff = (lane >> 5) * TP + (lane & 31);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < PD; ++r)
#pragma unroll
        for (int m = 0; m < MT; ++m) { ah[r][m] = wp.ld(wH + (m * KB + r) * 1024); alo[r][m] = wp.ld(wL + (m * KB + r) * 1024); an[r][m] = wp.ld(wN + (m * KB + r) * 1024); }
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = xh[boff + 32 * n]; bl[0][n] = xl[boff + 32 * n]; }
#pragma unroll
    for (int k = 0; k < KB; ++k) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int kk = k + PD < KB ? k + PD : KB - 1;
            ah[(k + PD) % R][m] = wp.ld(wH + (m * KB + kk) * 1024);
            alo[(k + PD) % R][m] = wp.ld(wL + (m * KB + kk) * 1024);
            an[(k + PD) % R][m] = wp.ld(wN + (m * KB + kk) * 1024);
        }
        const int kn = k + 1 < KB ? k + 1 : k;
#pragma unroll
        for (int n = 0; n < NT; ++n) { bh[(k + 1) & 1][n] = xh[boff + kn * 2 * TP + 32 * n]; bl[(k + 1) & 1][n] = xl[boff + kn * 2 * TP + 32 * n]; }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(ah[k % R][m], bh[k & 1][n], k == 0 ? zero : am[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(an[k % R][m], bl[k & 1][n], am[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(alo[k % R][m], bh[k & 1][n], am[m][n]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the finish work of schedule Z: as `finish`, the gate contraction on one accumulator as well (third gate-weight image at GWN), no merges
template <int NB>
__device__ __forceinline__ void finish1(const f32x16 (&p)[NB], f32x16 (&st)[NB], const WPool& wp, uint32_t GW, uint32_t GWN, char* XH,
                                        char* XL, float* PG, int slot, int lane, float pre, float neg) {
    const int half = lane >> 5, l31 = lane & 31;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 act[NB], gm = zero;
    h8 gwh[NB][2], gwl[NB][2], gwn[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            gwh[b][j] = wp.ld(GW + ((slot * NB + b) * 2 + j) * 1024); gwl[b][j] = wp.ld(GW + (32 + (slot * NB + b) * 2 + j) * 1024);
            gwn[b][j] = wp.ld(GWN + ((slot * NB + b) * 2 + j) * 1024);
        }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = p[b][r];
            act[b][r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x));
        }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h8 bh, bl;
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                h2 hi, lo;
                split16x2(act[b][8 * j + s], act[b][8 * j + s + 1], hi, lo, pre, neg);
                bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
            }
            asm("s_nop 1" : "+v"(bh), "+v"(bl));
            gm = MFMA16(gwh[b][j], bh, gm);
            gm = MFMA16(gwn[b][j], bl, gm);
            gm = MFMA16(gwl[b][j], bh, gm);
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gm[4 * t + i];
        *(v4f*)(PG + (((slot & 3) * 64 + l31) * 32 + 4 * ((2 * t + half) ^ (l31 & 7)))) = v;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[b][r] += act[b][r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h4 vh, vl;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                h2 hi, lo;
                split16x2(st[b][4 * q + t], st[b][4 * q + t + 1], hi, lo, pre, neg);
                vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
            }
            const int off = ((4 * (slot & 3) + q) * TP + 32 * b + l31) * 16 + 8 * half;
            *(h4*)(XH + off) = vh;
            *(h4*)(XL + off) = vl;
        }
    }
}

// finish work on NV = 2 blocks of 16 accumulator values: SiLU, gate contraction, residual add, state images
template <int NB>
__device__ __forceinline__ void finish(const f32x16 (&p)[NB], f32x16 (&st)[NB], const WPool& wp, uint32_t GW, char* XH,
                                       char* XL, float* PG, int slot, int lane, float pre, float neg, float inv) {
    const int half = lane >> 5, l31 = lane & 31;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 act[NB], gm = zero, gl = zero;
    h8 gwh[NB][2], gwl[NB][2];                      // gate weights of this wave's channels, requested ahead of the SiLU (the kernel's gate_prefetch)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) { gwh[b][j] = wp.ld(GW + ((slot * NB + b) * 2 + j) * 1024); gwl[b][j] = wp.ld(GW + (32 + (slot * NB + b) * 2 + j) * 1024); }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = p[b][r];
            act[b][r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x));
        }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h8 bh, bl;
#pragma unroll
            for (int s = 0; s < 8; s += 2) {
                h2 hi, lo;
                split16x2(act[b][8 * j + s], act[b][8 * j + s + 1], hi, lo, pre, neg);
                bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
            }
            asm("s_nop 1" : "+v"(bh), "+v"(bl));
            gm = MFMA16(gwh[b][j], bh, gm);
            gl = MFMA16(gwh[b][j], bl, gl);
            gl = MFMA16(gwl[b][j], bh, gl);
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) {                   // gate partial of this wave -> LDS (4 x ds_write_b128)
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gm[4 * t + i] + gl[4 * t + i] * inv;
        *(v4f*)(PG + (((slot & 3) * 64 + l31) * 32 + 4 * ((2 * t + half) ^ (l31 & 7)))) = v;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[b][r] += act[b][r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h4 vh, vl;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                h2 hi, lo;
                split16x2(st[b][4 * q + t], st[b][4 * q + t + 1], hi, lo, pre, neg);
                vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
            }
            const int off = ((4 * (slot & 3) + q) * TP + 32 * b + l31) * 16 + 8 * half;
            *(h4*)(XH + off) = vh;
            *(h4*)(XL + off) = vl;
        }
    }
}

// The finish work of ONE block of 16 values cut into 18 stages (one per k-block of the hosting GEMM)
struct FinStage {
    f32x16 p, act, gm, gl;
    h8 gwh[2], gwl[2], bh, bl;
    f32x16* st;
    char *XH, *XL; float* PG;
    int slot, lane, b;
    float pre, neg, inv;
    __device__ __forceinline__ void silu4(int i) {
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) { const float x = p[r]; act[r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }
    }
    __device__ __forceinline__ void gsplit(int j, int s0) {
#pragma unroll
        for (int s = s0; s < s0 + 4; s += 2) {
            h2 hi, lo;
            split16x2(act[8 * j + s], act[8 * j + s + 1], hi, lo, pre, neg);
            bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
        }
    }
    __device__ __forceinline__ void gmfma(int j) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        asm("s_nop 1" : "+v"(bh), "+v"(bl));
        gm = MFMA16(gwh[j], bh, j == 0 ? zero : gm);
        gl = MFMA16(gwh[j], bl, j == 0 ? zero : gl);
        gl = MFMA16(gwl[j], bh, gl);
    }
    __device__ __forceinline__ void gout(int t0) {
        const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int t = t0; t < t0 + 2; ++t) {
            v4f v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gm[4 * t + i] + gl[4 * t + i] * inv;
            *(v4f*)(PG + (((slot & 3) * 64 + 32 * b + l31) * 32 + 4 * ((2 * t + half) ^ (l31 & 7)))) = v;
        }
    }
    __device__ __forceinline__ void image(int q) {
        const int half = lane >> 5, l31 = lane & 31;
        h4 vh, vl;
#pragma unroll
        for (int t = 0; t < 4; ++t) (*st)[4 * q + t] += act[4 * q + t];
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            h2 hi, lo;
            split16x2((*st)[4 * q + t], (*st)[4 * q + t + 1], hi, lo, pre, neg);
            vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
        }
        const int off = ((4 * (slot & 3) + q) * TP + 32 * b + l31) * 16 + 8 * half;
        *(h4*)(XH + off) = vh;
        *(h4*)(XL + off) = vl;
    }
    __device__ __forceinline__ void run(int k) {
        if (k < 4) silu4(k);
        else if (k == 4) gsplit(0, 0);
        else if (k == 5) gsplit(0, 4);
        else if (k == 6) gmfma(0);
        else if (k == 7) gsplit(1, 0);
        else if (k == 8) gsplit(1, 4);
        else if (k == 9) gmfma(1);
        else if (k == 10) gout(0);
        else if (k == 11) gout(2);
        else if (k < 16) image(k - 12);
    }
};

// the same NUMBER of VALU instructions per stage as FinStage on average (12), but plain independent v_fma_f32 (what mfma_ubench5 hides at ~65 %)
struct FmaStage {
    float x[12];
    float c, d;
    __device__ __forceinline__ void run(int) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
    }
};
// transcendental-only filler: 6 x (v_exp_f32, v_rcp_f32) per stage
struct TransStage {
    float x[12];
    __device__ __forceinline__ void run(int) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])); asm volatile("v_rcp_f32 %0, %0" : "+v"(x[6 + i])); }
    }
};
// split-only filler: 2 pairs per stage (12 instructions: 2 mul, cvt_pk, 2 fma_mix, cvt_pk each)
struct SplitStage {
    float x[8];
    unsigned acc;
    float pre, neg;
    __device__ __forceinline__ void run(int k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            h2 hi, lo;
            split16x2(x[(2 * i + 4 * (k & 1)) & 7], x[(2 * i + 1 + 4 * (k & 1)) & 7], hi, lo, pre, neg);
            acc ^= __builtin_bit_cast(unsigned, hi) + __builtin_bit_cast(unsigned, lo);
        }
    }
};

template <int VAR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void kb(int n, float pre, float neg, float inv, const h8* __restrict__ W, float* out,
                                                                                     unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;                                   // GEMM operand images [36][65] x 16 B (hi), then lo'
    char* XL = XH + 36 * TP * 16;
    char* YH = XL + 36 * TP * 16;                      // images written by the finish work (a second buffer: no write / read ordering to model)
    char* YL = YH + 16 * TP * 16;
    float* PG = (float*)(YL + 16 * TP * 16);           // gate partials [4][64][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 36 * TP; i += 512) {
        h8 v;
        for (int s = 0; s < 8; ++s) v[s] = (_Float16)(0.001f * ((i + s) & 63));
        ((h8*)XH)[i] = v;
    }
    __syncthreads();
    WPool wp;
    wp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(W), 0, (3 * 8 * (KB + 4) + 96) * 1024, 0x00020000);
    wp.voff = (uint32_t)lane * 16u;
    const uint32_t wH = 0, wL = 8 * (KB + 4) * 1024;     // [8 M-tiles][KB][64] hi, then lo'
    const h8* xh = (const h8*)XH;
    const h8* xl = (const h8*)XL;
    const uint32_t GW = 2 * 8 * (KB + 4) * 1024;
    [[maybe_unused]] const uint32_t wN = GW + 64 * 1024, GWN = wN + 8 * (KB + 4) * 1024;      // schedule Z: natural-scale W_hi images (GEMM, gate)
    f32x16 st[2], p[2];
    for (int b = 0; b < 2; ++b)
        for (int r = 0; r < 16; ++r) { st[b][r] = 0.1f * r; p[b][r] = 0.01f * (lane + r); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        if (VAR == 0) {                                 // L: lockstep
            f32x16 am[1][2], al[1][2];
            gemm<1, 2>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh, xl, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) p[b][r] = am[0][b][r] + al[0][b][r] * inv;
            __syncthreads();
            finish<2>(p, st, wp, GW, YH, YL, PG, wave, lane, pre, neg, inv);
            __syncthreads();
        } else if (VAR == 20) {                         // Z: lockstep on ONE accumulator per N-tile (third weight image, no merges)
            f32x16 am[1][2];
            gemm1<1, 2>(am, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, wN + (uint32_t)wave * KB * 1024, xh, xl, lane);
            p[0] = am[0][0]; p[1] = am[0][1];
            __syncthreads();
            finish1<2>(p, st, wp, GW, GWN, YH, YL, PG, wave, lane, pre, neg);
            __syncthreads();
        } else if (VAR == 11 || VAR == 12) {
            // M (round 6): M-TILE de-phasing of the two waves of a SIMD.  Every wave keeps the kernel's tile (its M-tile x both N-tiles: nothing is streamed twice), but
            // waves 4-7 (the SIMD partners of waves 0-3) run half a GEMM behind: per GCP2 four slots separated by barriers,
            //     waves 0-3:  GEMM k-blocks 0-7 | GEMM 8-17 | finish N-tile 0 | finish N-tile 1 |
            //     waves 4-7:  finish N-tile 1'  | GEMM 0-7  | GEMM 8-17       | finish N-tile 0 |
            // so that in two of the four slots one wave of every SIMD is in its GEMM while the other is in a VALU phase.  Data flow of the real layer chain: waves 0-3 contract
            // first over the channels of M-tiles 0-3 (their own finish, complete one slot earlier), then over those of M-tiles 4-7 (complete when the first half ends).
            // VAR 12: the same with `s_setprio 1` on whichever wave is in a VALU phase (the GEMM wave needs one issue slot in eight).
            f32x16 am[1][2], al[1][2];
            const bool lead = wave < 4;
            auto fin_half = [&](int hf) {
                f32x16 ph[1] = {p[hf]}, sh[1] = {st[hf]};
                if (VAR == 12) __builtin_amdgcn_s_setprio(1);
                finish<1>(ph, sh, wp, GW, YH, YL, PG, wave * 2 + hf, lane, pre, neg, inv);
                if (VAR == 12) __builtin_amdgcn_s_setprio(0);
                st[hf] = sh[0];
            };
            if (lead) {
                gemm<1, 2, 0, void (*)(int), 8>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh, xl, lane);     // slots 0, 1 (barrier inside)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[b][r] = am[0][b][r] + al[0][b][r] * inv;
                __syncthreads();
                fin_half(0);                            // slot 2
                __syncthreads();
                fin_half(1);                            // slot 3
                __syncthreads();
            } else {
                fin_half(1);                            // slot 0: the previous GCP2's second half
                __syncthreads();
                gemm<1, 2, 0, void (*)(int), 8>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh, xl, lane);     // slots 1, 2
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[b][r] = am[0][b][r] + al[0][b][r] * inv;
                __syncthreads();
                fin_half(0);                            // slot 3
                __syncthreads();
            }
        } else if (VAR >= 5) {                          // F: the S loop with synthetic fillers of ONE kind (which kinds hide under the wave's own MFMAs?)
            FmaStage fm; TransStage tr; SplitStage sp;
            for (int i = 0; i < 12; ++i) { fm.x[i] = 0.001f * (lane + i); tr.x[i] = 0.01f * (lane + i); }
            for (int i = 0; i < 8; ++i) sp.x[i] = 0.37f * (lane + 1) + 0.011f * i;
            fm.c = 1.0001f; fm.d = 0.0003f; sp.pre = pre; sp.neg = neg; sp.acc = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x16 am[1][1], al[1][1];
                constexpr bool ILV = (VAR & 1) == 1;            // odd: between the MFMAs; even: the same instructions behind the GEMM
                auto filler = [&](int k) { if (k < 16) { if (VAR <= 6) fm.run(k); else if (VAR <= 8) tr.run(k); else sp.run(k); } };
                if (ILV) gemm<1, 1, 4>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh + 32 * hf, xl + 32 * hf, lane, filler);
                else {
                    gemm<1, 1>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh + 32 * hf, xl + 32 * hf, lane);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 16; ++k) filler(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) p[hf][r] = am[0][0][r] + al[0][0][r] * inv;
                __syncthreads();
            }
            p[0][0] += fm.x[0] + fm.x[11] + tr.x[0] + tr.x[11] + sp.x[0] + __builtin_bit_cast(float, sp.acc & 0x3f800000u);
        } else if (VAR >= 2) {                          // S: same-wave skew -- GEMM of one half with the finish work of the other half between its MFMAs
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x16 am[1][1], al[1][1];
                FinStage fs;
                fs.p = p[hf ^ 1]; fs.st = &st[hf ^ 1]; fs.XH = YH; fs.XL = YL; fs.PG = PG; fs.slot = wave; fs.lane = lane; fs.b = hf ^ 1;
                fs.pre = pre; fs.neg = neg; fs.inv = inv;
#pragma unroll
                for (int j = 0; j < 2; ++j) { fs.gwh[j] = wp.ld(GW + ((wave * 2 + (hf ^ 1)) * 2 + j) * 1024); fs.gwl[j] = wp.ld(GW + (32 + (wave * 2 + (hf ^ 1)) * 2 + j) * 1024); }
                constexpr int PER = VAR == 2 ? 4 : (VAR == 3 ? 6 : 8);
                gemm<1, 1, PER>(am, al, wp, wH + (uint32_t)wave * KB * 1024, wL + (uint32_t)wave * KB * 1024, xh + 32 * hf, xl + 32 * hf, lane, [&](int k) { fs.run(k); });
#pragma unroll
                for (int r = 0; r < 16; ++r) p[hf][r] = am[0][0][r] + al[0][0][r] * inv;
                __syncthreads();
            }
        } else {                                        // D: de-phased halves
#pragma unroll
            for (int slot = 0; slot < 2; ++slot) {
                if ((wave >> 2) == slot) {
                    f32x16 am[2][1], al[2][1];
                    const int mt0 = 2 * (wave & 3);
                    gemm<2, 1>(am, al, wp, wH + (uint32_t)mt0 * KB * 1024, wL + (uint32_t)mt0 * KB * 1024, xh + 32 * slot, xl + 32 * slot, lane);
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) p[b][r] = am[b][0][r] + al[b][0][r] * inv;
                } else {
                    finish<2>(p, st, wp, GW, YH, YL, PG, wave, lane, pre, neg, inv);
                }
                __syncthreads();
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int b = 0; b < 2; ++b)
        for (int r = 0; r < 16; ++r) sink += p[b][r] + st[b][r];
    out[blockIdx.x * 512 + tid] = sink;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}


// ---- schedule P (round 6): 4 waves x 512 registers, every wave 2 M-tiles x 2 N-tiles, M-tile skew -----------------------------------------
// One wave per SIMD.  A GCP2's 18 k-blocks are walked as 28 STEPS:
//     steps  0 ..  7   k-blocks 0 .. 7  (the images of the "M0 half" of the channels), both M-tiles: 12 MFMAs from 4 A loads + 4 B reads
//     steps  8 .. 17   k-blocks 8 .. 17 (the "M1 half" + the two extended-K blocks), M-tile 0 only:   6 MFMAs from 2 A loads + 4 B reads
//     steps 18 .. 27   k-blocks 8 .. 17, M-tile 1 only
// so that M-tile 0's accumulators are complete at step 17 and its finish work (merge, SiLU, gate split + 6 gate MFMAs per N-tile, residual add, state
// split, image stores) can ride between the MFMAs of steps 18 .. 27 (different weight rows: nothing is streamed twice), and M-tile 1's finish work
// between the MFMAs of steps 0 .. 7 of the NEXT GCP2, which contract over the M0 half of the channels only.  Barriers: in front of step 0 (every wave's
// M0 images are written) and in front of step 8 (M1 images).
//   PV = 0   (i)   lockstep: all 18 k-blocks on 2 x 2 tiles, barrier, the whole finish, barrier            -- prices 12 MFMAs per 4 + 4 operand requests
//                        and one wave per SIMD in the exposed VALU phase
//   PV = 1   (ii)  the step order above, M-tile 0's finish between the MFMAs of steps 18 .. 27, M-tile 1's finish exposed behind step 27
//   PV = 2   (iii) M-tile 1's finish between the MFMAs of steps 0 .. 7 of the next GCP2 as well: only the two barriers stay exposed
//   PV = 3   the step order alone (no finish work at all): the floor of the GEMM in this order
//   PV = 4   the step order, both finishes exposed (behind step 17 and behind step 27): what the skewed ORDER costs without any overlap
// PER = instructions of the hosted work issued behind each MFMA.
struct FinP {                          // finish work of ONE M-tile x both N-tiles, cut into granules
    f32x16 act[2];
    f32x16* st;                        // [2]
    f32x16* gm; f32x16* gl;            // [2] gate accumulators of the wave (per N-tile), carried from M-tile 0's finish to M-tile 1's
    h8 gwh[2], gwl[2], bh, bl;
    char *XH, *XL; float* PG;
    int slot, lane, mrow;              // mrow: which 4 image groups this M-tile owns
    float pre, neg, inv;
    bool first;                        // first M-tile of the wave: the gate accumulators start from zero
    bool merged = false;               // the two accumulators were merged already (am holds am + al * inv)
    __device__ __forceinline__ void silu4(const f32x16 (&am)[2], const f32x16 (&al)[2], int b, int i) {
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) { const float x = merged ? am[b][r] : am[b][r] + al[b][r] * inv; act[b][r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }
    }
    __device__ __forceinline__ void gsplit(int b, int j, int s0) {
#pragma unroll
        for (int s = s0; s < s0 + 4; s += 2) {
            h2 hi, lo;
            split16x2(act[b][8 * j + s], act[b][8 * j + s + 1], hi, lo, pre, neg);
            bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
        }
    }
    __device__ __forceinline__ void gmfma(int b, int j) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        asm("s_nop 1" : "+v"(bh), "+v"(bl));
        const bool z = first && j == 0;
        gm[b] = MFMA16(gwh[j], bh, z ? zero : gm[b]);
        gl[b] = MFMA16(gwh[j], bl, z ? zero : gl[b]);
        gl[b] = MFMA16(gwl[j], bh, gl[b]);
    }
    __device__ __forceinline__ void gout(int b, int t) {
        const int half = lane >> 5, l31 = lane & 31;
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gm[b][4 * t + i] + gl[b][4 * t + i] * inv;
        *(v4f*)(PG + (((slot & 1) * 64 + 32 * b + l31) * 32 + 4 * ((2 * t + half) ^ (l31 & 7)))) = v;      // (two of the four slots: the harness' LDS also holds a second image buffer)
    }
    __device__ __forceinline__ void image(int b, int q) {
        const int half = lane >> 5, l31 = lane & 31;
        h4 vh, vl;
#pragma unroll
        for (int t = 0; t < 4; ++t) st[b][4 * q + t] += act[b][4 * q + t];
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            h2 hi, lo;
            split16x2(st[b][4 * q + t], st[b][4 * q + t + 1], hi, lo, pre, neg);
            vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
        }
        const int off = ((4 * mrow + q) * TP + 32 * b + l31) * 16 + 8 * half;
        *(h4*)(XH + off) = vh;
        *(h4*)(XL + off) = vl;
    }
    // granule g of NG (NG = 28: first M-tile; 36: last M-tile, the gate partial goes out)
    template <int G>
    __device__ __forceinline__ void granule(const f32x16 (&am)[2], const f32x16 (&al)[2]) {
        if constexpr (G < 8) silu4(am, al, G >> 2, G & 3);
        else if constexpr (G < 20) {
            constexpr int u = G - 8, bj = u / 3, w = u % 3;
            if constexpr (w == 0) gsplit(bj >> 1, bj & 1, 0);
            else if constexpr (w == 1) gsplit(bj >> 1, bj & 1, 4);
            else gmfma(bj >> 1, bj & 1);
        } else if constexpr (G < 28) image((G - 20) >> 2, (G - 20) & 3);
        else if constexpr (G < 36) gout((G - 28) >> 2, (G - 28) & 3);
    }
    // stage S of NS: granules [S * NG / NS, (S + 1) * NG / NS)
    template <int S, int NS, int NG>
    __device__ __forceinline__ void stage(const f32x16 (&am)[2], const f32x16 (&al)[2]) {
        constexpr int g0 = S * NG / NS, g1 = (S + 1) * NG / NS;
        run_granules<g0, g1>(am, al);
    }
    template <int G0, int G1>
    __device__ __forceinline__ void run_granules(const f32x16 (&am)[2], const f32x16 (&al)[2]) {
        if constexpr (G0 < G1) { granule<G0>(am, al); run_granules<G0 + 1, G1>(am, al); }
    }
};

template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}

constexpr int P_STEPS = 28;
__host__ __device__ constexpr int p_kblk(int s) { return s < 18 ? s : s - 10; }
__host__ __device__ constexpr int p_mmask(int s) { return s < 8 ? 3 : (s < 18 ? 1 : 2); }

template <int PV, int PER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void kp(int n, float pre, float neg, float inv, const h8* __restrict__ W, float* out,
                                                                                     unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;
    char* XL = XH + 36 * TP * 16;
    char* YH = XL + 36 * TP * 16;                      // images written by the finish work: [32][65] x 16 B (a second buffer: the harness models the barriers, not the data flow)
    char* YL = YH + 32 * TP * 16;
    float* PG = (float*)(YL + 32 * TP * 16);           // gate partials [4][64][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 36 * TP; i += 256) {
        h8 v;
        for (int s = 0; s < 8; ++s) v[s] = (_Float16)(0.001f * ((i + s) & 63));
        ((h8*)XH)[i] = v;
    }
    __syncthreads();
    WPool wp;
    wp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(W), 0, (2 * 8 * (KB + 4) + 64) * 1024, 0x00020000);
    wp.voff = (uint32_t)lane * 16u;
    const uint32_t wH = (uint32_t)(2 * wave) * KB * 1024, wL = 8 * (KB + 4) * 1024 + (uint32_t)(2 * wave) * KB * 1024;     // this wave's two M-tiles
    const h8* xh = (const h8*)XH;
    const h8* xl = (const h8*)XL;
    const uint32_t GW = 2 * 8 * (KB + 4) * 1024;
    const int boff = (lane >> 5) * TP + (lane & 31);
    f32x16 st[2][2], am[2][2], al[2][2], gm[2], gl[2];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < 2; ++m)
        for (int b = 0; b < 2; ++b) {
            am[m][b] = zero; al[m][b] = zero;
            for (int r = 0; r < 16; ++r) st[m][b][r] = 0.1f * r;
        }
    gm[0] = gm[1] = gl[0] = gl[1] = zero;
    FinP f0, f1;                                         // finish of M-tile 0 / M-tile 1
    f0.act[0] = f0.act[1] = f1.act[0] = f1.act[1] = zero;
    f0.st = st[0]; f1.st = st[1];
    f0.gm = f1.gm = gm; f0.gl = f1.gl = gl;
    f0.XH = f1.XH = YH; f0.XL = f1.XL = YL; f0.PG = f1.PG = PG;
    f0.slot = f1.slot = wave; f0.lane = f1.lane = lane; f0.mrow = wave; f1.mrow = 4 + wave;
    f0.pre = f1.pre = pre; f0.neg = f1.neg = neg; f0.inv = f1.inv = inv;
    f0.first = true; f1.first = false;
    f32x16 p1m[2];                                       // M-tile 1's finished pre-activations (PV = 2): merged behind step 27, out of the way of the next GCP2's first steps
    p1m[0] = p1m[1] = zero;
    if (PV == 2) f1.merged = true;
    auto gate_w = [&](FinP& f, int m) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { f.gwh[j] = wp.ld(GW + ((2 * wave + m) * 2 + j) * 1024); f.gwl[j] = wp.ld(GW + (32 + (2 * wave + m) * 2 + j) * 1024); }
    };
    constexpr int R = PD + 1;
    h8 ah[R][2], alo[R][2], bh[2][2], bl[2][2];
    auto a_loads = [&](auto sc) {                        // A operands of step S into ring slot S % R
        constexpr int S = decltype(sc)::value % P_STEPS, slot = decltype(sc)::value % R, k = p_kblk(S), mm = p_mmask(S);
        // (a one-M-tile step uses ring entry [slot][0] whichever M-tile it is)
        if constexpr (mm == 3) {
            alo[slot][1] = wp.ld(wL + (KB + k) * 1024); alo[slot][0] = wp.ld(wL + k * 1024);
            ah[slot][1] = wp.ld(wH + (KB + k) * 1024); ah[slot][0] = wp.ld(wH + k * 1024);
        } else {
            constexpr int m = mm == 1 ? 0 : 1;
            alo[slot][0] = wp.ld(wL + (m * KB + k) * 1024); ah[slot][0] = wp.ld(wH + (m * KB + k) * 1024);
        }
    };
    auto b_reads = [&](auto sc) {                        // B operands of step S into buffer S & 1
        constexpr int S = decltype(sc)::value % P_STEPS, buf = decltype(sc)::value & 1, k = p_kblk(S);
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) { bh[buf][nn] = xh[boff + k * 2 * TP + 32 * nn]; bl[buf][nn] = xl[boff + k * 2 * TP + 32 * nn]; }
    };
    auto mfmas = [&](auto sc) {
        constexpr int S = decltype(sc)::value % P_STEPS, slot = decltype(sc)::value % R, buf = S & 1, k = p_kblk(S), mm = p_mmask(S);
        if constexpr (mm == 3) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) am[m][nn] = MFMA16(ah[slot][m], bh[buf][nn], k == 0 ? zero : am[m][nn]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) al[m][nn] = MFMA16(ah[slot][m], bl[buf][nn], k == 0 ? zero : al[m][nn]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) al[m][nn] = MFMA16(alo[slot][m], bh[buf][nn], al[m][nn]);
        } else {
            constexpr int m = mm == 1 ? 0 : 1;
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) am[m][nn] = MFMA16(ah[slot][0], bh[buf][nn], am[m][nn]);
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) al[m][nn] = MFMA16(ah[slot][0], bl[buf][nn], al[m][nn]);
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) al[m][nn] = MFMA16(alo[slot][0], bh[buf][nn], al[m][nn]);
        }
    };
    // prologue: A operands of steps 0, 1 (P_STEPS % R == 1: the ring slot of step S is S % R within an iteration; the wrap-around requests of steps 26, 27
    // target the slots of the NEXT iteration's steps 0, 1, i.e. (28 + s) % R -- handled by passing 28, 29 to a_loads)
    static_assert(P_STEPS % R == 1, "ring slot arithmetic of the wrap-around");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (PV == 0) {
        for (int it = 0; it < n; ++it) {
            f32x16 am2[2][2], al2[2][2];
            gemm<2, 2>(am2, al2, wp, wH, wL, xh, xl, lane);
            __syncthreads();
            gate_w(f0, 0); gate_w(f1, 1);
            f0.template run_granules<0, 28>(am2[0], al2[0]);
            f1.template run_granules<0, 36>(am2[1], al2[1]);
            __syncthreads();
        }
    } else {
        // ring slots: step s of an iteration uses slot (s + phase) % R with phase advancing by P_STEPS % R = 1 per iteration; to keep every index static
        // the loop body is unrolled over R iterations
        a_loads(std::integral_constant<int, 0>{}); a_loads(std::integral_constant<int, 1>{});
        for (int it = 0; it < n; it += R) {
            static_for<0, R>([&](auto uc) {
                constexpr int U = decltype(uc)::value;                // ring phase of this iteration
                __syncthreads();                                      // every wave's M0 images are written
                b_reads(std::integral_constant<int, 0 + 2 * 0>{});    // (buffer parity restarts at 0 with every iteration: P_STEPS is even)
                if constexpr (PV == 2) gate_w(f1, 1);
                static_for<0, P_STEPS>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    // ring slot of step S in this iteration: (S + U) % R  -> a_loads / mfmas take "virtual step" numbers congruent to that
                    constexpr int V = S + U * P_STEPS;                 // V % R == (S + U) % R since P_STEPS % R == 1; V & 1 == S & 1; V % P_STEPS == S
                    if constexpr (S == 8) {
                        __syncthreads();                              // every wave's M1 images (and the extended-K rows) are written
                        b_reads(std::integral_constant<int, 8>{});
                    }
                    if constexpr (S == 18 && (PV == 1 || PV == 2)) gate_w(f0, 0);
                    if constexpr (S == 18 && PV == 4) {               // both finishes exposed: M-tile 0's behind step 17
                        gate_w(f0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        f0.template run_granules<0, 28>(am[0], al[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        constexpr int slot = V % R, buf = S & 1, k = p_kblk(S), mm = p_mmask(S);
                        __builtin_amdgcn_s_waitcnt((15 << 8) | (7 << 4) | ((p_mmask((S + 1) % P_STEPS) == 3 ? 4 : 2) & 15));    // this step's A operands have landed (the next step's may be in flight)
                        __builtin_amdgcn_sched_barrier(0);
                        (void)slot; (void)buf; (void)k; (void)mm;
                        // MFMAs of this step (ring slot V % R)
                        mfmas(std::integral_constant<int, V>{});
                        if constexpr (S + 1 != 8 && S + 1 != P_STEPS) b_reads(std::integral_constant<int, S + 1>{});
                        a_loads(std::integral_constant<int, V + PD>{});
                        constexpr int NM = mm == 3 ? 12 : 6;
                        constexpr bool HOST0 = (PV == 1 || PV == 2) && S >= 18;
                        constexpr bool HOST1 = PV == 2 && S < 8;
                        if constexpr (HOST0) f0.template stage<S - 18, 10, 28>(am[0], al[0]);
                        if constexpr (HOST1) f1.template stage<S, 8, 36>(p1m, p1m);
                        if constexpr (HOST0 || HOST1) {
#pragma unroll
                            for (int i = 0; i < NM; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x002 | 0x400 | 0x200 | 0x100 | 0x020, PER, 0);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < NM; ++i) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x100 | 0x020, 1, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
                // behind step 27: M-tile 1's accumulators are complete
                if constexpr (PV == 3) { for (int m = 0; m < 2; ++m) for (int b = 0; b < 2; ++b) { st[m][b][0] += am[m][b][0]; st[m][b][1] += al[m][b][0]; } }
                if constexpr (PV == 2) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) p1m[b][r] = am[1][b][r] + al[1][b][r] * inv;      // (the merge of the finish work, moved here: 32 FMAs)
                } else if constexpr (PV == 1 || PV == 4) {
                    gate_w(f1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    f1.template run_granules<0, 36>(am[1], al[1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) sink += st[m][b][r] + am[m][b][r] + al[m][b][r] + p1m[b][r] + gm[b][r] + gl[b][r] + f0.act[b][r] + f1.act[b][r];
    out[blockIdx.x * 256 + tid] = sink;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PV, int PER>
void run_p(const char* name, int blocks) {
    const int n = 300;
    float* out; unsigned long long* ticks; h8* W;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    (void)hipMemset(ticks, 0, 8 * 8 * blocks);
    const size_t wbytes = (size_t)(2 * 8 * (KB + 4) + 64) * 64 * 16;
    (void)hipMalloc(&W, wbytes); (void)hipMemset(W, 0x11, wbytes);
    const size_t lds = (2 * 36 + 2 * 32) * TP * 16 + 2 * 64 * 32 * 4;
    (void)hipFuncSetAttribute((const void*)kp<PV, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kp<PV, PER>), dim3(blocks), dim3(256), lds, 0, 6, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed (LDS %zu B)\n", name, lds); return; }
    hipLaunchKernelGGL((kp<PV, PER>), dim3(blocks), dim3(256), lds, 0, n, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h(8 * blocks);
    (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < 4; ++w) m = m > h[i * 8 + w] ? m : h[i * 8 + w]; mx += m; }
    mx /= blocks;
    printf("%-64s %4d workgroups  %8.0f clk per GCP2   (MFMA floor per SIMD %d + gate %d)\n", name, blocks, mx / n, 216 * 32, 24 * 32);
    (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(W);
}


// ---- schedule Q (round 6): 4 waves x 512 registers, QUARTER-sequential with the weights of the current M-tile held in registers ----------------
// A wave owns 2 M-tiles x 2 N-tiles as in P, but walks them one 32 x 32 output tile ("quarter") at a time, M-major:
//     Q(0,0)  Q(0,1)  Q(1,0)  Q(1,1)        18 k-blocks x 3 MFMAs each
// The A operands (weights) of M-tile m are streamed into an 18-block register HOLD while Q(m,0) runs and are used again by Q(m,1) -- 144 registers, which
// a 512-register wave has -- so every weight byte is still streamed once per 64 edges; the hold entry of k-block k is re-requested (next M-tile) right behind
// its last use, 17 k-blocks ahead of the next one.  Every quarter hosts the finish work of the quarter before it (one accumulator tile: merge, SiLU, gate
// split + 6 gate MFMAs, residual add, state split, image stores): ~1 granule of ~20 instructions per k-block, uniformly -- no phase without hosted work and no
// finish work without MFMAs to hide under.  Data flow of the real layer chain: Q(m,n) of GCP2 k+1 contracts over the images of N-tile n only, i.e. needs the
// finish of Q(0,n) and Q(1,n) of GCP2 k of every wave; those are hosted by Q(0,n+1) / Q(1,n+1), so a barrier in front of Q(0,0) and one in front of Q(0,1)
// (two per GCP2, as today) are what the harness models.
//   QV = 0   the quarter order alone, no finish work (GEMM floor of this order: one accumulator chain pair per wave)
//   QV = 1   finish work of every quarter exposed behind it
//   QV = 2   finish work of every quarter between the MFMAs of the next one, PER instructions behind each MFMA
struct FinQ {                          // finish work of ONE accumulator tile, cut into granules
    f32x16 act;
    f32x16* gm; f32x16* gl;            // gate accumulators of the N-tile (carried from M-tile 0's quarter to M-tile 1's)
    h8 gwh[2], gwl[2], bh, bl;
    char *XH, *XL; float* PG;
    int slot, lane;
    float pre, neg, inv;
    __device__ __forceinline__ void silu4(const f32x16& am, const f32x16& al, int i) {
#pragma unroll
        for (int r = 4 * i; r < 4 * i + 4; ++r) { const float x = am[r] + al[r] * inv; act[r] = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }
    }
    __device__ __forceinline__ void gsplit(int j, int s0) {
#pragma unroll
        for (int s = s0; s < s0 + 4; s += 2) {
            h2 hi, lo;
            split16x2(act[8 * j + s], act[8 * j + s + 1], hi, lo, pre, neg);
            bh[s] = hi[0]; bh[s + 1] = hi[1]; bl[s] = lo[0]; bl[s + 1] = lo[1];
        }
    }
    template <bool FIRST>
    __device__ __forceinline__ void gmfma(int j) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        asm("s_nop 1" : "+v"(bh), "+v"(bl));
        *gm = MFMA16(gwh[j], bh, (FIRST && j == 0) ? zero : *gm);
        *gl = MFMA16(gwh[j], bl, (FIRST && j == 0) ? zero : *gl);
        *gl = MFMA16(gwl[j], bh, *gl);
    }
    __device__ __forceinline__ void gout(int b, int t) {
        const int half = lane >> 5, l31 = lane & 31;
        v4f v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (*gm)[4 * t + i] + (*gl)[4 * t + i] * inv;
        *(v4f*)(PG + (((slot & 1) * 64 + 32 * b + l31) * 32 + 4 * ((2 * t + half) ^ (l31 & 7)))) = v;
    }
    __device__ __forceinline__ void image(f32x16& st, int mrow, int b, int q) {
        const int half = lane >> 5, l31 = lane & 31;
        h4 vh, vl;
#pragma unroll
        for (int t = 0; t < 4; ++t) st[4 * q + t] += act[4 * q + t];
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            h2 hi, lo;
            split16x2(st[4 * q + t], st[4 * q + t + 1], hi, lo, pre, neg);
            vh[t] = hi[0]; vh[t + 1] = hi[1]; vl[t] = lo[0]; vl[t + 1] = lo[1];
        }
        const int off = ((4 * mrow + q) * TP + 32 * b + l31) * 16 + 8 * half;
        *(h4*)(XH + off) = vh;
        *(h4*)(XL + off) = vl;
    }
    // granule G of the finish of quarter (M, B): 0..3 SiLU, 4..9 gate (split, split, 3 MFMAs) x 2, 10..13 images, 14..17 gate partial out (M-tile 1 only)
    template <int G, int M>
    __device__ __forceinline__ void granule(const f32x16& am, const f32x16& al, f32x16& st, int mrow, int b) {
        if constexpr (G < 4) silu4(am, al, G);
        else if constexpr (G < 10) {
            constexpr int u = G - 4, j = u / 3, w = u % 3;
            if constexpr (w == 0) gsplit(j, 0);
            else if constexpr (w == 1) gsplit(j, 4);
            else gmfma<M == 0>(j);
        } else if constexpr (G < 14) image(st, mrow, b, G - 10);
        else if constexpr (G < 18 && M == 1) gout(b, G - 14);
    }
    template <int G0, int G1, int M>
    __device__ __forceinline__ void run_granules(const f32x16& am, const f32x16& al, f32x16& st, int mrow, int b) {
        if constexpr (G0 < G1) { granule<G0, M>(am, al, st, mrow, b); run_granules<G0 + 1, G1, M>(am, al, st, mrow, b); }
    }
};

template <int QV, int PER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void kq(int n, float pre, float neg, float inv, const h8* __restrict__ W, float* out,
                                                                                     unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* XH = smem;
    char* XL = XH + 36 * TP * 16;
    char* YH = XL + 36 * TP * 16;
    char* YL = YH + 32 * TP * 16;
    float* PG = (float*)(YL + 32 * TP * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 36 * TP; i += 256) {
        h8 v;
        for (int s = 0; s < 8; ++s) v[s] = (_Float16)(0.001f * ((i + s) & 63));
        ((h8*)XH)[i] = v;
    }
    __syncthreads();
    WPool wp;
    wp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(W), 0, (2 * 8 * (KB + 4) + 64) * 1024, 0x00020000);
    wp.voff = (uint32_t)lane * 16u;
    const uint32_t wH = (uint32_t)(2 * wave) * KB * 1024, wL = 8 * (KB + 4) * 1024 + (uint32_t)(2 * wave) * KB * 1024;
    const h8* xh = (const h8*)XH;
    const h8* xl = (const h8*)XL;
    const uint32_t GW = 2 * 8 * (KB + 4) * 1024;
    const int boff = (lane >> 5) * TP + (lane & 31);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 st[2][2], am[2], al[2], gm[2], gl[2];          // am / al: two accumulator sets, quarter q accumulates into set q & 1 while the finish of quarter q - 1 reads the other
    for (int m = 0; m < 2; ++m)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) st[m][b][r] = 0.1f * r;
    am[0] = am[1] = al[0] = al[1] = gm[0] = gm[1] = gl[0] = gl[1] = zero;
    FinQ fq;
    fq.act = zero;
    fq.XH = YH; fq.XL = YL; fq.PG = PG; fq.slot = wave; fq.lane = lane; fq.pre = pre; fq.neg = neg; fq.inv = inv;
    h8 hh[KB], hl[KB];                                    // the HOLD: A operands (hi, lo') of the current M-tile, all 18 k-blocks
    h8 bh[3], bl[3];                                      // B operands, two k-blocks ahead
    auto a_load = [&](auto mc, auto kc) {
        constexpr int m = decltype(mc)::value, k = decltype(kc)::value;
        hl[k] = wp.ld(wL + (m * KB + k) * 1024); hh[k] = wp.ld(wH + (m * KB + k) * 1024);
    };
    auto gate_w = [&](int m) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { fq.gwh[j] = wp.ld(GW + ((2 * wave + m) * 2 + j) * 1024); fq.gwl[j] = wp.ld(GW + (32 + (2 * wave + m) * 2 + j) * 1024); }
    };
    auto b_read = [&](auto bc, int nn, int k) {
        constexpr int buf = decltype(bc)::value;
        bh[buf] = xh[boff + k * 2 * TP + 32 * nn]; bl[buf] = xl[boff + k * 2 * TP + 32 * nn];
    };
    // prologue: M-tile 0's weights
    static_for<0, KB>([&](auto kc) { a_load(std::integral_constant<int, 0>{}, kc); });
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
        static_for<0, 4>([&](auto qc) {
            constexpr int Q = decltype(qc)::value, M = Q >> 1, B = Q & 1, SET = Q & 1;
            constexpr int PQ = (Q + 3) & 3, PM = PQ >> 1, PB = PQ & 1, PSET = PQ & 1;      // the quarter whose finish this one hosts
            if constexpr (B == 0 || true) {
                if constexpr (Q == 0 || Q == 1) __syncthreads();     // images of N-tile B complete (every wave's finish of Q(0,B), Q(1,B) of the previous GCP2)
            }
            b_read(std::integral_constant<int, 0>{}, B, 0);
            b_read(std::integral_constant<int, 1>{}, B, 1);
            if constexpr (QV >= 1) { fq.gm = &gm[PB]; fq.gl = &gl[PB]; }
            if constexpr (QV == 2) gate_w(PM);
            static_for<0, KB>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k + 2 < KB) b_read(std::integral_constant<int, (k + 2) % 3>{}, B, k + 2);
                // vmcnt: in a B == 0 quarter the hold entries were requested >= 17 blocks ago except right after the prologue; lgkmcnt: the two younger B reads may be in flight
                __builtin_amdgcn_sched_barrier(0);
                am[SET] = MFMA16(hh[k], bh[k % 3], k == 0 ? zero : am[SET]);
                al[SET] = MFMA16(hh[k], bl[k % 3], k == 0 ? zero : al[SET]);
                al[SET] = MFMA16(hl[k], bh[k % 3], al[SET]);
                if constexpr (B == 1) a_load(std::integral_constant<int, M ^ 1>{}, kc);       // last use of hold[k] for this M-tile: request the next M-tile's block
                if constexpr (QV == 2) {
                    constexpr int NG = PM == 1 ? 18 : 14, g0 = k * NG / KB, g1 = (k + 1) * NG / KB;
                    fq.template run_granules<g0, g1, PM>(am[PSET], al[PSET], st[PM][PB], 4 * PM + wave, PB);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400 | 0x200 | 0x100 | 0x020, PER, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (QV == 1) {                      // exposed: this quarter's own finish right behind it
                fq.gm = &gm[B]; fq.gl = &gl[B];
                gate_w(M);
                __builtin_amdgcn_sched_barrier(0);
                fq.template run_granules<0, (M == 1 ? 18 : 14), M>(am[SET], al[SET], st[M][B], 4 * M + wave, B);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (QV == 0) { st[M][B][0] += am[SET][0]; st[M][B][1] += al[SET][0]; }
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int b = 0; b < 2; ++b)
            for (int r = 0; r < 16; ++r) sink += st[m][b][r] + am[b][r] + al[b][r] + gm[b][r] + gl[b][r] + fq.act[r];
    out[blockIdx.x * 256 + tid] = sink;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int QV, int PER>
void run_q(const char* name, int blocks) {
    const int n = 300;
    float* out; unsigned long long* ticks; h8* W;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    (void)hipMemset(ticks, 0, 8 * 8 * blocks);
    const size_t wbytes = (size_t)(2 * 8 * (KB + 4) + 64) * 64 * 16;
    (void)hipMalloc(&W, wbytes); (void)hipMemset(W, 0x11, wbytes);
    const size_t lds = (2 * 36 + 2 * 32) * TP * 16 + 2 * 64 * 32 * 4;
    (void)hipFuncSetAttribute((const void*)kq<QV, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kq<QV, PER>), dim3(blocks), dim3(256), lds, 0, 6, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed (LDS %zu B)\n", name, lds); return; }
    hipLaunchKernelGGL((kq<QV, PER>), dim3(blocks), dim3(256), lds, 0, n, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h(8 * blocks);
    (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < 4; ++w) m = m > h[i * 8 + w] ? m : h[i * 8 + w]; mx += m; }
    mx /= blocks;
    printf("%-64s %4d workgroups  %8.0f clk per GCP2   (MFMA floor per SIMD %d + gate %d)\n", name, blocks, mx / n, 216 * 32, 24 * 32);
    (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(W);
}

template <int VAR>
void run(const char* name, int blocks) {
    const int n = 300;
    float* out; unsigned long long* ticks; h8* W;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    const size_t wbytes = (size_t)(3 * 8 * (KB + 4) + 96) * 64 * 16;
    (void)hipMalloc(&W, wbytes); (void)hipMemset(W, 0x11, wbytes);       // every f16 = 0x1111 = 1.3e-4: finite data, the state stays bounded
    const size_t lds = (2 * 36 + 2 * 16) * TP * 16 + 4 * 64 * 32 * 4;
    (void)hipFuncSetAttribute((const void*)kb<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((kb<VAR>), dim3(blocks), dim3(512), lds, 0, 5, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%s: launch failed (LDS %zu B)\n", name, lds); return; }
    hipLaunchKernelGGL((kb<VAR>), dim3(blocks), dim3(512), lds, 0, n, 4.8828125e-4f, -2048.f, 4.8828125e-4f, W, out, ticks);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<unsigned long long> h(8 * blocks);
    (void)hipMemcpy(h.data(), ticks, 8 * 8 * blocks, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int i = 0; i < blocks; ++i) { double m = 0; for (int w = 0; w < 8; ++w) m = m > h[i * 8 + w] ? m : h[i * 8 + w]; mx += m; }
    mx /= blocks;
    printf("%-40s %4d workgroups  %8.0f clk per GCP2   (MFMA floor per SIMD %d + gate %d)\n", name, blocks, mx / n, 216 * 32, 24 * 32);
    (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(W);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'Z') {       // round 6: one accumulator per N-tile (third weight image), lockstep
        for (int rep = 0; rep < 3; ++rep) {
            run<0>("L lockstep (8 waves GEMM, then finish)", 256);
            run<20>("Z lockstep, ONE accumulator (3 weight images, no merge)", 256);
        }
        run<0>("L lockstep, one workgroup", 1);
        run<20>("Z, one workgroup", 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'M') {       // round 6: M-tile de-phasing of the two waves of a SIMD (8 waves, the kernel's own tile per wave)
        for (int rep = 0; rep < 2; ++rep) {
            run<0>("L lockstep (8 waves GEMM, then finish)", 256);
            run<11>("M de-phased by M-tile (waves 4-7 half a GEMM behind)", 256);
            run<12>("M + s_setprio 1 in the VALU slots", 256);
        }
        run<11>("M, one workgroup", 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'Q') {       // round 6: schedule Q (quarter-sequential, weights held in registers)
        run<0>("L lockstep (8 waves GEMM, then finish)", 256);
        run_q<0, 0>("Q      quarter order alone, no finish work (GEMM floor)", 256);
        run_q<1, 0>("Q      every quarter's finish exposed behind it", 256);
        run_q<2, 4>("Q      finish under the next quarter's MFMAs, 1 MFMA : 4", 256);
        run_q<2, 6>("Q      finish under the next quarter's MFMAs, 1 MFMA : 6", 256);
        run_q<2, 8>("Q      finish under the next quarter's MFMAs, 1 MFMA : 8", 256);
        run_q<2, 10>("Q      finish under the next quarter's MFMAs, 1 MFMA : 10", 256);
        run_q<2, 6>("Q      1 MFMA : 6, one workgroup", 1);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'P') {       // round 6: schedule P only (with the L baseline)
        run<0>("L lockstep (8 waves GEMM, then finish)", 256);
        run_p<0, 0>("P(i)   4 waves x (2M x 2N), lockstep GEMM | finish", 256);
        run_p<3, 0>("P      step order alone, no finish work (GEMM floor)", 256);
        run_p<4, 0>("P      step order, both finishes exposed", 256);
        run_p<1, 4>("P(ii)  M-tile 0's finish under steps 18..27, 1 MFMA : 4", 256);
        run_p<1, 6>("P(ii)  M-tile 0's finish under steps 18..27, 1 MFMA : 6", 256);
        run_p<1, 8>("P(ii)  M-tile 0's finish under steps 18..27, 1 MFMA : 8", 256);
        run_p<1, 12>("P(ii)  M-tile 0's finish under steps 18..27, 1 MFMA : 12", 256);
        run_p<2, 4>("P(iii) + M-tile 1's finish under steps 0..7 of the next, 1 : 4", 256);
        run_p<2, 6>("P(iii) + M-tile 1's finish under steps 0..7 of the next, 1 : 6", 256);
        run_p<2, 8>("P(iii) + M-tile 1's finish under steps 0..7 of the next, 1 : 8", 256);
        run_p<2, 12>("P(iii) + M-tile 1's finish under steps 0..7 of the next, 1 : 12", 256);
        run<0>("L lockstep (8 waves GEMM, then finish)", 256);
        run_p<2, 6>("P(iii), one workgroup", 1);
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("L lockstep (8 waves GEMM, then finish)", 256);
        run<1>("D de-phased halves (4 GEMM | 4 finish)", 256);
    }
    run<2>("S same-wave skew, 1 MFMA : 4", 256);
    run<3>("S same-wave skew, 1 MFMA : 6", 256);
    run<4>("S same-wave skew, 1 MFMA : 8", 256);
    run<6>("F fma fillers (192/slot) behind the GEMM", 256);
    run<5>("F fma fillers between the MFMAs", 256);
    run<8>("F exp/rcp fillers (192/slot) behind the GEMM", 256);
    run<7>("F exp/rcp fillers between the MFMAs", 256);
    run<10>("F split fillers (192/slot) behind the GEMM", 256);
    run<9>("F split fillers between the MFMAs", 256);
    run<0>("L lockstep, one workgroup", 1);
    run<1>("D de-phased, one workgroup", 1);
    return 0;
}
