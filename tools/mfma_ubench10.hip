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
template <int MT, int NT, int PER = 0, class Hook = void (*)(int)>
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
    wp.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h8*>(W), 0, (2 * 8 * (KB + 4) + 64) * 1024, 0x00020000);
    wp.voff = (uint32_t)lane * 16u;
    const uint32_t wH = 0, wL = 8 * (KB + 4) * 1024;     // [8 M-tiles][KB][64] hi, then lo'
    const h8* xh = (const h8*)XH;
    const h8* xl = (const h8*)XL;
    const uint32_t GW = 2 * 8 * (KB + 4) * 1024;
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

template <int VAR>
void run(const char* name, int blocks) {
    const int n = 300;
    float* out; unsigned long long* ticks; h8* W;
    (void)hipMalloc(&out, 4 * 512 * blocks); (void)hipMalloc(&ticks, 8 * 8 * blocks);
    const size_t wbytes = (size_t)(2 * 8 * (KB + 4) + 64) * 64 * 16;
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

int main() {
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
