// gcdm_edge_x3.hip.h -- split-precision variant of the fused edge-message kernel (gfx950).
//
// Same tile, phases and LDS budget as k_edge_msg (gcdm_kernels.hip.h), but the dense contractions run on
// v_mfma_f32_32x32x16_f16 with every fp32 operand x represented as  x = hi + 2^-11 * lo',  hi = f16(x),
// lo' = f16((x - hi) * 2^11):
//     W.X  =  Whi.Xhi  +  2^-11 (Whi.Xlo' + Wlo'.Xhi)        [dropped: 2^-22 Wlo'.Xlo' ~ 2^-24 relative]
// Products of two f16 values are exact in fp32 and the MFMA accumulates in fp32, so the result carries fp32-class error
// (measured against an fp64 run of the reference: 7e-7 vs 8e-7 for plain fp32, tests/test_oracle_golden.py::test_f16x3_emulation)
// at 3 f16 MFMAs per 16-deep block instead of 8 fp32 MFMAs: 5.3x the matrix rate.
//   * weights are split once on the host (two packed f16 arrays, same bytes as fp32);
//   * the message scalars live as fp32 in REGISTERS of the wave that owns their 64 channels (residual adds are exact fp32);
//     LDS holds only their hi / lo' images in 8-channel groups (XH8 / XL8: 16 B per group, same footprint as fp32);
//   * f16 range: images hold x * 2^-11 (the packed weights carry the 2^11), so |x| up to 1.2e8 is representable; beyond that
//     GCDM_FLAG_F16_RANGE is raised and the caller re-runs in fp32 mode.
//   * round 3: the kernel is PERSISTENT -- one workgroup per CU walks its XCD's contiguous range of tiles, the next tile's index words,
//     per-edge constants and gathered node rows are requested while the current tile still computes (k_edge_msg_x3 below); the attention
//     logits are formed on the state registers and the segment sums read an image that is already attention-weighted.
#pragma once
#include "gcdm_kernels.hip.h"
#include <type_traits>

// SiLU in the scaled units of this kernel: the message scalars are kept as c * m.s with c = -log2(e) (the host folds c into the weights, gcdm_api.hip
// X3_C), so for x' = c * x:  c * SiLU(x) = x' / (1 + exp2(x'))  -- exp2, add, rcp, mul: one multiply less than x * sigmoid(x)
#define X3_C (-1.4426950408889634f)
// Timing ablations (-DGCDM_ABL_<piece> removes one piece of the kernel: wrong results by construction; tools/ab_run.sh reads the in-kernel end-of-tile
// stamp of such builds, profiles/r03_ablation_edge_kernel.md is the record).  They exist only in a build that ALSO says -DGCDM_ABLATIONS.
#ifndef GCDM_ABLATIONS
#undef GCDM_ABL_NOSILU
#undef GCDM_ABL_NOWLOAD
#undef GCDM_ABL_NOB
#undef GCDM_ABL_MFMA1
#undef GCDM_ABL_NOGATHER
#undef GCDM_ABL_NOCONST
#undef GCDM_ABL_FINW
#undef GCDM_ABL_NOP1
#undef GCDM_ABL_NOBETA
#undef GCDM_ABL_NOP1W
#undef GCDM_ABL_NOPQ
#undef GCDM_ABL_NOGATE
#undef GCDM_ABL_GATE_NOPG
#undef GCDM_ABL_NOSTORE
#undef GCDM_ABL_NOAGG
#undef GCDM_ABL_VECFMA
#undef GCDM_ABL_VECNONE
#endif
#ifdef GCDM_ABL_NOSILU
__device__ __forceinline__ float silu_scaled(float xs) { return xs; }
#else
__device__ __forceinline__ float silu_scaled(float xs) { return xs * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs)); }
#endif

// ---- packed fp32 (round 5: measured, NOT the default) -------------------------------------------------------------------------------
// v_pk_{mul,add,fma}_f32 process an aligned register PAIR per instruction.  Every element-wise step of the VALU phases whose operands sit in
// adjacent accumulator registers (channels 2i, 2i+1 of one edge) is written below on float2 values through pk_*<PK>: the accumulator merge, SiLU's
// denominator and product, the prescale of every hi / lo' split, the residual add, the gate-partial fold, the attention scale.  With PK the
// QM9 kernel issues 13 % fewer VALU instructions per tile (static census 3 887 -> 3 390) -- and runs SLOWER: tile 62 290 -> 63 170 cycles with the
// packed forms in the VALU phases only, -> 67 070 with them between the MFMAs of the GEMM phases as well (a GEMM phase 9 620 -> 11 000 cycles);
// GEOM 61 940 -> 60 960 / 63 900.  Same bits in every variant (forward and sampler hashes equal).  A packed fp32 instruction costs this kernel what
// the two scalar ones cost, and beside MFMAs much more (profiles/r05_packed_fp32_ab.md; round 3 saw the same with the compiler's own packing).
// So: PK = false, and the library keeps the rounds 1-4 build line (-packed-fp32-ops off; the assembler refuses v_pk_*_f32 without the feature).
//     A/B:  GCDM_BUILD_BASE=-fno-slp-vectorize tools/build_variants.sh "pk:-DGCDM_X3_PK=1" "pkB:-DGCDM_X3_PK=1 -DGCDM_X3_PK_GEMM=0"
typedef float f32x2 __attribute__((ext_vector_type(2)));
#if defined(GCDM_X3_PK) && GCDM_X3_PK
constexpr bool X3_PK = true;
#else
constexpr bool X3_PK = false;
#endif
// ... and whether the pieces that are issued BETWEEN the MFMAs of a GEMM phase (the tail-skewed SiLU of N-tile 0, the hooked vector stages) use them too
#ifdef GCDM_X3_PK_GEMM
constexpr bool X3_PKG = X3_PK && (GCDM_X3_PK_GEMM != 0);
#else
constexpr bool X3_PKG = X3_PK;
#endif
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { if constexpr (PK) return a * b; else return (f32x2){a[0] * b[0], a[1] * b[1]}; }
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { if constexpr (PK) return a + b; else return (f32x2){a[0] + b[0], a[1] + b[1]}; }
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 pk_mul1(f32x2 a, float s) { if constexpr (PK) return a * (f32x2){s, s}; else return (f32x2){a[0] * s, a[1] * s}; }
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 pk_add1(f32x2 a, float s) { if constexpr (PK) return a + (f32x2){s, s}; else return (f32x2){a[0] + s, a[1] + s}; }
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 pk_fma1(f32x2 a, float s, f32x2 c) {
    if constexpr (PK) return __builtin_elementwise_fma(a, (f32x2){s, s}, c);
    else return (f32x2){__builtin_fmaf(a[0], s, c[0]), __builtin_fmaf(a[1], s, c[1])};
}
// SiLU (scaled units, see silu_scaled) of two adjacent channels: exp2 x2, packed add, rcp x2, packed multiply -- 6 issue slots instead of 8
#ifdef GCDM_ABL_NOSILU
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 silu_scaled2(f32x2 xs) { return xs; }
#else
template <bool PK = X3_PK> __device__ __forceinline__ f32x2 silu_scaled2(f32x2 xs) {
    const f32x2 e = {__builtin_amdgcn_exp2f(xs[0]), __builtin_amdgcn_exp2f(xs[1])};
    const f32x2 d = pk_add1<PK>(e, 1.0f);
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return pk_mul<PK>(xs, r);
}
#endif
// merge of the two accumulators + SiLU over one 32 x 32 accumulator tile (16 registers per lane)
template <bool PK = X3_PK> __device__ __forceinline__ void silu_merge16(f32x16& dst, const f32x16& am, const f32x16& al, float inv) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 p = pk_fma1<PK>((f32x2){al[r], al[r + 1]}, inv, (f32x2){am[r], am[r + 1]});
        const f32x2 s = silu_scaled2<PK>(p);
        dst[r] = s[0]; dst[r + 1] = s[1];
    }
}
// the node kernels' form (true units, fast_silu of gcdm_kernels.hip.h): merge, x * rcp(1 + exp(-x)) with the add and the product packed
__device__ __forceinline__ void fast_silu_merge16(f32x16& dst, const f32x16& am, const f32x16& al, float inv) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 p = pk_fma1((f32x2){al[r], al[r + 1]}, inv, (f32x2){am[r], am[r + 1]});
        const f32x2 e = {__expf(-p[0]), __expf(-p[1])};
        const f32x2 d = pk_add1(e, 1.0f);
        const f32x2 q = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        const f32x2 s = pk_mul(p, q);
        dst[r] = s[0]; dst[r + 1] = s[1];
    }
}
// merge only (GCP2s without a scalar nonlinearity)
__device__ __forceinline__ void merge16(f32x16& am, const f32x16& al, float inv) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const f32x2 p = pk_fma1((f32x2){al[r], al[r + 1]}, inv, (f32x2){am[r], am[r + 1]});
        am[r] = p[0]; am[r + 1] = p[1];
    }
}

// v_sqrt_f32 (1 ulp) instead of the ~20-instruction correctly rounded expansion: the argument is >= 1e-8, never denormal
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// How the 16 exponent bits of an f16 pair are shared between weights and activations is a property of the CHECKPOINT, chosen once per handle
// (gcdm_finalize_weights): packed weights carry w = 2^(11-k), activation images 1/w, with k the smallest shift that keeps the largest
// weight inside f16 -- k = 0 (|W| < 31.9, activations up to 1.3e8) for every model seen so far; a checkpoint with larger weights moves the
// split (k = 5: |W| < 1023, activations up to 4.2e6) instead of losing the split-precision mode.  The three constants are the first
// member of every split-precision kernel's argument struct and are read from the kernel-argument segment (scalar loads, no plumbing through
// the device functions):   hi = f16(x * pre),  lo' = f16(x - hi * scale),  result = am + al * inv   (inv = pre; all powers of two, exact).
struct X3Const {
    float pre, scale, inv, range;          // 2^(k-11), 2^(11-k), 2^(k-11), 6e4 * scale (bound on the un-scaled activation)
};
typedef const X3Const __attribute__((address_space(4))) * x3const_ptr;
#define X3_KARG ((x3const_ptr)__builtin_amdgcn_kernarg_segment_ptr())
#define X3_SCALE (X3_KARG->scale)
#define X3_INV_SCALE (X3_KARG->inv)
#define X3_PRE (X3_KARG->pre)
#define X3_RANGE (X3_KARG->range)
// Range guard.  An activation beyond the images' range becomes inf in its hi image, every product with it is inf / NaN, and -- all state
// updates of the network being residual -- the non-finite value reaches the network output: the LAST node kernel raises
// GCDM_FLAG_F16_RANGE when vel or a projected feature is not finite (the caller then re-runs in fp32 mode, which also decides whether a
// NaN was the model's own).  Per-element |x| tracking in every split cost ~10 % of the edge kernel's VALU instructions; compile with
// -DGCDM_X3_TRACK_RANGE to get it back (debugging).
#ifdef GCDM_X3_TRACK_RANGE
#define X3_TRACK(amax, x0, x1) (amax) = fmaxf((amax), fmaxf(fabsf(x0), fabsf(x1)))
#define X3_OVER(expr) (expr)
#else
#define X3_TRACK(amax, x0, x1) ((void)0)
#define X3_OVER(expr) false
#endif

__device__ __forceinline__ void split16(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)(x * X3_PRE);
    // x - hi * 2^11 (hi holds x * 2^-11), written so that it maps to one mixed-precision FMA (v_fma_mixlo_f16): the product is exact (power of two)
    lo = (_Float16)__builtin_fmaf((float)hi, -X3_SCALE, x);
}

// Two values at once.  tools/valu_ubench8.hip (MI355X, two waves per SIMD): a v_fma_mix{lo,hi}_f16 -- the 16-bit-destination form --
// occupies the SIMD for 6.9 clk, v_fma_mix_f32 / v_cvt_pk_f16_f32 / v_mul_f32 for 3.0-3.8.  So instead of four f16-destination FMAs per
// pair (round 2) the split is: scale (v_mul x2; the packed v_pk_mul_f32 measured slower beside MFMAs), v_cvt_pk_f16_f32 (gfx950: round-to-nearest pack),
// two f32 mixed FMAs that read hi as f16 from the low / high half, v_cvt_pk_f16_f32: 5-6 full-rate instructions instead of 4 half-rate
// ones (-15 % / -20 % on the state-image phase in the micro-benchmark), every destination a full 32-bit write.  Bit-identical to split16:
// all products are by powers of two, x - hi * 2^11 is exact in fp32, one rounding to f16 at the end of each half.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <bool PK = X3_PK>
__device__ __forceinline__ void split16x2(float x0, float x1, h2& hi, h2& lo) {
    const float pre = X3_PRE, neg = -X3_SCALE;
    uint32_t hiu, lou;
    const f32x2 t = pk_mul1<PK>((f32x2){x0, x1}, pre);    // (round 5: one packed multiply)
    const float t0 = t[0], t1 = t[1];
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hiu) : "v"(t0), "v"(t1));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hiu), "s"(neg), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hiu), "s"(neg), "v"(x1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lou) : "v"(r0), "v"(r1));
    __builtin_memcpy(&hi, &hiu, 4);
    __builtin_memcpy(&lo, &lou, 4);
}

// Registers written by the inline-asm splits above and consumed DIRECTLY by an MFMA pass through x3_settle first.  The last writer of
// such a register is a 16-bit partial write (op_sel destination).  On gfx950 an MFMA issued with NO wait state behind such a write reads
// the old register -- always when the two are back to back, in ~0.1 % of the cases when another MFMA was issued just before the write
// (tools/mfma_partial_write_hazard.hip); one wait state is enough there, and one is what the compiler inserts between an inline-asm
// definition and its consumer.  The edge-embedding kernel nevertheless produced stale B columns (16 of a wave's 32 edges, in a few of
// 1 300 waves, different waves from run to run, only in waves of the grid's second round) while the compiler interleaved the splits of
// one k-block with the MFMAs of the previous one; the exact instruction pair was not isolated (tests/gpu_ragged_diag.py: 16 s_nops
// around the MFMAs without a data dependence do not help, the compiler-generated split and this fence both do).  The fence is a
// data dependence: every split of an operand is complete, plus two wait states, before the first MFMA that reads it.
// Round 5: since round 3 the splits above end in v_cvt_pk_f16_f32 -- FULL 32-bit writes -- so the partial-write precondition of the fault is gone from every
// MFMA operand.  Round 6 priced the fence on alternating runs of one box (profiles/r06_hazards.txt): 58 825 -> 58 680 cycles per QM9 tile, 57 290 -> 57 115
// GEOM (-0.25 % / -0.3 %), the same output bits, the hazard probes green -- the shipped build is UN-fenced; -DGCDM_X3_SETTLE brings the fence back and
// tests/test_hazards_gpu.py builds that fallback and holds it to the shipped build's bits on the configuration that exposed the fault.
#ifndef GCDM_X3_SETTLE
__device__ __forceinline__ void x3_settle(h8&, h8&) {}
__device__ __forceinline__ void x3_settle(h8&) {}
#else
__device__ __forceinline__ void x3_settle(h8& a, h8& b) { asm("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void x3_settle(h8& a) { asm("s_nop 1" : "+v"(a)); }
#endif

// ---- tile GEMM on split operands: am += Whi.Xhi ; al += Whi.Xlo' + Wlo'.Xhi ------------------------------------------------
// A-operand ring of one GEMM (PD+1 statically indexed register sets of hi / lo' weights).  `x3_prefetch` issues the loads of the
// first PD k-blocks; it is called one phase EARLY (right after the previous GEMM), so that the cold L2 round trip of a GEMM that
// is only 7-18 blocks long overlaps the VALU phases instead of stalling the matrix pipe.
template <int MT, int PD>
struct X3Ring {
    h8 ah[PD + 1][MT], alo[PD + 1][MT];
};

// Weight stream through buffer loads: one resource descriptor for the whole weight pool (SGPRs), per-lane offset lane * 16 (one VGPR for
// the whole kernel), and the array / M-tile / k-block offset as a scalar -- the address arithmetic of the stream is SALU only (global
// loads cost the compiler a 64-bit VALU add per 4 KB of immediate-offset reach and stream).  Out-of-range reads return 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WPool {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;                 // lane * 16
    const char* base;
    __device__ __forceinline__ uint32_t off(const void* p) const { return (uint32_t)((const char*)p - base); }      // uniform
    __device__ __forceinline__ h8 ld(uint32_t soff) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        h8 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    }
    // k-block `blk` of a packed array at scalar offset `arr`: blocks are 1 KB apart, so four consecutive blocks share ONE scalar offset
    // and differ in the instruction's 12-bit immediate (the compiler folds a constant added to the lane offset into it): one s_add per
    // four blocks and array instead of one per load
    template <int BLK>
    __device__ __forceinline__ h8 ldk(uint32_t arr) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (uint32_t)((BLK & 3) * 1024), arr + (uint32_t)((BLK >> 2) * 4096), 0);
        h8 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    }
};
__device__ __forceinline__ WPool make_wpool(const void* pool, uint32_t bytes, int lane) {
    WPool w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pool), 0, bytes, 0x00020000);
    w.voff = (uint32_t)lane * 16u;
    w.base = (const char*)pool;
    return w;
}

// the same for the per-edge / per-node fp32 arrays of the workspace pool: per-lane byte offset (an edge or node index, computed once) in
// the VGPR, array base + row * stride as the scalar offset
struct BufView {
    __amdgpu_buffer_rsrc_t rsrc;
    const char* base;
    __device__ __forceinline__ uint32_t off(const void* p) const { return (uint32_t)((const char*)p - base); }
    __device__ __forceinline__ float ld1(uint32_t voff, uint32_t soff) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0)); }
    __device__ __forceinline__ v4f ld4(uint32_t voff, uint32_t soff) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        return __builtin_bit_cast(v4f, v);
    }
    // per-edge constants: read once per launch by one workgroup (136 MB per layer at QM9 1024 x 19, streaming through the L2 that also has to keep the
    // layer's 2.9 MB of weights).  AUX = 2 is the non-temporal hint: QM9 tile 62.77 k -> 62.34 k cycles, step -0.9 %; GEOM (a quarter of the bytes per
    // edge) +0.25 % -- so the 64-channel edge width uses it, the 16-channel one does not (round 4; same bits).  Non-temporal STORES of the aggregated
    // rows were measured too: +1.7 % (QM9), +-0 (GEOM) -- not used.
    template <int AUX>
    __device__ __forceinline__ float ld1s(uint32_t voff, uint32_t soff) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, AUX)); }
    template <int AUX>
    __device__ __forceinline__ v4f ld4s(uint32_t voff, uint32_t soff) const {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, AUX);
        return __builtin_bit_cast(v4f, v);
    }
};
__device__ __forceinline__ BufView make_view(const void* pool, uint32_t bytes) {
    BufView b;
    b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pool), 0, bytes, 0x00020000);
    b.base = (const char*)pool;
    return b;
}

// One explicit wait for ALL A operands of the current k-block (at most VM younger loads -- the next blocks' -- stay in flight) in front of
// its MFMAs: the compiler's own per-operand waits (2-3 per block) become redundant and are dropped.  Never weaker than what the
// compiler would insert, so only the instruction count changes.
template <int VM>
__device__ __forceinline__ void x3_wait_block() {
    static_assert(VM < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt((15 << 8) | (7 << 4) | (VM & 15) | ((VM >> 4) << 14));
}

template <int MT, int PD>
__device__ __forceinline__ void x3_prefetch_b(X3Ring<MT, PD>& ring, const WPool& wp, uint32_t oH, uint32_t oL, int KB) {
    const uint32_t wstride = KB * 64 * 16;
#pragma unroll
    for (int r = 0; r < PD; ++r) {
#pragma unroll
        for (int m = MT - 1; m >= 0; --m) ring.alo[r][m] = wp.ld(oL + m * wstride + r * 1024);
#pragma unroll
        for (int m = MT - 1; m >= 0; --m) ring.ah[r][m] = wp.ld(oH + m * wstride + r * 1024);
    }
}

// NOTE: the prefetches run PD k-blocks (A) / one k-block (B) past the end of the contraction without clamping: the packed weight
// arrays carry X3_TAIL_BLOCKS zero blocks of padding behind the last M-tile, and the LDS reads stay inside the XH8|XL8|VV allocation.
#define X3_TAIL_BLOCKS 4
// K' of msg0's per-edge part (kernel and host packing, gcdm_api.hip): [e' (Se) | norms of the H0 hidden vectors | 9 frame scalars | 0 ...] in units of
// 16-deep k-blocks.  Round 5: the three pieces are COMPACT -- 64 + 20 + 9 = 93 slots = 6 k-blocks at QM9, 16 + 18 + 9 = 43 = 3 at GEOM; rounds 1-4 started
// the norms and the frame scalars on 8-slot groups of their own (7 / 4 k-blocks; -DGCDM_X3_MSG0_PADDED brings that layout back for an A/B).  The contraction
// order of msg0 changes with the layout: not the same bits as round 4, same error model.
#ifdef GCDM_X3_MSG0_PADDED
__host__ __device__ constexpr int x3_msg0_qpos(int Se, int H0) { return 8 * (Se / 8 + (H0 + 7) / 8); }
__host__ __device__ constexpr int x3_msg0_kb(int Se, int H0) { return (x3_msg0_qpos(Se, H0) + 16 + 15) / 16; }
#else
__host__ __device__ constexpr int x3_msg0_qpos(int Se, int H0) { return Se + H0; }
__host__ __device__ constexpr int x3_msg0_kb(int Se, int H0) { return (Se + H0 + 9 + 15) / 16; }
#endif
constexpr int X3_PD = 2;         // k-blocks of weight prefetch distance in the edge kernel (register ring of PD + 1 sets)
#ifndef GCDM_STAMP_K
#define GCDM_STAMP_K 0              // which residual GCP2 (0..2) carries the per-phase stamps 10..17 of a -DGCDM_STAMPS build
#endif
#ifndef GCDM_X3_TAIL_PER_MFMA
#define GCDM_X3_TAIL_PER_MFMA 8      // round 5, shipped-style builds (end-of-tile stamp): 4 / 6 / 8 / 10 / 12 / 14 / 20 -> 59 360 / 59 550 / 59 460 / 59 470 / 59 670 / 59 820 / 59 530 cycles (QM9), same order at GEOM
#endif
#ifndef GCDM_X3_VEC_PER_MFMA
#define GCDM_X3_VEC_PER_MFMA 6
#endif
constexpr int X3_TAIL_PER_MFMA = GCDM_X3_TAIL_PER_MFMA;  // tail skew: instructions of N-tile 0's SiLU issued behind each of N-tile 1's last MFMAs
constexpr int X3_VEC_PER_MFMA = GCDM_X3_VEC_PER_MFMA;   // instructions of a vector stage issued behind each MFMA of the hosting k-block (0 / 3 / 4 / 10: +-0.3 %)

template <int MT, int PD>
__device__ __forceinline__ void x3_prefetch(X3Ring<MT, PD>& ring, const h8* __restrict__ wH, const h8* __restrict__ wL, int KB, int lane) {
    const int wstride = KB * 64;
#pragma unroll
    for (int r = 0; r < PD; ++r) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { ring.ah[r][m] = wH[m * wstride + r * 64 + lane]; ring.alo[r][m] = wL[m * wstride + r * 64 + lane]; }
    }
}

// KBC > 0: the number of k-blocks is a compile-time constant (no tail branches, no accumulator copies at their joins)
template <int MT, int NT, int PD, int KBC = 0>
__device__ __forceinline__ void tile_gemm_x3(f32x16 (&am)[MT][NT], f32x16 (&al)[MT][NT], X3Ring<MT, PD>& ring, const h8* __restrict__ wH,
                                             const h8* __restrict__ wL, int KBrt, const h8* xh8, const h8* xl8, int TP, int lane) {
    constexpr int R = PD + 1;
    const int KB = KBC ? KBC : KBrt;
    const int wstride = KB * 64;
    const h8* wh = wH + lane + PD * 64;          // running pointers: one add per k-block, no clamping
    const h8* wl = wL + lane + PD * 64;
    const int boff = (lane >> 5) * TP + (lane & 31);
    const h8* sh = xh8 + boff;
    const h8* sl = xl8 + boff;
    h8 bh[2][NT], bl[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = sh[n * 32]; bl[0][n] = sl[n * 32]; }
    auto body = [&](int r) {
#pragma unroll
        for (int m = 0; m < MT; ++m) { ring.ah[(r + PD) % R][m] = wh[m * wstride]; ring.alo[(r + PD) % R][m] = wl[m * wstride]; }
        wh += 64;
        wl += 64;
        sh += 2 * TP;
        sl += 2 * TP;
#pragma unroll
        for (int n = 0; n < NT; ++n) { bh[(r + 1) & 1][n] = sh[n * 32]; bl[(r + 1) & 1][n] = sl[n * 32]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(ring.ah[r % R][m], bh[r & 1][n], am[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.ah[r % R][m], bl[r & 1][n], al[m][n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.alo[r % R][m], bh[r & 1][n], al[m][n]);
        __builtin_amdgcn_sched_barrier(0);
    };
    int k0 = 0;
    for (; k0 + 2 * R <= KB; k0 += 2 * R) {
#pragma unroll
        for (int r = 0; r < 2 * R; ++r) body(r);
    }
#pragma unroll
    for (int r = 0; r < 2 * R - 1; ++r)
        if (k0 + r < KB) body(r);
}

// Fully unrolled variant for compile-time k-block counts whose FIRST block starts the accumulators from the MFMA's inline zero C operand
// (ZAM: the main accumulator too; ZAL: the 2^-11-scaled one) -- saves the v_mov initialisation of 16 registers per accumulator tile.
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

struct NoTail {
    __device__ __forceinline__ void operator()() const {}
};

// TAIL SKEW (round 4) of a two-N-tile GEMM: the last two k-blocks r0, r0 + 1 run N-tile 0 first (6 MFMAs), then N-tile 1 (6 MFMAs) with `tail()` -- the
// SiLU of N-tile 0's finished accumulators -- issued between them: a wave's own VALU / transcendental instructions cost ~1-3 clk between its MFMAs
// instead of 4 / ~11 in a VALU phase (profiles/r04_overlap_experiments.md).  Both blocks' A operands are in the ring anyway (PD = 2), the B operands
// take the two halves of the double buffer (bh / bl [r0 & 1] must already hold block r0); every accumulator sees its MFMAs in the same order as in
// the plain loop: same bits.
template <int PD, int R0, class Tail>
__device__ __forceinline__ void x3_tail_skew(f32x16 (&am)[1][2], f32x16 (&al)[1][2], X3Ring<1, PD>& ring, h8 (&bh)[2][2], h8 (&bl)[2][2], const h8* sh,
                                             const h8* sl, int TP, Tail&& tail) {
    static_assert(PD >= 2, "tail skew: two k-blocks of A operands resident");
    constexpr int r0 = R0, r1 = R0 + 1, R_ = PD + 1;
#pragma unroll
    for (int n = 0; n < 2; ++n) { bh[r1 & 1][n] = sh[r1 * 2 * TP + n * 32]; bl[r1 & 1][n] = sl[r1 * 2 * TP + n * 32]; }
    x3_wait_block<0>();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        am[0][n] = MFMA16(ring.ah[r0 % R_][0], bh[r0 & 1][n], am[0][n]);
        al[0][n] = MFMA16(ring.ah[r0 % R_][0], bl[r0 & 1][n], al[0][n]);
        al[0][n] = MFMA16(ring.alo[r0 % R_][0], bh[r0 & 1][n], al[0][n]);
        am[0][n] = MFMA16(ring.ah[r1 % R_][0], bh[r1 & 1][n], am[0][n]);
        al[0][n] = MFMA16(ring.ah[r1 % R_][0], bl[r1 & 1][n], al[0][n]);
        al[0][n] = MFMA16(ring.alo[r1 % R_][0], bh[r1 & 1][n], al[0][n]);
        if (n == 0) __builtin_amdgcn_sched_barrier(0);
    }
    tail();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, X3_TAIL_PER_MFMA, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MT, int NT, int PD, int KB, bool ZAM, bool ZAL, int RO, class Tail = NoTail>
__device__ __forceinline__ void tile_gemm_x3z(f32x16 (&am)[MT][NT], f32x16 (&al)[MT][NT], X3Ring<MT, PD>& ring, const WPool& wp, uint32_t wH,
                                              uint32_t wL, const h8* xh8, const h8* xl8, int TP, int lane, Tail&& tail = NoTail{}) {
    constexpr bool SKEW = !std::is_same<std::decay_t<Tail>, NoTail>::value && NT == 2 && MT == 1 && KB >= 3;
    constexpr int KBL = SKEW ? KB - 2 : KB;              // k-blocks of the plain loop
    constexpr int R = PD + 1;
    constexpr int wstride = KB * 64;
    // weights: uniform base (SGPR) + compile-time block offset + lane -- no per-block 64-bit VALU pointer arithmetic
    const int boff = (lane >> 5) * TP + (lane & 31);
    const h8* sh = xh8 + boff;
    const h8* sl = xl8 + boff;
    h8 bh[2][NT], bl[2][NT];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = sh[n * 32]; bl[0][n] = sl[n * 32]; }
    static_for<0, KBL>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        auto loads = [&] {
            if constexpr (r + PD < KB) {             // (the skewed tail needs blocks KB-2, KB-1 and nothing behind them)
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.alo[(r + PD) % R][m] = wp.template ldk<r + PD>(wL + m * wstride * 16);
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.ah[(r + PD) % R][m] = wp.template ldk<r + PD>(wH + m * wstride * 16);
            } else if constexpr (!SKEW) {
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.alo[(r + PD) % R][m] = wp.template ldk<r + PD>(wL + m * wstride * 16);
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.ah[(r + PD) % R][m] = wp.template ldk<r + PD>(wH + m * wstride * 16);
            }
        };
        auto breads = [&] {
#pragma unroll
            for (int n = 0; n < NT; ++n) { bh[(r + 1) & 1][n] = sh[(r + 1) * 2 * TP + n * 32]; bl[(r + 1) & 1][n] = sl[(r + 1) * 2 * TP + n * 32]; }
        };
        auto mfmas = [&] {
#if !defined(GCDM_X3_MFMA_ORDER) || GCDM_X3_MFMA_ORDER >= 1
            if constexpr (RO >= 1 && MT == 1 && NT == 2) {      // the block's six products in the order of tile_gemm_x3s (round 6: al0 al1 am0 al0 al1 am1; same bits)
                al[0][0] = MFMA16(ring.ah[r % R][0], bl[r & 1][0], (ZAL && r == 0) ? zero : al[0][0]);
                al[0][1] = MFMA16(ring.ah[r % R][0], bl[r & 1][1], (ZAL && r == 0) ? zero : al[0][1]);
                am[0][0] = MFMA16(ring.ah[r % R][0], bh[r & 1][0], (ZAM && r == 0) ? zero : am[0][0]);
                al[0][0] = MFMA16(ring.alo[r % R][0], bh[r & 1][0], al[0][0]);
                al[0][1] = MFMA16(ring.alo[r % R][0], bh[r & 1][1], al[0][1]);
                am[0][1] = MFMA16(ring.ah[r % R][0], bh[r & 1][1], (ZAM && r == 0) ? zero : am[0][1]);
                return;
            }
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(ring.ah[r % R][m], bh[r & 1][n], (ZAM && r == 0) ? zero : am[m][n]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.ah[r % R][m], bl[r & 1][n], (ZAL && r == 0) ? zero : al[m][n]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.alo[r % R][m], bh[r & 1][n], al[m][n]);
        };
        // the next blocks' operand requests ride between this block's MFMAs (see tile_gemm_x3s)
        static_assert(PD >= 2, "interleaved requests are waited for one block later: two blocks of prefetch distance");
        x3_wait_block<2 * MT * (PD - 1)>();
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        breads();
        loads();
#pragma unroll
        for (int i = 0; i < 3 * MT * NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 2 * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (i < 2 * NT + 2 * MT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (SKEW) x3_tail_skew<PD, KB - 2>(am, al, ring, bh, bl, sh, sl, TP, tail);
}

// gate partial from registers: contraction over the channels this wave holds (two 16-deep blocks per M-tile)
template <int MT, int NT, bool ZERO = false>
__device__ __forceinline__ void gate_partial_x3(f32x16 (&gm)[NT], f32x16 (&gl)[NT], const f32x16 (&act)[MT][NT], const h8* __restrict__ wgH,
                                                const h8* __restrict__ wgL, int mt0, int lane) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const h8 aH = wgH[((mt0 + m) * 2 + j) * 64 + lane];
            const h8 aL = wgL[((mt0 + m) * 2 + j) * 64 + lane];
            h8 bh[NT], bl[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int s = 0; s < 8; s += 2) {
                    h2 hi, lo;
                    split16x2(act[m][n][8 * j + s], act[m][n][8 * j + s + 1], hi, lo);
                    bh[n][s] = hi[0]; bh[n][s + 1] = hi[1];
                    bl[n][s] = lo[0]; bl[n][s + 1] = lo[1];
                }
#pragma unroll
            for (int n = 0; n < NT; ++n) x3_settle(bh[n], bl[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gm[n] = MFMA16(aH, bh[n], (ZERO && m == 0 && j == 0) ? zero : gm[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gl[n] = MFMA16(aH, bl[n], (ZERO && m == 0 && j == 0) ? zero : gl[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gl[n] = MFMA16(aL, bh[n], gl[n]);
        }
}

// The same with the A operands (this wave's slices of vector_out_scale: 2 k-blocks per M-tile, hi and lo') requested AHEAD of the SiLU
// that produces the B operand: as plain loads at the point of use their L2 round trip was exposed once per GCP2 (-3 % of the tile's
// cycles in the ablation, tools/ab_run.sh g_noload)
template <int MT>
struct GateW {
    h8 aH[MT][2], aL[MT][2];
};
template <int MT>
__device__ __forceinline__ void gate_prefetch(GateW<MT>& g, const WPool& wp, const h8* wgH, const h8* wgL, int mt0) {
    const uint32_t oH = wp.off(wgH + (size_t)mt0 * 2 * 64), oL = wp.off(wgL + (size_t)mt0 * 2 * 64);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) { g.aH[m][j] = wp.ld(oH + (m * 2 + j) * 1024); g.aL[m][j] = wp.ld(oL + (m * 2 + j) * 1024); }
}
template <int MT, int NT, bool ZERO = false>
__device__ __forceinline__ void gate_partial_x3p(f32x16 (&gm)[NT], f32x16 (&gl)[NT], const f32x16 (&act)[MT][NT], const GateW<MT>& g) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h8 bh[NT], bl[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int s = 0; s < 8; s += 2) {
                    h2 hi, lo;
                    split16x2(act[m][n][8 * j + s], act[m][n][8 * j + s + 1], hi, lo);
                    bh[n][s] = hi[0]; bh[n][s + 1] = hi[1];
                    bl[n][s] = lo[0]; bl[n][s + 1] = lo[1];
                }
#pragma unroll
            for (int n = 0; n < NT; ++n) x3_settle(bh[n], bl[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gm[n] = MFMA16(g.aH[m][j], bh[n], (ZERO && m == 0 && j == 0) ? zero : gm[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gl[n] = MFMA16(g.aH[m][j], bl[n], (ZERO && m == 0 && j == 0) ? zero : gl[n]);
#pragma unroll
            for (int n = 0; n < NT; ++n) gl[n] = MFMA16(g.aL[m][j], bh[n], gl[n]);
        }
}

// Gate partials PG[slot][e][32 c] (fp32, c contiguous: the vector waves read the four channels they own as one ds_read_b128 per slot).
// 16-byte granule (c >> 2) of row e sits at granule index (c >> 2) ^ (e & 7): conflict-free both for the writers (32x32 accumulator
// layout: lane = edge, 4 consecutive channels per register quad) and for the readers (16x16 layout of the vector-path MFMAs).
// The 8 channel blocks are folded pairwise in two stages (LDS float atomics run at ~1 lane/clk on gfx950): waves 0-3 store their
// partial into slot w, (barrier), waves 4-7 add theirs onto slot w-4.  Fixed order -> deterministic.  (A symmetric fold -- every wave
// stores one N-tile and adds the other -- was measured: +0.5 %; waves s and s + 4 share a SIMD, the "idle" wave never cost SIMD time.
// What pays is that the barrier between the two stages already says "all waves are done reading the operand images": no barrier behind the fold.)
template <int ET>
__device__ __forceinline__ int pg_off(int slot, int e, int c4) { return ((slot * ET + e) * 32 + 4 * (c4 ^ (e & 7))); }

template <int NT, int ET>
__device__ __forceinline__ void put_gate_partial(float* PG, const f32x16 (&gm)[NT], const f32x16 (&gl)[NT], int slot, int lane, bool add) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float inv = X3_INV_SCALE;
            f32x2 v0 = pk_fma1((f32x2){gl[n][4 * t], gl[n][4 * t + 1]}, inv, (f32x2){gm[n][4 * t], gm[n][4 * t + 1]});
            f32x2 v1 = pk_fma1((f32x2){gl[n][4 * t + 2], gl[n][4 * t + 3]}, inv, (f32x2){gm[n][4 * t + 2], gm[n][4 * t + 3]});
            v4f* p = (v4f*)(PG + pg_off<ET>(slot, 32 * n + l31, 2 * t + half));      // channels 8t + 4 half + {0..3}
            if (add) {
                const v4f o = *p;
                v0 = pk_add((f32x2){o[0], o[1]}, v0);
                v1 = pk_add((f32x2){o[2], o[3]}, v1);
            }
            *p = (v4f){v0[0], v0[1], v1[0], v1[1]};
        }
}

// (node kernels) gate partials PG[slot][c][e], same two-stage fold
template <int NT>
__device__ __forceinline__ void put_gate_partial(float* PG, const f32x16 (&gm)[NT], const f32x16 (&gl)[NT], int TP, int slot, int lane, bool add) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* p = &PG[(slot * 32 + c) * TP + 32 * n + l31];
            const float v = gm[n][r] + gl[n][r] * X3_INV_SCALE;
            *p = add ? *p + v : v;
        }
}

// hi / lo' images of the wave's fp32 state -> XH8 / XL8 (8 bytes per lane and group: channels 8q+4*half+{0..3})
template <int MT, int NT>
__device__ __forceinline__ void store_state_x3(char* XH, char* XL, int gbase8, const f32x16 (&st)[MT][NT], int TP, int mt0, int lane, float& amax) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h4 vh, vl;
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    const float x0 = st[m][n][4 * q + t], x1 = st[m][n][4 * q + t + 1];
                    X3_TRACK(amax, x0, x1);
                    h2 hi, lo;
                    split16x2(x0, x1, hi, lo);
                    vh[t] = hi[0]; vh[t + 1] = hi[1];
                    vl[t] = lo[0]; vl[t + 1] = lo[1];
                }
                const int off = ((gbase8 + 4 * (mt0 + m) + q) * TP + 32 * n + l31) * 16 + 8 * half;
                *(h4*)(XH + off) = vh;
                *(h4*)(XL + off) = vl;
            }
}

template <int MT, int NT>
__device__ __forceinline__ bool store_state_x3(char* XH, char* XL, int gbase8, const f32x16 (&st)[MT][NT], int TP, int mt0, int lane) {
    float amax = 0.f;
    store_state_x3<MT, NT>(XH, XL, gbase8, st, TP, mt0, lane, amax);
    return amax > X3_RANGE;
}

__device__ __forceinline__ bool put16(char* XH, char* XL, int TP, int g8, int slot, int e, float x) {
    _Float16 hi, lo;
    split16(x, hi, lo);
    const int off = (g8 * TP + e) * 16 + 2 * slot;
    *(_Float16*)(XH + off) = hi;
    *(_Float16*)(XL + off) = lo;
    return X3_OVER(fabsf(x) > X3_RANGE);
}

// ---- the vector path of the message GCP2s on the matrix pipe ----------------------------------------------------------------
// The per-edge vector contractions of a GCP2 -- vector_down / vector_down_frames ([11 x 32] . [32 x 3] per edge, gcpnet.py:442-459)
// and vector_up ([32 x H] . [H x 3], :388-411) -- are GEMMs with a tiny M: as VALU FMAs they cost 13 % of the step for 2 % of the
// FLOPs (8 threads share an edge and each re-reads all 96 vector components).  Here one wave per 16 edges ("vector wave") evaluates
// them with v_mfma_f32_16x16x32_f16 (split precision, same error model as the scalar GEMMs), and -- because a SIMD does not overlap
// one wave's VALU work with another wave's MFMAs (tools/mfma_ubench6.hip) -- it does so IN THE SHADOW OF ITS OWN SCALAR GEMM: the
// vector part of GCP2 k-1 (vector_up, gate, residual update) and the pre-phase of GCP2 k (vector_down, norms, frame scalars) are cut
// into 16 branch-free stages that are issued between the MFMAs of the first 16 k-blocks of GEMM k; the last two k-blocks (the
// extended-K rows the pre-phase produces) follow after a workgroup barrier.
//   operand maps (lane l: q = l >> 4, n = l & 15):  A[row n][k = 8q + j]   B[k = 8q + j][col n]   D[row 4q + i][col n]
//   columns = the wave's 16 edges, one MFMA set per spatial component x.
// The fp32 master of the message vectors lives in LDS as float4 channel groups VV4[x][cg][e]; lane (q, n) owns groups cg = q and
// 4 + q of edge n for the whole tile, which is at once
//   * the D layout of vector_up (M-tile m = channels 16m .. 16m+15: rows 4q + i <-> channels 16m + 4q + i = group 4m + q), and
//   * a valid B layout of vector_down, because the contraction order is free: k = 8q + j <-> channel (j < 4 ? 4q + j : 16 + 4q + j - 4)
//     (the host packs A with that column permutation),
// so vector_up(k-1) -> vector_down(k) runs on registers without any cross-lane traffic.
// vector_down's 16 output rows are dealt out so that all lanes run the same code: D row 4q + i = hidden vector 3q + i for i < 3 (8 real
// ones), = frame vector q for i = 3 (3 real ones).  Lane (q, n) therefore produces the 16-byte extended-K group 32 + q of its edge,
// [n(3q) n(3q+1) n(3q+2) | q(3q) q(3q+1) q(3q+2) | 0 | 1 (q = 3: the bias column)], and the image [hi(3) 0 | lo'(3) 0] of its three hidden
// vectors that feeds vector_up's B operand (k = 8q + j <-> hidden channel 3q + (j & 3); two MFMAs, A1 = [W_hi | 0], A2 = [W_lo' | W_hi]).
#define MFMA1632(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool PK = X3_PK>
__device__ __forceinline__ void split4(const float (&x)[4], h4& hi, h4& lo, float& amax) {
#pragma unroll
    for (int s = 0; s < 4; s += 2) {
        h2 a, b;
        split16x2<PK>(x[s], x[s + 1], a, b);
        hi[s] = a[0]; hi[s + 1] = a[1];
        lo[s] = b[0]; lo[s + 1] = b[1];
        X3_TRACK(amax, x[s], x[s + 1]);
    }
}

__device__ __forceinline__ h8 cat44(h4 a, h4 b) { return (h8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

template <int ET, int H0, bool FIRST, bool PIPE = false>
struct VecStage {
    static constexpr int ETP = ET + 1;
    static constexpr int NSTAGES = 16;
    // tile state (LDS) and this lane's place in it
    const float* PG; const float* bg; const float* FR; const float* VH; h8* VHB; v4f* VV4; char* XH; char* XL;
    const h8* fA; const h8* fB;          // A operands of the finish part [2][64]: residual GCP2: A1, A2; msg0 (FIRST): W_hi, W_lo'
    const h8* pH; const h8* pL;          // A operand of the pre part [64]
    int ve, vq, lane;
    // registers handed from stage to stage
    v4f g0, g1;
    h8 w1[2], w2[2], pa[2];
    h8 bh[3], bl[3];
    v4f va[3], vb[3];
    float o[3][4];
    float vraw[8];
    float ev[6];
    h4 sh0, sl0, sh1, sl1;
    float amax = 0.f;
    bool preloaded = false;          // w1 / w2 were requested by the caller (finish_only of the last GCP2: ahead of the next tile's HBM prefetches)

    __device__ __forceinline__ void load_fin() { w1[0] = fA[lane]; w1[1] = fA[64 + lane]; w2[0] = fB[lane]; w2[1] = fB[64 + lane]; }
    __device__ __forceinline__ v4f gate_sum(int m) const {
        v4f g = *(const v4f*)(bg + 16 * m + 4 * vq);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4f v = *(const v4f*)(PG + pg_off<ET>(s, ve, 4 * m + vq));
            if constexpr (X3_PKG) g += v;
            else { g[0] += v[0]; g[1] += v[1]; g[2] += v[2]; g[3] += v[3]; }
        }
        return g;
    }
    static __device__ __forceinline__ v4f sig4(v4f g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = fast_sigmoid(g[i]);
        return g;
    }
    // vector_up of component x (both M-tiles), gate, (residual) update of the message vectors
    // (`b` = which image register set holds the B operand, `x` = the spatial component it belongs to)
    __device__ __forceinline__ void finish_x(int b, int x = -1) {
        if (x < 0) x = b;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
            if (FIRST) {
                am = MFMA1632(w1[m], bh[b], am);
                al = MFMA1632(w1[m], bl[b], al);
                al = MFMA1632(w2[m], bh[b], al);
            } else {
                am = MFMA1632(w1[m], bh[b], am);
                al = MFMA1632(w2[m], bh[b], al);
            }
            const v4f sg = m == 0 ? g0 : g1;
            v4f* p = &VV4[(x * 8 + 4 * m + vq) * ETP + ve];
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (!FIRST) v = *p;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (am[i] + al[i] * X3_INV_SCALE) * sg[i];
            *p = v;
            if (m == 0) va[x] = v; else vb[x] = v;
        }
    }
    // software-pipelined form (PIPE, round 4): the small MFMAs of a step are issued one stage (= one k-block of the hosting GEMM) ahead of the VALU
    // instructions that read their results, so that the wave never waits in order for a matrix result.  Costs 24 more live registers: used where the
    // kernel has them (the 8-channel edge width, GEOM: tile 62.4 k -> 61.9 k cycles, same bits); the 16-channel instantiation would spill (+2 %).
    f32x4 fam[2], fal[2];
    v4f vold[2];
    f32x4 pam, pal;
    __device__ __forceinline__ void issue_f(int b, int x) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            fam[m] = MFMA1632(w1[m], bh[b], z);
            fal[m] = MFMA1632(w2[m], bh[b], z);
            vold[m] = VV4[(x * 8 + 4 * m + vq) * ETP + ve];
        }
    }
    __device__ __forceinline__ void consume_f(int x) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const v4f sg = m == 0 ? g0 : g1;
            v4f v = vold[m];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (fam[m][i] + fal[m][i] * X3_INV_SCALE) * sg[i];
            VV4[(x * 8 + 4 * m + vq) * ETP + ve] = v;
            if (m == 0) va[x] = v; else vb[x] = v;
        }
    }
    __device__ __forceinline__ void issue_p() {
        h8 xh = cat44(sh0, sh1), xl = cat44(sl0, sl1);
        x3_settle(xh, xl);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        pam = MFMA1632(pa[0], xh, z);
        pal = MFMA1632(pa[0], xl, z);
        pal = MFMA1632(pa[1], xh, pal);
    }
    __device__ __forceinline__ void consume_p(int x) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[x][i] = pam[i] + pal[i] * X3_INV_SCALE;
    }
    __device__ __forceinline__ void load_vh(int x) {                                              // msg0: hidden channels 8q .. 8q+7 of component x
#pragma unroll
        for (int j = 0; j < 8; ++j) vraw[j] = VH[(min(8 * vq + j, H0 - 1) * 3 + x) * ETP + ve];    // rows >= H0: any finite value (A is zero there)
    }
    __device__ __forceinline__ void split_vh(int b) {
        const float v0[4] = {vraw[0], vraw[1], vraw[2], vraw[3]}, v1[4] = {vraw[4], vraw[5], vraw[6], vraw[7]};
        split4<X3_PKG>(v0, sh0, sl0, amax);
        split4<X3_PKG>(v1, sh1, sl1, amax);
        bh[b] = cat44(sh0, sh1);
        bl[b] = cat44(sl0, sl1);
        x3_settle(bh[b], bl[b]);
    }
    __device__ __forceinline__ void split_vv(int x) {                                             // B images of vector_down: own groups q | 4 + q
        const float b0[4] = {va[x][0], va[x][1], va[x][2], va[x][3]}, b1[4] = {vb[x][0], vb[x][1], vb[x][2], vb[x][3]};
        split4<X3_PKG>(b0, sh0, sl0, amax);
        split4<X3_PKG>(b1, sh1, sl1, amax);
    }
    __device__ __forceinline__ void pre_x(int x) {
        h8 xh = cat44(sh0, sh1), xl = cat44(sl0, sl1);
        x3_settle(xh, xl);
        f32x4 am = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
        am = MFMA1632(pa[0], xh, am);
        al = MFMA1632(pa[0], xl, al);
        al = MFMA1632(pa[1], xh, al);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[x][i] = am[i] + al[i] * X3_INV_SCALE;
    }
    __device__ __forceinline__ void vhb_x(int x, h8& img) {                                       // [hi(3) 0 | lo'(3) 0] of hidden vectors 3q .. 3q+2
        const float v[4] = {o[x][0], o[x][1], o[x][2], 0.f};
        h4 vh, vl;
        split4<X3_PKG>(v, vh, vl, amax);
        img = cat44(vh, vl);
        x3_settle(img);
    }

    // the common tail: vector_down of GCP2 k from va / vb (stages T0 .. T0+9)
    template <int I, int T0>
    __device__ __forceinline__ void pre_stage() {
        if constexpr (I == T0) { pa[0] = pH[lane]; pa[1] = pL[lane]; split_vv(0); }
        else if constexpr (I == T0 + 1) { pre_x(0); split_vv(1); }
        else if constexpr (I == T0 + 2) { pre_x(1); split_vv(2); }
        else if constexpr (I == T0 + 3) pre_x(2);
        else if constexpr (I == T0 + 4) {       // norms of the hidden vectors (gcpnet.py:442-452), frame scalars q[3j + r] = F[r,:] . u_j (scalarize)
            float f[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) f[r] = FR[r * ETP + ve];
#pragma unroll
            for (int i = 0; i < 3; ++i) ev[i] = fast_sqrt(o[0][i] * o[0][i] + o[1][i] * o[1][i] + o[2][i] * o[2][i] + 1e-8f) + 1e-8f;
#pragma unroll
            for (int r = 0; r < 3; ++r) ev[3 + r] = f[3 * r] * o[0][3] + f[3 * r + 1] * o[1][3] + f[3 * r + 2] * o[2][3];
        } else if constexpr (I == T0 + 5) {       // extended-K group 32 + q of this edge
            const float c0[4] = {ev[0], ev[1], ev[2], ev[3]}, c1[4] = {ev[4], ev[5], 0.f, vq == 3 ? 1.0f : 0.f};
            h4 h0_, l0_, h1_, l1_;
            split4<X3_PKG>(c0, h0_, l0_, amax);
            split4<X3_PKG>(c1, h1_, l1_, amax);
            *(h8*)(XH + ((32 + vq) * ETP + ve) * 16) = cat44(h0_, h1_);
            *(h8*)(XL + ((32 + vq) * ETP + ve) * 16) = cat44(l0_, l1_);
        } else if constexpr (I == T0 + 6) { vhb_x(0, bh[0]); }
        else if constexpr (I == T0 + 7) { vhb_x(1, bh[1]); }
        else if constexpr (I == T0 + 8) {
            vhb_x(2, bh[2]);
            if (vq < 3) {
#pragma unroll
                for (int x = 0; x < 3; ++x) VHB[(x * 3 + vq) * ET + ve] = bh[x];
            }
        }
    }

    template <int I>
    __device__ __forceinline__ void run() {
        if constexpr (FIRST) {        // vector_up of msg0 (hidden vectors: fp32 rows VH[h*3 + x][e] written by P1) ...
            if constexpr (I == 0) {
                if (!preloaded) load_fin();
                g0 = gate_sum(0);
            } else if constexpr (I == 1) {
                g0 = sig4(g0);
                g1 = gate_sum(1);
                load_vh(0);
            } else if constexpr (I == 2) {
                g1 = sig4(g1);
                split_vh(0);
            } else if constexpr (I == 3) { finish_x(0); load_vh(1); }
            else if constexpr (I == 4) { split_vh(0); finish_x(0, 1); load_vh(2); }
            else if constexpr (I == 5) { split_vh(0); finish_x(0, 2); }
            else pre_stage<I, 6>();   // ... then vector_down of the first residual GCP2
        } else if constexpr (PIPE) {
            if constexpr (I == 0) {
                if (!preloaded) load_fin();
                g0 = gate_sum(0);
            } else if constexpr (I == 1) {
                g0 = sig4(g0);
                g1 = gate_sum(1);
#pragma unroll
                for (int x = 0; x < 3; ++x) bh[x] = VHB[(x * 3 + min(vq, 2)) * ET + ve];
            } else if constexpr (I == 2) { g1 = sig4(g1); issue_f(0, 0); }
            else if constexpr (I == 3) { consume_f(0); issue_f(1, 1); }
            else if constexpr (I == 4) { consume_f(1); issue_f(2, 2); }
            else if constexpr (I == 5) { consume_f(2); pa[0] = pH[lane]; pa[1] = pL[lane]; split_vv(0); }
            else if constexpr (I == 6) { issue_p(); split_vv(1); }
            else if constexpr (I == 7) { consume_p(0); issue_p(); split_vv(2); }
            else if constexpr (I == 8) { consume_p(1); issue_p(); }
            else if constexpr (I == 9) consume_p(2);
            else pre_stage<I, 6>();              // (stages 10 .. 14: norms, extended-K group, hidden-vector images: as before)
        } else {
            if constexpr (I == 0) {
                if (!preloaded) load_fin();
                g0 = gate_sum(0);
            } else if constexpr (I == 1) {
                g0 = sig4(g0);
                g1 = gate_sum(1);
#pragma unroll
                for (int x = 0; x < 3; ++x) bh[x] = VHB[(x * 3 + min(vq, 2)) * ET + ve];      // lanes q = 3 read a finite image, their A columns are zero
            } else if constexpr (I == 2) g1 = sig4(g1);
            else if constexpr (I == 3) finish_x(0);
            else if constexpr (I == 4) finish_x(1);
            else if constexpr (I == 5) finish_x(2);
            else pre_stage<I, 6>();
        }
    }
    // vector part of the LAST GCP2 (no GEMM left to hide it in)
    __device__ __forceinline__ void finish_only() {
        if constexpr (PIPE && !FIRST) {
            run<0>(); run<1>(); run<2>(); run<3>(); run<4>();
            consume_f(2);                    // (stage 5 of the pipelined form without the first step of a following pre part)
        } else {
            run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>();
        }
    }
};

// Scalar GEMM of a residual GCP2 with the vector stages in its shadow: k-blocks [0, SPLIT) carry hook(stage r) between their MFMAs
// (vector waves; the others pass a no-op), then a workgroup barrier (the extended-K rows are complete), then k-blocks [SPLIT, KB).
template <int MT, int NT, int PD, int KB, int SPLIT, bool HOOKED, int RO, class Hook, class Tail = NoTail>
__device__ __forceinline__ void tile_gemm_x3s(f32x16 (&am)[MT][NT], f32x16 (&al)[MT][NT], X3Ring<MT, PD>& ring, const WPool& wp, uint32_t wH,
                                              uint32_t wL, const h8* xh8, const h8* xl8, int TP, int lane, Hook&& hook, Tail&& tail = NoTail{}) {
    constexpr int R = PD + 1;
    constexpr int wstride = KB * 64;
    const int boff = (lane >> 5) * TP + (lane & 31);
    const h8* sh = xh8 + boff;
    const h8* sl = xl8 + boff;
    h8 bh[2][NT], bl[2][NT];
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[0][n] = sh[n * 32]; bl[0][n] = sl[n * 32]; }
    auto body = [&](auto rc, auto hooked) {
        constexpr int r = decltype(rc)::value;
        constexpr bool HK = decltype(hooked)::value;
        auto loads = [&] {
#ifndef GCDM_ABL_NOWLOAD                         // (timing ablation only: the GEMM without its weight stream)
            // uniform base + compile-time block offset + lane: no per-block VALU pointer arithmetic.  Issue order: the operand of the
            // block's FIRST MFMA (ah[0]) last, so that one s_waitcnt covers the whole block instead of one per operand
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.alo[(r + PD) % R][m] = wp.template ldk<r + PD>(wL + m * wstride * 16);
#pragma unroll
            for (int m = MT - 1; m >= 0; --m) ring.ah[(r + PD) % R][m] = wp.template ldk<r + PD>(wH + m * wstride * 16);
#endif
        };
        auto breads = [&] {
#ifndef GCDM_ABL_NOB
            if constexpr (r + 1 != SPLIT) {          // the block behind the barrier is read after the barrier
#pragma unroll
                for (int n = 0; n < NT; ++n) { bh[(r + 1) & 1][n] = sh[(r + 1) * 2 * TP + n * 32]; bl[(r + 1) & 1][n] = sl[(r + 1) * 2 * TP + n * 32]; }
            }
#else
            for (int n = 0; n < NT; ++n) { bh[(r + 1) & 1][n] = bh[r & 1][n]; bl[(r + 1) & 1][n] = bl[r & 1][n]; }
#endif
        };
        auto mfmas = [&] {
#if (!defined(GCDM_X3_MFMA_ORDER) || GCDM_X3_MFMA_ORDER >= 1) && !defined(GCDM_ABL_MFMA1)
            // Round 6: the block's six products in the order al0 al1 am0 al0 al1 am1 -- the two visits of an `al` accumulator three MFMAs apart instead of two
            // (am0 am1 al0 al1 al0 al1 before; -DGCDM_X3_MFMA_ORDER=0).  Every accumulator still sees its products in the same order: same bits; with msg0's GEMM
            // (tile_gemm_x3z) in the same order QM9 59 110 -> 58 975 cycles per tile of the fused form (-0.25 %; -0.4 % on a second box), GEOM 58 075 -> 57 845 (-0.4 %)
            // (profiles/r06_ab_log.txt run 13).  The same reorder in the gate contraction costs +0.6 % and stays out; the two other orders with every distance >= 3
            // (am0 al0 al1 am1 al0 al1; al0 am0 al1 al0 am1 al1) are +0.1 / +0.2 % at QM9 (the second -0.3 % at GEOM).  RO: which instantiations take the new order (X3_RO in
            // the kernel) -- the two-launch form of the 8-channel edge width is 0.3 % FASTER in the old one (57 150 against 57 325 cycles), its fused form and both forms of the
            // 16-channel width in the new one (58 755 -> 58 490 two-launch QM9); its fused form is another 0.3 % faster in the third order (RO = 3).
            if constexpr (RO == 1 && MT == 1 && NT == 2) {
                al[0][0] = MFMA16(ring.ah[r % R][0], bl[r & 1][0], r == 0 ? zero : al[0][0]);
                al[0][1] = MFMA16(ring.ah[r % R][0], bl[r & 1][1], r == 0 ? zero : al[0][1]);
                am[0][0] = MFMA16(ring.ah[r % R][0], bh[r & 1][0], r == 0 ? zero : am[0][0]);
                al[0][0] = MFMA16(ring.alo[r % R][0], bh[r & 1][0], al[0][0]);
                al[0][1] = MFMA16(ring.alo[r % R][0], bh[r & 1][1], al[0][1]);
                am[0][1] = MFMA16(ring.ah[r % R][0], bh[r & 1][1], r == 0 ? zero : am[0][1]);
                return;
            } else if constexpr (RO == 2 && MT == 1 && NT == 2) {      // am0 al0 al1 am1 al0 al1
                am[0][0] = MFMA16(ring.ah[r % R][0], bh[r & 1][0], r == 0 ? zero : am[0][0]);
                al[0][0] = MFMA16(ring.ah[r % R][0], bl[r & 1][0], r == 0 ? zero : al[0][0]);
                al[0][1] = MFMA16(ring.ah[r % R][0], bl[r & 1][1], r == 0 ? zero : al[0][1]);
                am[0][1] = MFMA16(ring.ah[r % R][0], bh[r & 1][1], r == 0 ? zero : am[0][1]);
                al[0][0] = MFMA16(ring.alo[r % R][0], bh[r & 1][0], al[0][0]);
                al[0][1] = MFMA16(ring.alo[r % R][0], bh[r & 1][1], al[0][1]);
                return;
            } else if constexpr (RO == 3 && MT == 1 && NT == 2) {      // al0 am0 al1 al0 am1 al1
                al[0][0] = MFMA16(ring.ah[r % R][0], bl[r & 1][0], r == 0 ? zero : al[0][0]);
                am[0][0] = MFMA16(ring.ah[r % R][0], bh[r & 1][0], r == 0 ? zero : am[0][0]);
                al[0][1] = MFMA16(ring.ah[r % R][0], bl[r & 1][1], r == 0 ? zero : al[0][1]);
                al[0][0] = MFMA16(ring.alo[r % R][0], bh[r & 1][0], al[0][0]);
                am[0][1] = MFMA16(ring.ah[r % R][0], bh[r & 1][1], r == 0 ? zero : am[0][1]);
                al[0][1] = MFMA16(ring.alo[r % R][0], bh[r & 1][1], al[0][1]);
                return;
            }
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) am[m][n] = MFMA16(ring.ah[r % R][m], bh[r & 1][n], r == 0 ? zero : am[m][n]);
#ifndef GCDM_ABL_MFMA1
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.ah[r % R][m], bl[r & 1][n], r == 0 ? zero : al[m][n]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) al[m][n] = MFMA16(ring.alo[r % R][m], bh[r & 1][n], al[m][n]);
#else
            if (r == 0) { for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) al[m][n] = zero; }
#endif
        };
        if constexpr (!HK) {
            // the next blocks' operand requests ride BETWEEN this block's MFMAs (one per MFMA: B reads first, their latency is the shorter
            // one to cover), issued while the matrix pipe works on the MFMA in front of them, instead of in a burst ahead of the block
            static_assert(PD >= 2, "interleaved requests are waited for one block later: two blocks of prefetch distance");
            x3_wait_block<2 * MT * (PD - 1)>();
            __builtin_amdgcn_sched_barrier(0);
            mfmas();
            breads();
            loads();
#pragma unroll
            for (int i = 0; i < 3 * MT * NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 2 * NT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                else if (i < 2 * NT + 2 * MT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            loads();
            breads();
            __builtin_amdgcn_sched_barrier(0);
            x3_wait_block<2 * MT * PD>();
            __builtin_amdgcn_sched_barrier(0);
            mfmas();
            if constexpr (HK) {
                hook(rc);
#pragma unroll
                for (int i = 0; i < 3 * MT * NT; ++i) {      // one MFMA, then up to X3_VEC_PER_MFMA other instructions of the stage, ... (12 groups of 4 / 6, 9 of 5,
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     //  or no pattern at all: +2.0 / +0.7 / +1.3 / +0.2 % tile cycles, round 4)
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x080 | 0x400 | 0x020, X3_VEC_PER_MFMA, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_for<0, SPLIT>([&](auto rc) { body(rc, std::integral_constant<bool, HOOKED>{}); });
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n) { bh[SPLIT & 1][n] = sh[SPLIT * 2 * TP + n * 32]; bl[SPLIT & 1][n] = sl[SPLIT * 2 * TP + n * 32]; }
    if constexpr (std::is_same<std::decay_t<Tail>, NoTail>::value || NT != 2 || MT != 1 || KB - SPLIT != 2) {
        static_for<SPLIT, KB>([&](auto rc) { body(rc, std::false_type{}); });
    } else {
        x3_tail_skew<PD, SPLIT>(am, al, ring, bh, bl, sh, sl, TP, tail);        // (see x3_tail_skew)
    }
}

struct AblFmaFill {                   // (GCDM_ABL_VECFMA: 24 independent FMAs per hooked stage)
    float x[24];
    __device__ __forceinline__ void run() {
#pragma unroll
        for (int i = 0; i < 24; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(0.0003f));
    }
};

struct EdgeMsgX3Args {
    X3Const x3c;                      // MUST stay the first member (X3_KARG)
    EdgeMsgArgs base;                 // everything the fp32 kernel takes (tables, biases, vector weights, attention)
    const h8* w0H; const h8* w0L; int KB0;          // msg0 per-edge part, packed [8][KB0][64] x 8 f16
    const h8* wg0H; const h8* wg0L;                 // msg0 gate, packed [8][2][64]
    const h8* wH[3]; const h8* wL[3]; int KB;       // msg1..3 scalar_out, packed [8][KB][64]
    const h8* wgH[3]; const h8* wgL[3];
    uint32_t* flags_dev;                            // bit GCDM_FLAG_F16_RANGE
    // vector path (16x16x32 A operands, [64 lanes] x 8 f16 each; packing: gcdm_api.hip pack_vec_*)
    const h8* vpH[3]; const h8* vpL[3];             // msg1..3 [W_down; W_frames] (11 x 32 -> 16 x 32), K permuted to the VV4 lane ownership
    const h8* vf1[3]; const h8* vf2[3];             // msg1..3 vector_up [32 x 8] as two M-tiles: A1 = [W_hi | 0], A2 = [W_lo' | W_hi]
    const h8* vf0H; const h8* vf0L;                 // msg0 vector_up [32 x H0] as two M-tiles, K = hidden channel
    const h8* wbeH; const h8* wbeL;                 // msg0: the edge block of [W_down; W_frames] ((H0 + 3) x Ve -> 32 x 16) as ONE A operand [64 lanes] x 8 f16
    int wg_stride;                                  // persistent schedule: workgroups per XCD (gridDim.x / 8; >= tiles per XCD when every workgroup takes one tile)
    const float* wax;                               // scalar_message_attention weights / c
    const void* wpool; uint32_t wpool_bytes;        // the whole weight pool (every packed array above lies inside): base of the buffer-load stream
    const void* wspool; uint32_t wspool_bytes;      // the workspace pool (EP4, AL, U, FR, PQ4, VDI, VDJ lie inside)
};

#define GCDM_FLAG_F16_RANGE_BIT 8u

// ET = 64: 8 waves, wave w owns M-tile w x both N-tiles (every weight byte is loaded once per CU and tile).
// ET = 32: 4 waves, wave w owns M-tiles 2w, 2w+1 x one N-tile; half the LDS.  Designed for two workgroups per CU running out of phase (one in its GEMM
//          while the other is in a VALU phase) at the price of streaming the weights twice per 64 edges -- but since the attention weights (round 3) and the
//          unit vectors (round 5) are staged in LDS its footprint is 83 260 B, 1.3 KB more than half of the CU's 160 KB: the launch's 2 x CUs workgroups run
//          in two rounds of ONE per CU (QM9: 0.89 ms per launch against 0.72 ms with 64-edge tiles).  Kept as an option (edge_tile = 32) for A/B runs and
//          for the determinism tests, which cover both tile sizes; not a production configuration.
// TAIL ROLE (round 6).  `TR` is a policy: NoTailRole (below) = the kernel as it was; NodeTailRole (gcdm_layer_x3.hip.h) = the layer's NODE tiles run as a tail
// role of the same persistent workgroups: a workgroup that has walked its (statically scheduled) edge tiles takes node tiles from its XCD's queue, each gated by
// a readiness counter that the edge tiles covering the node tile's rows bump when their AGG / PART rows are in the L2.  The hooks the policy provides (static):
//     TR::Args            kernel-argument struct whose FIRST member `e` is the EdgeMsgX3Args (X3_KARG and the opaque kernel-argument copy rely on offset 0)
//     TR::arrived(...)    kernel start, one thread
//     TR::published(...)  behind the first barrier of the NEXT tile (every wave has waited for its stores): the previous tile's rows are visible -> bump
//     TR::tail(...)       behind the persistent loop: publish the last tile, then the node role
// The hooks inside the tile loop are deliberately minimal (three scalars carried, one thread's atomics per tile: +1.3 % cycles per tile).  A unified, dynamically
// scheduled work queue of edge AND node items was built and measured in round 6 (docs/r06_dynamic_queue/): any code that puts a second 250-register body or a stack
// into this function costs the GEMM phases their schedule (the scheduler reverts regions whose pressure exceeds the budget once a register is reserved for spills):
// +11 % cycles per edge tile, more than the balance returns.
struct NoTailRole {
    static constexpr bool ON = false;
    struct Args { EdgeMsgX3Args e; };
};

template <int SE, int VE, int ET, class TR = NoTailRole>
__global__ __launch_bounds__(ET * 8) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_edge_msg_x3(typename TR::Args lx) {
    constexpr int NW = ET / 8, MT = 8 / NW, NT = ET / 32;     // waves, M-tiles and N-tiles per wave
    const EdgeMsgX3Args& ax0 = lx.e;
    const EdgeMsgArgs& a0 = ax0.base;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Geo = EdgeGeo<ET>;
    constexpr int ETP = Geo::TP, EK_THREADS = Geo::THREADS, PARTS = Geo::PARTS;
    // MFMA order inside a k-block of the tile GEMMs (tile_gemm_x3s: 0 = am0 am1 al0 al1 al0 al1, 1 = al0 al1 am0 al0 al1 am1, 3 = al0 am0 al1 al0 am1 al1), measured per
    // instantiation: 16-channel edge width 1 in both forms; 8-channel: two-launch form 0 (57 130 cycles; 1: 57 325, 2: 57 250, 3: 57 220), fused form 3 (57 690; 1: 57 880, 0: 58 030)
    constexpr int X3_RO = VE == 16 ? 1 : (TR::ON ? 3 : 0);
    constexpr int X3_GROUPS8 = 36;                      // 32 state + 1 norm + 2 frame scalars + 1 pad (K' = 288)
    char* XH = smem + Geo::OFF_XS;                      // [36][65] x 16 B : hi images
    char* XL = XH + X3_GROUPS8 * ETP * 16;              // [36][65] x 16 B : lo' images
    static_assert(2 * X3_GROUPS8 * ETP * 16 <= Geo::OFF_VV, "XH8/XL8 must fit the fp32 XS4 region");
    v4f* XS4 = (v4f*)(smem + Geo::OFF_XS);              // fp32 alias, written after the last GEMM: attention-weighted messages for the segment sums
    float* VH = (float*)(smem + Geo::OFF_VH);
    float* PG = (float*)(smem + Geo::OFF_PG);
    float* FR = (float*)(smem + Geo::OFF_FR);
    int2* m_rec = (int2*)(smem + Geo::OFF_META);        // [segment] {node, first edge | end << 8 | whole row << 16}: one LDS read per work item of the aggregation
    int* m_seg = (int*)(m_rec + ET);                    // (the fp32 kernel's segment-start table and attention table follow: unused here, they only
    float* m_att = (float*)(m_seg + ET + 2);            //  place m_misc where the shared geometry has it)
    int* m_misc = (int*)(m_att + ET);

    constexpr int H0 = (2 * GCDM_V + VE) / 4;
    constexpr int SEG = SE / 4;                          // float4 groups of e' in global memory
    constexpr int N8 = SE / 8;                           // first 8-group of the norm rows in msg0
    constexpr int QPOS = x3_msg0_qpos(SE, H0);           // first slot of the 9 frame scalars
    constexpr int KB0C = x3_msg0_kb(SE, H0);             // k-blocks of the msg0 per-edge part (host: gcdm_api.hip, same functions); msg1..3: 18
    // who am I: recomputed from an opaque copy of the thread index at the top of every tile, so that the compiler does not hoist the dozens of
    // lane-dependent LDS / buffer offsets of the tile body out of the persistent loop and keep them in registers across the GEMM phases
    struct Who {
        int tid, lane, wave, e, part, mt0;
        bool need_fr;
    };
    auto who_am_i = [&](int tid_) {
        Who w;
        w.tid = tid_; w.lane = tid_ & 63;
        w.wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
        w.e = (ET == 64) ? w.lane : (tid_ & (ET - 1));
        w.part = (ET == 64) ? w.wave : (tid_ / ET);
        w.mt0 = (8 / (ET / 8)) * w.wave;
        // Round 5: the frame rows and the unit vector of an edge are loaded ONCE per workgroup, by the thread that stages them in LDS (the last part); the
        // threads whose rows of the pre-phase are the three frame vectors, and every thread's u, read them back from LDS behind the pre-phase barrier.
        // Rounds 1-4 had 4 of the 8 waves load the same nine frame rows and all 8 the same three u rows: 60 wave-level loads per tile instead of 12, each
        // one or two cache-line misses to HBM (non-temporal stream) in flight together with the tile's other 32 -- and with the weight stream of the
        // GEMM phases queued behind them: removing EITHER the FR or the AL loads alone returned 1.3-1.7 k of the tile's 61.2 k cycles, removing all
        // constants no more than that (profiles/r05_ab_log.txt, runs 11-13): the cost was the number of misses in flight, not the bytes.
        w.need_fr = w.part == PARTS - 1;
        return w;
    };
    const Who w0 = who_am_i(threadIdx.x);
    const int E = a0.E, N = a0.N;
    // Persistent workgroups (one per CU at ET = 64): workgroup b runs on XCD b % 8 (round-robin dispatch) and walks tiles
    //     start(x) + (b >> 3) + j * wg_stride,   j = 0, 1, ...
    // of the x-th contiguous eighth of the tile list, so the node rows its tiles gather (PQ4 / VDI / VDJ of consecutive molecules) stay in
    // that XCD's own L2 (round 2: -0.5 % / -0.2 %).  What the loop buys is the NEXT tile's operands: its index words are requested behind the
    // last GEMM and its per-edge constants and gathered node rows behind the last state image, into registers that are free there (the
    // accumulators and the state are dead), so the two dependent round trips that opened every tile (index -> gather -> use, ~4 k cycles
    // with nothing else to issue: one workgroup per CU) run under the attention + aggregation phase of the tile before.
    const int G_ = (E + ET - 1) / ET, xcd_ = blockIdx.x & 7, base_ = G_ >> 3, rem_ = G_ & 7;
    const int cnt_ = base_ + (xcd_ < rem_ ? 1 : 0), start_ = xcd_ * base_ + min(xcd_, rem_), stride_ = ax0.wg_stride;
    int it_ = blockIdx.x >> 3;
    constexpr int PD = X3_PD;
    X3Ring<MT, PD> ring;
    const WPool wp = make_wpool(ax0.wpool, ax0.wpool_bytes, w0.lane);
    // per-edge constants (streamed from HBM, independent of the edge list) and gathered node rows.  Buffer loads: the per-lane offset is the
    // edge (node) index, array base and row stride are scalars -- no 64-bit VALU address arithmetic per load
    const BufView ws = make_view(ax0.wspool, ax0.wspool_bytes);
    const BufView wv = make_view(ax0.wpool, ax0.wpool_bytes);
#ifdef GCDM_X3_SAUX
    constexpr int SAUX = GCDM_X3_SAUX;
#else
    constexpr int SAUX = SE == 64 ? 2 : 0;               // cache policy of the streamed per-edge constants (BufView::ld1s)
#endif
    constexpr int EPN = (SE / 4) / PARTS;                // e' float4 groups per thread (QM9 2, GEOM: parts 0..3 one each)
    constexpr int EPN1 = EPN > 0 ? EPN : 1;
    const uint32_t rowE = (uint32_t)E * 4u, rowN = (uint32_t)N * 4u;
    constexpr int ROWS0 = H0 + 3, NH0 = (ROWS0 + PARTS - 1) / PARTS;
    // beta products of the pre-phase: on the matrix pipe for the 16-channel edge width (QM9: -1.8 % tile cycles), the round-2 VALU form
    // (every thread loads the edge's alpha and its rows of W_e) for the 8-channel one, where half of the MFMA's K would be padding (GEOM: +0.5 %)
    constexpr bool BETA_MFMA = VE == 16;
    struct TileIdx {
        int ni, nj;                      // this thread's edge (pre-phase layout: edge = lane, part = wave)
        int ri[NT], cj[NT];              // this lane's GEMM-layout edges (32 n + (lane & 31))
    };
    struct TileIn {
        float fr[9];
        v4f epv[EPN1];
        float al[BETA_MFMA ? 1 : VE];
        float av[8];                     // BETA_MFMA: alpha rows of the contracting waves
        float u0, u1, u2;
        v4f pqi[MT][NT][4], pqj[MT][NT][4];          // node-level halves of msg0 (PQ4 rows), consumed after P1
        float gi[NH0][3], gj[NH0][3];    // vector_down halves of the end nodes (VDI / VDJ rows)
        int ncnt;                        // wave 0: number of edges of the row of this thread's edge (is a segment of the tile a whole row?)
    };
    auto load_idx = [&](const EdgeMsgArgs& a, const Who& w, int tile) {
        TileIdx ix;
        const int e0 = tile * ET, eid = min(e0 + w.e, E - 1);
        ix.ni = a.EROW[eid]; ix.nj = a.ECOL[eid];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int eg = min(e0 + 32 * n + (w.lane & 31), E - 1);
            ix.ri[n] = a.EROW[eg]; ix.cj[n] = a.ECOL[eg];
        }
        return ix;
    };
    auto load_const = [&](auto pc, const EdgeMsgArgs& a, const Who& w, int tile, TileIn& in) {       // independent of the edge list; three bursts (P = 0, 1, 2; P < 0: all)
        constexpr int P = decltype(pc)::value;
#ifdef GCDM_ABL_NOCONST      // (timing ablation, round 5: the tile WITHOUT [some of] its 368 B per edge of streamed constants -- the ceiling of what recomputing them per layer could return)
        // GCDM_ABL_NOCONST = bit mask of the streams replaced by constants: 1 FR, 2 EP4, 4 AL (alpha), 8 U
        constexpr int ABL = GCDM_ABL_NOCONST;
#else
        constexpr int ABL = 0;
#endif
        if constexpr ((ABL & 1) != 0 && (P < 0 || P == 0)) { for (int r = 0; r < 9; ++r) in.fr[r] = 0.01f * (r + 1); }
        if constexpr ((ABL & 2) != 0 && (P < 0 || P == 1)) { for (int i = 0; i < EPN1; ++i) in.epv[i] = (v4f){0.1f, -0.2f, 0.3f, 0.05f}; }
        if constexpr ((ABL & 4) != 0 && (P < 0 || P == 1)) { for (int j = 0; j < 8; ++j) in.av[j] = 0.02f * j; }
        if constexpr ((ABL & 4) != 0 && (P < 0 || P == 2)) { for (int c = 0; c < (BETA_MFMA ? 1 : VE); ++c) in.al[c] = 0.03f; }
        if constexpr ((ABL & 8) != 0 && (P < 0 || P == 2)) { in.u0 = 0.6f; in.u1 = 0.0f; in.u2 = 0.8f; }
        const int e0 = tile * ET, eid = min(e0 + w.e, E - 1);
        const uint32_t ve4 = (uint32_t)eid * 4u, ve16 = (uint32_t)eid * 16u;
#ifndef GCDM_X3_AL_AUX
#define GCDM_X3_AL_AUX SAUX
#endif
#ifndef GCDM_X3_AV_BURST
#define GCDM_X3_AV_BURST 1
#endif
        if constexpr ((ABL & 4) == 0 && (P < 0 || P == GCDM_X3_AV_BURST) && BETA_MFMA && GCDM_X3_AV_BURST == 0) {
            if (w.wave >= ET / 8 - ET / 32) {
                const uint32_t eg4 = (uint32_t)min(e0 + 32 * (w.wave - (ET / 8 - ET / 32)) + (w.lane & 31), E - 1) * 4u + (uint32_t)(8 * (w.lane >> 5)) * rowE, o = ws.off(a.AL);
#pragma unroll
                for (int j = 0; j < 8; ++j) in.av[j] = ws.template ld1s<GCDM_X3_AL_AUX>(eg4, o + (uint32_t)j * rowE);
            }
        }
        if constexpr ((ABL & 1) == 0 && (P < 0 || P == 0)) {
            const uint32_t o = ws.off(a.FR);
#pragma unroll
            for (int r = 0; r < 9; ++r) in.fr[r] = w.need_fr ? ws.template ld1s<SAUX>(ve4, o + r * rowE) : 0.f;
        }
        if constexpr ((ABL & 2) == 0 && (P < 0 || P == 1)) {
            const uint32_t o = ws.off(a.EP4);
#pragma unroll
            for (int i = 0; i < EPN1; ++i)
                in.epv[i] = ws.template ld4s<SAUX>(ve16 + (uint32_t)min(w.part + PARTS * i, SE / 4 - 1) * (rowE * 4u), o);          // the group index depends on the lane's part
        }
        if constexpr ((ABL & 4) == 0 && (P < 0 || P == 2) && !BETA_MFMA) {
            const uint32_t o = ws.off(a.AL);
#pragma unroll
            for (int c = 0; c < VE; ++c) in.al[c] = ws.template ld1s<SAUX>(ve4, o + c * rowE);
        }
        if constexpr ((ABL & 4) == 0 && (P < 0 || P == 1) && BETA_MFMA && GCDM_X3_AV_BURST != 0) {
            if (w.wave >= ET / 8 - ET / 32) {         // the LAST waves contract beta: they are the first to leave the previous tile's segment sums
                // (the half of K this lane holds rides in the per-lane offset: a lane-dependent scalar offset would cost a waterfall loop per load)
                const uint32_t eg4 = (uint32_t)min(e0 + 32 * (w.wave - (ET / 8 - ET / 32)) + (w.lane & 31), E - 1) * 4u + (uint32_t)(8 * (w.lane >> 5)) * rowE, o = ws.off(a.AL);
#pragma unroll
                for (int j = 0; j < 8; ++j) in.av[j] = ws.template ld1s<GCDM_X3_AL_AUX>(eg4, o + (uint32_t)j * rowE);
            }
        }
        if constexpr ((ABL & 8) == 0 && (P < 0 || P == 2)) {
            const uint32_t oU = ws.off(a.U);
            if (w.need_fr) { in.u0 = ws.template ld1s<SAUX>(ve4, oU); in.u1 = ws.template ld1s<SAUX>(ve4, oU + rowE); in.u2 = ws.template ld1s<SAUX>(ve4, oU + 2 * rowE); }
            else { in.u0 = 0.f; in.u1 = 0.f; in.u2 = 0.f; }
        }
    };
    // node rows (need the index words), in GCH chunks: the 16 PQ4 rows (16 B per lane each: 16 clk of the L1 path per wave instruction) and the
    // 18 VDI / VDJ words go through the texture-address path at 64 B/clk/CU -- ~2.6 k cycles per tile, which a wave that issues them back to
    // back spends waiting at the issue queue (measured: the phase they are issued in grows by just that).  So they are dealt out, a few at a
    // time, over phases of LDS + VALU work
    constexpr int GCH = GCDM_SG / PARTS, NPQ = NT * MT * 4 * 2, NVD = NH0 * 3 * 2;
    // PQ4 rows of the CURRENT tile (its index words were prefetched): dealt out over the steps of the pre-phase, consumed right behind it
    auto load_pq_part = [&](auto cc, const EdgeMsgArgs& a, const Who& w, const TileIdx& ix, TileIn& in) {
        constexpr int C = decltype(cc)::value;
        const uint32_t oP = ws.off(a.PQ4), rowP = (uint32_t)N * 16u;
        static_for<C * NPQ / GCH, (C + 1) * NPQ / GCH>([&](auto fc) {
            constexpr int f = decltype(fc)::value, which = f & 1, q = (f >> 1) & 3, m = (f >> 3) % MT, n = (f >> 3) / MT;
            const int g = 8 * (w.mt0 + m) + 2 * q;
            const uint32_t v = ((uint32_t)(w.lane >> 5) * N + (which ? ix.cj[n] : ix.ri[n])) * 16u;     // group 2q + half: the half rides in the lane offset
            if (which) in.pqj[m][n][q] = ws.ld4(v, oP + (uint32_t)(64 + g) * rowP);
            else in.pqi[m][n][q] = ws.ld4(v, oP + (uint32_t)g * rowP);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // VDI / VDJ words of the NEXT tile: over the steps of the attention dot product
    auto load_gather_part = [&](auto cc, const EdgeMsgArgs& a, const Who& w, const TileIdx& ix, TileIn& in) {
        constexpr int C = decltype(cc)::value;
        const uint32_t oI = ws.off(a.VDI), oJ = ws.off(a.VDJ);
        static_for<C * NVD / GCH, (C + 1) * NVD / GCH>([&](auto vc) {
            constexpr int f = decltype(vc)::value, which = f & 1, x = (f >> 1) % 3, i = (f >> 1) / 3;
            const int hh = min(w.part + PARTS * i, ROWS0 - 1);                                      // the row depends on the lane's part
#ifdef GCDM_ABL_NOGATHER
            if (which) in.gj[i][x] = 0.2f; else in.gi[i][x] = 0.1f * x;
#else
            const uint32_t v = ((uint32_t)(hh * 3) * N + (which ? ix.nj : ix.ni)) * 4u;
            if (which) in.gj[i][x] = ws.ld1(v, oJ + x * rowN);
            else in.gi[i][x] = ws.ld1(v, oI + x * rowN);
#endif
        });
        if (C == GCH - 1) in.ncnt = w.wave == 0 ? a.NCNT[ix.ni] : 0;
    };
    v4f* WAX4 = (v4f*)(smem + Geo::OFF_WAX);
    if (w0.tid < GCDM_SG) WAX4[w0.tid] = *(const v4f*)(ax0.wax + 4 * w0.tid);       // visible after the first tile's barriers
    [[maybe_unused]] int tail_prev_first = 0, tail_prev_last = 0;      // row nodes of the tile whose publication is pending (TR::ON)
    [[maybe_unused]] bool tail_have_prev = false;
    [[maybe_unused]] int tail_rel_end = 0;                              // rows below this node are (also) consumed on the previous XCD: released (NodeTailRole)
    if constexpr (TR::ON) { if (w0.tid == 0) TR::arrived(lx, smem, xcd_); tail_rel_end = TR::rel_node_end(lx, xcd_); }
    TileIdx ix = load_idx(a0, w0, start_ + it_);
    TileIn in;
    load_const(std::integral_constant<int, -1>{}, a0, w0, start_ + it_, in);
    static_for<0, GCH>([&](auto cc) { load_gather_part(cc, a0, w0, ix, in); });
    for (;;) {
    // ... and the kernel arguments through an opaque copy of the kernel-argument pointer: their scalar loads stay where they are used instead
    // of being hoisted out of the loop into SGPRs that then spill to VGPR lanes inside the GEMM loops
    typedef const EdgeMsgX3Args __attribute__((address_space(4)))* karg_ptr;
    karg_ptr kp_ = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp_));
    const EdgeMsgX3Args& ax = *(const EdgeMsgX3Args*)kp_;
    const EdgeMsgArgs& a = ax.base;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const Who me = who_am_i(tid_);
    const int tid = me.tid, lane = me.lane, wave = me.wave, e = me.e, part = me.part, mt0 = me.mt0;
    const uint32_t o0H = wp.off(ax.w0H + (size_t)mt0 * KB0C * 64), o0L = wp.off(ax.w0L + (size_t)mt0 * KB0C * 64);
    const int prof_tile = start_ + it_;
    const int e0 = prof_tile * ET;
    const int nvalid = min(ET, E - e0);
    const int eid = min(e0 + e, E - 1);
    const int ni = ix.ni;
    [[maybe_unused]] const uint64_t t_start = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    bool over = false;
    x3_prefetch_b<MT, PD>(ring, wp, o0H, o0L, KB0C);   // flies during P1
    float* US = (float*)(smem + Geo::OFF_US);            // [3][ETP]: the edges' unit vectors (staged by the last part, read by everyone behind the barrier)

    // ---- P1: msg0 pre-phase ---------------------------------------------------------------------------------------------
#ifndef GCDM_ABL_NOP1
    {
        float* BETA = PG;                                   // [ET][33]: BETA[e * 33 + h]
        load_pq_part(std::integral_constant<int, 0>{}, a, me, ix, in);
        float* BETA2 = PG + ET * 33;                        // self-conditioning: the second rank (BL)
        if constexpr (BETA_MFMA) {
        // beta[h][e] = sum_c W_e[h][c] alpha_c[e] -- the edge block of vector_down / vector_down_frames applied to the rank-1 embedded edge
        // vectors (gcpnet.py:442-459) -- is a [23 x 16] . [16 x ET] contraction: ONE wave per 32 edges evaluates it on the matrix pipe (3 MFMAs,
        // 8 alpha loads per lane) and hands it over in LDS, instead of 16 alpha loads + 48 weight loads + 48 FMAs in each of the 8 threads
        // of an edge.  BETA aliases the gate-partial buffer PG, which is first written after the msg0 GEMM.
        if (wave >= NW - ET / 32) {
            const int n_ = lane & 31, kh_ = lane >> 5, wb_ = wave - (NW - ET / 32);
            const uint32_t eg4 = (uint32_t)min(e0 + 32 * wb_ + n_, E - 1) * 4u;
            const h8 aH = wp.ld(wp.off(ax.wbeH)), aL = wp.ld(wp.off(ax.wbeL));
            const f32x16 zero_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            auto contract = [&](const float (&av)[8], float* DST) {
                h8 bh, bl;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    h2 hi, lo;
                    split16x2(av[j], av[j + 1], hi, lo);
                    bh[j] = hi[0]; bh[j + 1] = hi[1];
                    bl[j] = lo[0]; bl[j + 1] = lo[1];
                }
                x3_settle(bh, bl);
                f32x16 am_ = MFMA16(aH, bh, zero_);
                f32x16 al_ = MFMA16(aH, bl, zero_);
                al_ = MFMA16(aL, bh, al_);
#pragma unroll
                for (int r = 0; r < 16; ++r) DST[(32 * wb_ + n_) * 33 + (r & 3) + 8 * (r >> 2) + 4 * kh_] = am_[r] + al_[r] * X3_INV_SCALE;
            };
            contract(in.av, BETA);
            if (a.BL) {                  // self-conditioning: the second rank, loaded here (not on the production path)
                const uint32_t o = ws.off(a.BL), eg4k = eg4 + (uint32_t)(8 * kh_) * rowE;
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) bv[j] = ws.ld1(eg4k, o + (uint32_t)j * rowE);
                contract(bv, BETA2);
            }
        }
        }
        if (part == PARTS - 1) {
#pragma unroll
            for (int r = 0; r < 9; ++r) FR[r * ETP + e] = in.fr[r];
            US[e] = in.u0; US[ETP + e] = in.u1; US[2 * ETP + e] = in.u2;
        }
        load_pq_part(std::integral_constant<int, 1>{}, a, me, ix, in);
        load_pq_part(std::integral_constant<int, 2>{}, a, me, ix, in);
        load_pq_part(std::integral_constant<int, 3>{}, a, me, ix, in);
        float beta[NH0], beta2[NH0];
        [[maybe_unused]] const uint32_t oW = wv.off(a.wddE);
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            [[maybe_unused]] const int hh = min(part + PARTS * i, ROWS0 - 1);
            if constexpr (!BETA_MFMA) {
                const uint32_t vW = (uint32_t)(hh * VE) * 4u;
                float bsum = 0.f;
#ifdef GCDM_ABL_NOBETA
                bsum = in.al[0];
#else
#pragma unroll
                for (int c = 0; c < VE; ++c) bsum += wv.ld1(vW, oW + c * 4) * in.al[c];
#endif
                beta[i] = bsum;
            }
            beta2[i] = 0.f;
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (a.BL) {                      // self-conditioning: rank-2 embedded edge vectors, xi'_c = AL_c u + BL_c u_sc (uniform branch)
            s0 = a.USC[eid]; s1 = a.USC[(size_t)E + eid]; s2 = a.USC[2 * (size_t)E + eid];
        }
        if constexpr (!BETA_MFMA) {
        if (a.BL) {
            float bl[VE];
#pragma unroll
            for (int c = 0; c < VE; ++c) bl[c] = a.BL[(size_t)c * E + eid];
#pragma unroll
            for (int i = 0; i < NH0; ++i) {
                const float* w = a.wddE + min(part + PARTS * i, ROWS0 - 1) * VE;
                float bsum = 0.f;
#pragma unroll
                for (int c = 0; c < VE; ++c) bsum += w[c] * bl[c];
                beta2[i] = bsum;
            }
        }
        __syncthreads();                 // the END-OF-TILE barrier of the persistent loop, behind the register-only beta sums (see the other branch)
        } else {
        // BETA complete (the gathers above are in flight across it).  It is also the END-OF-TILE barrier of the persistent loop: everything this
        // tile has written so far (BETA in the gate-partial buffer, the frame rows) goes to LDS the previous tile's segment sums do not read, so the
        // waves that finished those early start here instead of waiting; the images (which alias the fp32 image the sums read) and the segment
        // table are written behind it
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            const int hh = min(part + PARTS * i, ROWS0 - 1);
            beta[i] = BETA[e * 33 + hh];
            if (a.BL) beta2[i] = BETA2[e * 33 + hh];
        }
        }
        const float u0 = US[e], u1 = US[ETP + e], u2 = US[2 * ETP + e];
        if (wave == 0) {
            const bool own = lane < ET;
            // (ds_bpermute on the tile's own lane index: __shfl_up computes the lane id with v_mbcnt, which the compiler hoists out of the persistent loop and --
            //  at 256 VGPRs -- SPILLS; each scratch reload is then the youngest load in flight and waits for every prefetched row of the next tile, round 5)
            const int prev = __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, ni);      // lane 0's value is not used (e == 0)
            const bool start = own && (e < nvalid) && (e == 0 || prev != ni);
            const unsigned long long mask = __ballot(start);
            const int sid = __popcll(mask & ((2ull << lane) - 1ull)) - 1;
            if (start) {
                const unsigned long long rest = lane < 63 ? mask >> (lane + 1) : 0ull;
                const int next = rest ? lane + 1 + __builtin_ctzll(rest) : nvalid;
                // (whole: the segment is the node's whole row; else a partial for the cut-row fix-up)
                m_rec[sid] = make_int2(ni, e | (next << 8) | (((next - e) == in.ncnt ? 1 : 0) << 16));
            }
            if (lane == 0) m_misc[0] = __popcll(mask);
        }
#pragma unroll
        for (int i = 0; i < EPN1; ++i) {       // e' (fp32 in HBM) -> hi / lo' halves of an 8-group
            const int g = part + PARTS * i;
            if (g >= SEG) break;
            const v4f v = in.epv[i];
            h4 vh, vl;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                h2 hi, lo;
                split16x2(v[t], v[t + 1], hi, lo);
                vh[t] = hi[0]; vh[t + 1] = hi[1];
                vl[t] = lo[0]; vl[t + 1] = lo[1];
                over |= X3_OVER(fmaxf(fabsf(v[t]), fabsf(v[t + 1])) > X3_RANGE);
            }
            const int off = ((g >> 1) * ETP + e) * 16 + 8 * (g & 1);
            *(h4*)(XH + off) = vh;
            *(h4*)(XL + off) = vl;
        }
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            if (i == 0) load_pq_part(std::integral_constant<int, 4>{}, a, me, ix, in);
            if (i == 1) load_pq_part(std::integral_constant<int, 5>{}, a, me, ix, in);
            if (i == 2) load_pq_part(std::integral_constant<int, 6>{}, a, me, ix, in);
            const int hh = part + PARTS * i;
            const float vx = in.gi[i][0] + beta[i] * u0 + beta2[i] * s0 + in.gj[i][0];
            const float vy = in.gi[i][1] + beta[i] * u1 + beta2[i] * s1 + in.gj[i][1];
            const float vz = in.gi[i][2] + beta[i] * u2 + beta2[i] * s2 + in.gj[i][2];
#ifdef GCDM_ABL_NOP1W
            if (vx + vy + vz == 123.456f) VH[e] = vx;
            continue;
#endif
            if (hh < H0) {
                over |= put16(XH, XL, ETP, N8 + (hh >> 3), hh & 7, e, fast_sqrt(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f);
                VH[(hh * 3 + 0) * ETP + e] = vx;
                VH[(hh * 3 + 1) * ETP + e] = vy;
                VH[(hh * 3 + 2) * ETP + e] = vz;
            } else if (hh < ROWS0) {
                const int k = hh - H0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int idx = 3 * k + r;
                    const float f0 = FR[(3 * r) * ETP + e], f1 = FR[(3 * r + 1) * ETP + e], f2 = FR[(3 * r + 2) * ETP + e];
                    over |= put16(XH, XL, ETP, (QPOS + idx) >> 3, (QPOS + idx) & 7, e, f0 * vx + f1 * vy + f2 * vz);
                }
            }
        }
        static_for<(NH0 < 3 ? 4 + NH0 : 7), GCH>([&](auto cc) { load_pq_part(cc, a, me, ix, in); });
        if (part == PARTS - 1) {
            // the slots no row is written to: between the norms and the frame scalars (padded layout only), and behind the last frame scalar
            for (int pos = SE + H0; pos < QPOS; ++pos) put16(XH, XL, ETP, pos >> 3, pos & 7, e, 0.f);
            for (int pos = QPOS + 9; pos < 16 * KB0C; ++pos) put16(XH, XL, ETP, pos >> 3, pos & 7, e, 0.f);
        }
    }
#endif
    STAMP(1);
    if constexpr (TR::ON) x3_wait_block<0>();      // every store of the PREVIOUS tile's segment sums has been acknowledged by the L2 (the loads still in flight -- msg0's
                                                   // first weight blocks, the PQ rows -- are consumed right behind the barrier anyway)
    __syncthreads();
    if constexpr (TR::ON) {
        if (tail_have_prev && tid == 0) TR::published(lx, tail_prev_first, tail_prev_last, tail_rel_end);
        // this tile's first / last row (uniform: lane e of every wave holds edge e0 + e, clamped to the last edge) for the publication one tile later
        tail_prev_first = __builtin_amdgcn_readlane(ni, 0);
        tail_prev_last = __builtin_amdgcn_readlane(ni, 63);
        tail_have_prev = true;
    }
    STAMP(2);

    const int half = lane >> 5, l31 = lane & 31;
    f32x16 st[MT][NT];   // fp32 message scalars of this wave's channels x edges (live in registers for the whole tile)
    f32x16 am[MT][NT], al2[MT][NT];
    f32x16 gm[NT], gl[NT];
    const h8* xh8 = (const h8*)XH;
    const h8* xl8 = (const h8*)XL;
    // vector path: the workgroup's waves form two sets of ET/16 "vector waves" (wave g of a set owns edges 16g .. 16g+15; lane: q = lane >> 4,
    // edge 16g + (lane & 15)).  The role alternates between the sets from GCP2 to GCP2 (the state is handed over in LDS), so that every
    // SIMD carries the same share of the vector work whatever the placement of the co-resident workgroup.
    constexpr int NVW = ET / 16;
    static_assert(NW == 2 * NVW, "two alternating sets of vector waves");
    const int vhalf = wave / NVW;                        // which set this wave belongs to
    const int vq = lane >> 4, ve = 16 * (wave - vhalf * NVW) + (lane & 15);
    v4f* VV4 = (v4f*)(smem + Geo::OFF_VV);               // [3][8][ETP] float4: message vectors, component x, channel group cg, edge
    h8* VHB = (h8*)(smem + Geo::OFF_VHB);                // [3][3][ET]: hidden vectors of the current GCP2 as [hi(3) 0 | lo'(3) 0] images
    float amax = 0.f;                                    // largest |x| that went into an f16 image (range guard)

    // ---- P2: msg0 GEMM ----------------------------------------------------------------------------------------------------
    {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
#ifdef GCDM_ABL_NOPQ
                    for (int t = 0; t < 4; ++t) am[m][n][4 * q + t] = 0.f;
#else
                    for (int t = 0; t < 4; t += 2) {
                        const f32x2 s2 = pk_add((f32x2){in.pqi[m][n][q][t], in.pqi[m][n][q][t + 1]}, (f32x2){in.pqj[m][n][q][t], in.pqj[m][n][q][t + 1]});
                        am[m][n][4 * q + t] = s2[0]; am[m][n][4 * q + t + 1] = s2[1];
                    }
#endif
        STAMP(3);
        constexpr int SILU0_N0 = (NT == 2 && MT == 1) ? 1 : 0;       // N-tile 0's SiLU rides in the GEMM's tail (x3_tail_skew)
        if constexpr (SILU0_N0) {
            tile_gemm_x3z<MT, NT, PD, KB0C, false, true, X3_RO>(am, al2, ring, wp, o0H, o0L, xh8, xl8, ETP, lane, [&]() {
                silu_merge16<X3_PKG>(st[0][0], am[0][0], al2[0][0], X3_INV_SCALE);
            });
        } else {
            tile_gemm_x3z<MT, NT, PD, KB0C, false, true, X3_RO>(am, al2, ring, wp, o0H, o0L, xh8, xl8, ETP, lane);
        }
        x3_prefetch_b<MT, PD>(ring, wp, wp.off(ax.wH[0] + (size_t)mt0 * 18 * 64), wp.off(ax.wL[0] + (size_t)mt0 * 18 * 64), 18);
        GateW<MT> gw0;
        gate_prefetch<MT>(gw0, wp, ax.wg0H, ax.wg0L, mt0);
        STAMP(4);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = SILU0_N0; n < NT; ++n) silu_merge16(st[m][n], am[m][n], al2[m][n], X3_INV_SCALE);
        STAMP(5);
#ifndef GCDM_ABL_NOGATE
        gate_partial_x3p<MT, NT, true>(gm, gl, st, gw0);
        if (NW == 4) {               // four partials = the four slots the vector waves sum: no fold needed
            put_gate_partial<NT, ET>(PG, gm, gl, wave, lane, false);
        } else {
            if (wave < 4) put_gate_partial<NT, ET>(PG, gm, gl, wave, lane, false);
            __syncthreads();         // also: every wave is done reading the msg0 operand images
            if (wave >= 4) put_gate_partial<NT, ET>(PG, gm, gl, wave - 4, lane, true);
        }
#endif
        STAMP(6);
    }
    if (NW == 4) __syncthreads();
    STAMP(7);
    // ---- P3: state images ---------------------------------------------------------------------------------------------------
#ifndef GCDM_ABL_NOSTORE
    store_state_x3<MT, NT>(XH, XL, 0, st, ETP, mt0, lane, amax);
#endif
    STAMP(8);
    __syncthreads();
    STAMP(9);

    // ---- residual message GCP2s k = 1..3; the vector part of the previous GCP2 rides in the shadow of each GEMM ------------------
    constexpr int GPP = GCDM_SG / PARTS;                 // attention: float4 groups of the message scalars per thread
    int nxt_tile = start_ + it_;                         // the tile whose operands are requested from the last GCP2 on (set there)
    static_for<0, 3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const GcpW& w = a.mk[k];
        const uint32_t gwH = wp.off(ax.wH[k] + (size_t)mt0 * 18 * 64), gwL = wp.off(ax.wL[k] + (size_t)mt0 * 18 * 64);
        if (k == GCDM_STAMP_K) STAMP(10);
        [[maybe_unused]] auto silu_n0 = [&]() {             // SiLU of N-tile 0, issued between the last MFMAs of N-tile 1 (tile_gemm_x3s, tail skew)
#pragma unroll
            for (int m = 0; m < MT; ++m) silu_merge16<X3_PKG>(am[m][0], am[m][0], al2[m][0], X3_INV_SCALE);
        };
        if (vhalf == (k & 1)) {
            VecStage<ET, H0, k == 0, VE == 8> vs;
            vs.PG = PG; vs.bg = k == 0 ? a.bg0 : a.mk[k == 0 ? 0 : k - 1].bg; vs.FR = FR; vs.VH = VH; vs.VHB = VHB; vs.VV4 = VV4; vs.XH = XH; vs.XL = XL;
            vs.fA = k == 0 ? ax.vf0H : ax.vf1[k == 0 ? 0 : k - 1]; vs.fB = k == 0 ? ax.vf0L : ax.vf2[k == 0 ? 0 : k - 1];
            vs.pH = ax.vpH[k]; vs.pL = ax.vpL[k];
            vs.ve = ve; vs.vq = vq; vs.lane = lane;
#ifdef GCDM_ABL_VECFMA       // (timing ablation: the hooked stages replaced by the same number of INDEPENDENT fp32 FMAs -- is it the instruction count or the stages' dependency structure?)
            AblFmaFill fx_;
            for (int i = 0; i < 24; ++i) fx_.x[i] = 0.001f * (lane + i);
            tile_gemm_x3s<MT, NT, PD, 18, 16, true, X3_RO>(am, al2, ring, wp, gwH, gwL, xh8, xl8, ETP, lane, [&](auto) { fx_.run(); }, silu_n0);
            amax = fmaxf(amax, fx_.x[0] * 1e-30f + fx_.x[23] * 1e-30f);
#elif defined(GCDM_ABL_VECNONE)
            tile_gemm_x3s<MT, NT, PD, 18, 16, true, X3_RO>(am, al2, ring, wp, gwH, gwL, xh8, xl8, ETP, lane, [&](auto) {}, silu_n0);
#else
            tile_gemm_x3s<MT, NT, PD, 18, 16, true, X3_RO>(am, al2, ring, wp, gwH, gwL, xh8, xl8, ETP, lane, [&](auto rc) { vs.template run<decltype(rc)::value>(); }, silu_n0);
#endif
            amax = fmaxf(amax, vs.amax);
        } else {
            tile_gemm_x3s<MT, NT, PD, 18, 16, false, X3_RO>(am, al2, ring, wp, gwH, gwL, xh8, xl8, ETP, lane, [](auto) {}, silu_n0);
        }
        if (k < 2) x3_prefetch_b<MT, PD>(ring, wp, wp.off(ax.wH[k < 2 ? k + 1 : 2] + (size_t)mt0 * 18 * 64), wp.off(ax.wL[k < 2 ? k + 1 : 2] + (size_t)mt0 * 18 * 64), 18);
        GateW<MT> gwk;
        gate_prefetch<MT>(gwk, wp, ax.wgH[k], ax.wgL[k], mt0);
        if (k == GCDM_STAMP_K) STAMP(12);
        constexpr int SILU_N0 = (NT == 2 && MT == 1) ? 1 : 0;      // N-tile 0 was done inside the GEMM's tail
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = SILU_N0; n < NT; ++n) silu_merge16(am[m][n], am[m][n], al2[m][n], X3_INV_SCALE);
        if (k == GCDM_STAMP_K) STAMP(13);
        [[maybe_unused]] h8 fin_w1[2], fin_w2[2];
#ifndef GCDM_ABL_NOGATE
        gate_partial_x3p<MT, NT, true>(gm, gl, am, gwk);
        // the next tile (clamped to the workgroup's last one: no branch): its index words and per-edge constants are requested behind the last
        // gate contraction (its operands are dead) and arrive under the fold of the gate partials and the state image; the gathers that need the index words are dealt out
        // over the attention phase (load_gather_part) and arrive under the aggregation
        // the vector_up operands of the last GCP2's vector part (finish_only below) are requested ahead of the next tile's index words and constants: loads
        // return in order, so behind those HBM streams they would wait for them (-DGCDM_X3_FIN_LATE: requested at the point of use, rounds 1-4)
#ifndef GCDM_X3_FIN_LATE
        if (k == 2 && vhalf == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { fin_w1[j] = wp.ld(wp.off(ax.vf1[2] + 64 * j)); fin_w2[j] = wp.ld(wp.off(ax.vf2[2] + 64 * j)); }
        }
#endif
        if (k == 2) {
            nxt_tile = start_ + min(it_ + stride_, cnt_ - 1);
            ix = load_idx(a, me, nxt_tile);
            load_const(std::integral_constant<int, 0>{}, a, me, nxt_tile, in);      // (in three bursts with the fold / the residual add between them)
        }
#ifdef GCDM_ABL_GATE_NOPG
        asm volatile("" ::"v"(gm[0]), "v"(gl[0]));
        if (false) {
#else
        if (NW == 4) {
#endif
            put_gate_partial<NT, ET>(PG, gm, gl, wave, lane, false);
        } else {
            if (wave < 4) put_gate_partial<NT, ET>(PG, gm, gl, wave, lane, false);
            __syncthreads();         // also: every wave is done reading the old XH8 / XL8 images
            if (wave >= 4) put_gate_partial<NT, ET>(PG, gm, gl, wave - 4, lane, true);
        }
#endif
        if (k == GCDM_STAMP_K) STAMP(14);
        // 4 waves: every wave is done reading the old images; gate partials complete.  8 waves: the barrier inside the fold said the first, and
        // the partials are complete at the barrier behind the state images (last GCP2: behind the attention partials)
        if (k == 2) load_const(std::integral_constant<int, 1>{}, a, me, nxt_tile, in);
        if (NW == 4) __syncthreads();
        if (k == GCDM_STAMP_K) STAMP(15);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                              // residual add in fp32 (gcpnet.py:701), two channels per instruction
                    const f32x2 s2 = pk_add((f32x2){st[m][n][r], st[m][n][r + 1]}, (f32x2){am[m][n][r], am[m][n][r + 1]});
                    st[m][n][r] = s2[0]; st[m][n][r + 1] = s2[1];
                }
        if (k == 2) load_const(std::integral_constant<int, 2>{}, a, me, nxt_tile, in);
        if (k < 2) {
#ifndef GCDM_ABL_NOSTORE
            store_state_x3<MT, NT>(XH, XL, 0, st, ETP, mt0, lane, amax);
#endif
        } else {
            // last GCP2: scalar message attention (gcpnet.py:703-707) on the registers.  Each wave contracts its 32 channels of the final state
            // with the attention weights (WAX4, staged once per workgroup; the two lane halves hold different channels of the same edge), the
            // 8 partial logits of an edge meet in LDS (the frame buffer is dead by now), and behind ONE barrier -- which also completes the gate
            // partials the vector part reads -- every wave forms the edge's sigmoid itself and scales its channels before they go out as the
            // fp32 image the segment sums read: no pass over the image, no attention table, two barriers instead of four.
            float* ATT_P = FR;                                   // [edge][NW] partial logits
            float pd[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) pd[n] = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                v4f wq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wq[q] = WAX4[8 * (mt0 + m) + 2 * q + half];        // channels 32 (mt0 + m) + 8 q + 4 half + {0..3}
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pd[n] += wq[r >> 2][r & 3] * st[m][n][r];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                pd[n] += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, pd[n])));
                if (half == 0) ATT_P[(32 * n + l31) * NW + wave] = pd[n];
            }
            __syncthreads();
            if (vhalf == 1) {             // the vector part of the last GCP2 (no GEMM left to hide it in)
                VecStage<ET, H0, false, VE == 8> vs;       // (pipelined where it pays: GEOM -0.5 %; the 16-channel instantiation measures +1.1 % with it)
                vs.PG = PG; vs.bg = a.mk[2].bg; vs.FR = FR; vs.VH = VH; vs.VHB = VHB; vs.VV4 = VV4; vs.XH = XH; vs.XL = XL;
                vs.fA = ax.vf1[2]; vs.fB = ax.vf2[2]; vs.pH = nullptr; vs.pL = nullptr;
                vs.ve = ve; vs.vq = vq; vs.lane = lane;
#ifndef GCDM_X3_FIN_LATE
                vs.w1[0] = fin_w1[0]; vs.w1[1] = fin_w1[1]; vs.w2[0] = fin_w2[0]; vs.w2[1] = fin_w2[1]; vs.preloaded = true;
#endif
#ifdef GCDM_ABL_FINW        // (timing ablation, round 5: the last vector part WITHOUT its four operand loads, which are issued behind the next tile's HBM prefetches and return in order)
                { const h8 c_ = {(_Float16)0.01f, (_Float16)0.02f, (_Float16)0.03f, (_Float16)0.01f, (_Float16)0.02f, (_Float16)0.03f, (_Float16)0.01f, (_Float16)0.02f};
                  vs.w1[0] = c_; vs.w1[1] = c_; vs.w2[0] = c_; vs.w2[1] = c_; vs.preloaded = true; }
#endif
                vs.finish_only();
            }
            float att[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float lg = a.ba;
#pragma unroll
                for (int w4 = 0; w4 < NW / 4; ++w4) {
                    const v4f pp = *(const v4f*)(ATT_P + (32 * n + l31) * NW + 4 * w4);
                    lg += pp[0]; lg += pp[1]; lg += pp[2]; lg += pp[3];
                }
                att[n] = fast_sigmoid(lg);
            }
            static_for<0, GCH / 2>([&](auto gc) { load_gather_part(gc, a, me, ix, in); });          // next tile's VDI / VDJ words: first half
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 s2 = pk_mul1((f32x2){st[m][n][r], st[m][n][r + 1]}, att[n]);
                        st[m][n][r] = s2[0]; st[m][n][r + 1] = s2[1];
                    }
            static_for<GCH / 2, GCH>([&](auto gc) { load_gather_part(gc, a, me, ix, in); });
            __builtin_amdgcn_sched_barrier(0);
            store_state<MT, NT, false>(XS4, 0, st, ETP, mt0, lane, 0);
        }
        if (k == GCDM_STAMP_K) STAMP(16);
        __syncthreads();
        if (k == GCDM_STAMP_K) STAMP(17);
    });
    STAMP(18);
    over |= amax > X3_RANGE;
    if (__any(over) && lane == 0) atomicOr(ax.flags_dev, GCDM_FLAG_F16_RANGE_BIT);

    // ---- aggregation: segment sums of the attention-weighted messages (fp32 data) -------------------------------------------------------
#ifndef GCDM_ABL_NOAGG
    STAMP(19);
    {
        const int nseg = m_misc[0];
        // work items: (segment, float4 unit) -- 64 scalar groups + 3 x 8 groups of the message vectors (VV4[x][cg][edge] is already float4 by
        // channel): 88 per segment, so the 4-5 row segments of a QM9 tile are ONE trip of the 512 threads (round 2: 160 units, the vector
        // components one float each -> 1.4 trips, the second one on three waves only)
        constexpr int UNITS = GCDM_SG + 3 * (GCDM_V / 4);
        // work-item order (round 5): all scalar units of all segments first (GCDM_SG = 64 per segment: every wave of that part runs ONE path on one segment),
        // then the vector units -- rounds 2-4 numbered the items segment by segment (88 per segment), so that most waves held both kinds and ran both paths
        // one after the other.  Same sums in the same order per item: same bits.
        const int nscal = nseg * GCDM_SG;
        for (int wk = tid; wk < nseg * UNITS; wk += EK_THREADS) {
#ifdef GCDM_X3_AGG_BY_SEGMENT
            const int sg = wk / UNITS, un = wk - sg * UNITS;
#else
            const int rv = wk - nscal;
            const int sg = wk < nscal ? wk / GCDM_SG : rv / (UNITS - GCDM_SG), un = wk < nscal ? wk - sg * GCDM_SG : GCDM_SG + rv - sg * (UNITS - GCDM_SG);
#endif
            const int2 rec = m_rec[sg];
            const int node = rec.x, sb = rec.y & 255, en = (rec.y >> 8) & 255;
            const bool whole = (rec.y >> 16) != 0;
            float* dst = whole ? a.AGG + (size_t)node * GCDM_AGGW : a.PART + ((size_t)(e0 / ET) * 2 + (sb == 0 ? 0 : 1)) * GCDM_AGGW;
            if (un < GCDM_SG) {
                // (4 edges per trip: the LDS reads of a trip are independent, the additions keep the edge order -- same bits as the plain loop;
                //  8 per trip with a masked tail, i.e. 3 round trips instead of 7 for a 19-edge row, was measured: +-0, more instructions)
                v4f s = {0.f, 0.f, 0.f, 0.f};
                const v4f* xp = XS4 + un * ETP;
                int x = sb;
                for (; x + 4 <= en; x += 4) {
                    const v4f v0 = xp[x], v1 = xp[x + 1], v2 = xp[x + 2], v3 = xp[x + 3];
                    s += v0; s += v1; s += v2; s += v3;
                }
                for (; x < en; ++x) s += xp[x];
                *(v4f*)(dst + 4 * un) = s * (1.0f / X3_C);           // back to true units
            } else {
                const int r = un - GCDM_SG, comp = r / (GCDM_V / 4), cg = r - comp * (GCDM_V / 4);
                const v4f* vp = VV4 + (comp * 8 + cg) * ETP;
                v4f s = {0.f, 0.f, 0.f, 0.f};
                int x = sb;
                for (; x + 4 <= en; x += 4) {
                    const v4f v0 = vp[x], v1 = vp[x + 1], v2 = vp[x + 2], v3 = vp[x + 3];
                    s += v0; s += v1; s += v2; s += v3;
                }
                for (; x < en; ++x) s += vp[x];
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[GCDM_S + 3 * (4 * cg + j) + comp] = s[j];      // AGG column S + 3c + comp (reference flatten layout)
            }
        }
    }
#endif
    STAMP_END(20);
    it_ += stride_;
    if (it_ >= cnt_) break;
    // (no barrier here: the one inside the next tile's pre-phase says that every wave is done with this tile's LDS)
    }
    if constexpr (TR::ON) {
        static_assert(ET == 64, "tail role: 64-edge tiles (one workgroup per CU)");
        x3_wait_block<0>();
        __syncthreads();                           // every wave's stores of the last tile are in the L2; LDS is free
        TR::tail(lx, smem, tail_have_prev, tail_prev_first, tail_prev_last, tail_rel_end);      // (takes the thread index and the workgroup's XCD group afresh: nothing is carried across the loop for it)
    }
}
