// gcdm_embed_x3.hip.h -- split-precision (f16 x3) variant of the edge-embedding kernel (gfx950).
//
// Same outputs as k_edge_embed (gcdm_kernels.hip.h; reference: GCPEmbedding.edge_embedding, a GCP2 (1,1) -> (Se,Ve) with bottleneck 1,
// gcpnet.py:551-603 with 418-491, and localize, components/__init__.py:122-171): e' = SiLU(W_s [e | norms | q] + b), the gated scale of
// the embedded edge vectors, the unit vector and the frames.  k_edge_embed evaluates the two small contractions ([Se x 26] and
// [Ve x Se] per edge) as 3 000 VALU FMAs per edge with scalar-loaded weights; here a wave takes 32 edges and runs them on
// v_mfma_f32_32x32x16_f16 with split operands (gcdm_edge_x3.hip.h): 24 MFMAs per 32 edges, about a third of the VALU work.
//   lanes l and l + 32 share edge (l & 31); both evaluate the geometry (same instructions), each one half of the vector-norm inputs.
//   K slots of the scalar contraction (k = 16 kb + 8 half + j -- the B operand a lane supplies for k-block kb):
//     kb 0:  norm[half * Ve/2 + j]  (j < Ve/2)
//     kb 1:  j = 0: e (half 0) / e_sc (half 1, self-conditioning);  j = 1..5: q[0..4] (half 0) / q[5..8] (half 1);  j = 7: 1 (half 0: bias row)
//   the host packs scalar_out.weight / bias with that column order (pack_embed_x3, gcdm_api.hip).
//   The gate contraction takes its B operand straight from the accumulator registers (same map as gate_partial_x3 / pack_gate_x3).
#pragma once
#include "gcdm_edge_x3.hip.h"

struct EdgeEmbedX3Args {
    X3Const x3c;            // MUST stay the first member (X3_KARG)
    EdgeEmbedArgs base;
    const h8 *wH, *wL;      // scalar_out as A operands [ceil(Se/32)][2][64]
    const h8 *wgH, *wgL;    // vector_out_scale as A operands [Se/16][64]
};

template <int SE, int VE>
__global__ __launch_bounds__(256) void k_edge_embed_x3(EdgeEmbedX3Args ax) {
    const EdgeEmbedArgs& a = ax.base;
    constexpr int MT = (SE + 31) / 32, NH = VE / 2, GB = SE / 16;
    constexpr int RV = SE >= 32 ? 16 : 8;            // accumulator registers that hold real channels (Se = 16: rows 0..15 of the one M-tile)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int E = a.E, N = a.N;
    const int eraw = blockIdx.x * 128 + wave * 32 + l31;
    const bool valid = eraw < E;
    const int eid = valid ? eraw : E - 1;
    const int i = a.EROW[eid], j = a.ECOL[eid];
    // weights of this lane (independent of the geometry: requested first)
    h8 wh[MT][2], wl[MT][2], gh[GB], gl_[GB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { wh[m][kb] = ax.wH[(m * 2 + kb) * 64 + lane]; wl[m][kb] = ax.wL[(m * 2 + kb) * 64 + lane]; }
#pragma unroll
    for (int b = 0; b < GB; ++b) { gh[b] = ax.wgH[b * 64 + lane]; gl_[b] = ax.wgL[b * 64 + lane]; }
    // e = |x_i - x_j|^2, xi = unit vector, both from the UN-centralised positions (gcpnet.py:1102,1109)
    const float d0 = a.X0[i] - a.X0[j], d1 = a.X0[N + i] - a.X0[N + j], d2 = a.X0[2 * N + i] - a.X0[2 * N + j];
    const float es = d0 * d0 + d1 * d1 + d2 * d2;
    const float nr = sqrtf(es);
    float u[3] = {0.f, 0.f, 0.f};
    if (nr > 0.f) { u[0] = d0 / nr; u[1] = d1 / nr; u[2] = d2 / nr; }
    float f[9];
    frame_of(a.XC[i], a.XC[N + i], a.XC[2 * N + i], a.XC[j], a.XC[N + j], a.XC[2 * N + j], f);
    float usc[3] = {0.f, 0.f, 0.f}, es_sc = 0.f;
    if (a.sc) {
        const float s0 = a.X0SC[i] - a.X0SC[j], s1 = a.X0SC[N + i] - a.X0SC[N + j], s2 = a.X0SC[2 * N + i] - a.X0SC[2 * N + j];
        es_sc = s0 * s0 + s1 * s1 + s2 * s2;
        const float ns = sqrtf(es_sc);
        if (ns > 0.f) { usc[0] = s0 / ns; usc[1] = s1 / ns; usc[2] = s2 / ns; }
    }
    if (valid) {                                     // the two lanes of an edge share the stores of its geometry
        if (half == 0) {
#pragma unroll
            for (int x = 0; x < 3; ++x) a.U[(size_t)x * E + eid] = u[x];
#pragma unroll
            for (int r = 0; r < 3; ++r) a.FR[(size_t)r * E + eid] = f[r];
            if (a.sc) {
#pragma unroll
                for (int x = 0; x < 3; ++x) a.USC[(size_t)x * E + eid] = usc[x];
            }
        } else {
#pragma unroll
            for (int r = 3; r < 9; ++r) a.FR[(size_t)r * E + eid] = f[r];
        }
    }
    // inputs of the scalar contraction
    float in0[8], in1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { in0[s] = 0.f; in1[s] = 0.f; }
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        const int hh = half * NH + s;
        const float w = a.wd[hh], w1 = a.sc ? a.wd1[hh] : 0.f;
        const float v0 = u[0] * w + usc[0] * w1, v1 = u[1] * w + usc[1] * w1, v2 = u[2] * w + usc[2] * w1;
        in0[s] = fast_sqrt(v0 * v0 + v1 * v1 + v2 * v2 + 1e-8f) + 1e-8f;
    }
    float q[10];
    q[9] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = a.wdf[k], w1 = a.sc ? a.wdf1[k] : 0.f;
        const float v0 = u[0] * w + usc[0] * w1, v1 = u[1] * w + usc[1] * w1, v2 = u[2] * w + usc[2] * w1;
#pragma unroll
        for (int r = 0; r < 3; ++r) q[3 * k + r] = f[3 * r] * v0 + f[3 * r + 1] * v1 + f[3 * r + 2] * v2;
    }
    in1[0] = half ? es_sc : es;
#pragma unroll
    for (int s = 1; s <= 5; ++s) in1[s] = half ? q[4 + s] : q[s - 1];
    in1[7] = half ? 0.f : 1.f;
    h8 bh[2], bl[2];
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
        h2 hi, lo;
        split16x2(in0[s], in0[s + 1], hi, lo);
        bh[0][s] = hi[0]; bh[0][s + 1] = hi[1];
        bl[0][s] = lo[0]; bl[0][s + 1] = lo[1];
        split16x2(in1[s], in1[s + 1], hi, lo);
        bh[1][s] = hi[0]; bh[1][s + 1] = hi[1];
        bl[1][s] = lo[0]; bl[1][s + 1] = lo[1];
    }
    x3_settle(bh[0], bl[0]);
    x3_settle(bh[1], bl[1]);
    __builtin_amdgcn_sched_barrier(0);          // both k-blocks' splits complete before the first MFMA: none is scheduled into the MFMA burst
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 p[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 am = MFMA16(wh[m][0], bh[0], zero);
        f32x16 al = MFMA16(wh[m][0], bl[0], zero);
        al = MFMA16(wl[m][0], bh[0], al);
        am = MFMA16(wh[m][1], bh[1], am);
        al = MFMA16(wh[m][1], bl[1], al);
        al = MFMA16(wl[m][1], bh[1], al);
#pragma unroll
        for (int r = 0; r < RV; ++r) p[m][r] = fast_silu(am[r] + al[r] * X3_INV_SCALE);
#pragma unroll
        for (int r = RV; r < 16; ++r) p[m][r] = 0.f;
        if (valid) {
#pragma unroll
            for (int t = 0; t < RV / 4; ++t)         // channels 32 m + 8 t + 4 half + {0..3} = float4 group 8 m + 2 t + half
                a.EP4[(size_t)(8 * m + 2 * t + half) * E + eid] = (v4f){p[m][4 * t], p[m][4 * t + 1], p[m][4 * t + 2], p[m][4 * t + 3]};
        }
    }
    // vector gate: Wg . e' (channels of block b = (m, jb): registers 8 jb .. 8 jb + 7 of M-tile m)
    // every split of every block first, fenced off the matrix pipe (sched_barrier): no v_fma_mix is scheduled between two MFMAs of this kernel
    // (tools/isa_census.py --lint rule 2; the stale-B-operand fault of this kernel appeared exactly when the compiler interleaved them)
    f32x16 gm = zero, gl = zero;
    h8 xh[GB], xl[GB];
    __builtin_amdgcn_sched_barrier(0);          // ... nor between the MFMAs of the scalar_out contraction above
#pragma unroll
    for (int b = 0; b < GB; ++b) {
        const int m = b >> 1, jb = b & 1;
#pragma unroll
        for (int s = 0; s < 8; s += 2) {
            h2 hi, lo;
            split16x2(p[m][8 * jb + s], p[m][8 * jb + s + 1], hi, lo);
            xh[b][s] = hi[0]; xh[b][s + 1] = hi[1];
            xl[b][s] = lo[0]; xl[b][s + 1] = lo[1];
        }
        x3_settle(xh[b], xl[b]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < GB; ++b) {
        gm = MFMA16(gh[b], xh[b], gm);
        gl = MFMA16(gh[b], xl[b], gl);
        gl = MFMA16(gl_[b], xh[b], gl);
    }
    if (valid) {
#pragma unroll
        for (int r = 0; r < VE / 2; ++r) {           // gate rows c = (r & 3) + 8 (r >> 2) + 4 half
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            const float sg = fast_sigmoid(gm[r] + gl[r] * X3_INV_SCALE + a.bg[c]);
            a.AL[(size_t)c * E + eid] = a.kappa[c] * sg;
            if (a.sc) a.BL[(size_t)c * E + eid] = a.kappa1[c] * sg;
        }
    }
}
