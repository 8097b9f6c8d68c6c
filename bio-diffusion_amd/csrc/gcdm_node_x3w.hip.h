// gcdm_node_x3w.hip.h -- the per-layer node kernel of the split-precision mode on 64-node tiles (round 4).
//
// Same work and the same arithmetic as k_node_x3<false> (gcdm_node_x3.hip.h: feed-forward GCP2 + residual, position-update GCP2, then the node-level
// halves of the next layer's msg0 or the output projection; gcpnet.py:834-930, 1191-1216), but one workgroup takes 64 nodes: every wave still owns
// one M-tile (32 output channels) and now TWO N-tiles, so each 2 KB of streamed weights feeds 6 MFMAs instead of 3.  The 32-node kernel's GEMM
// phases ran at half the matrix rate because of exactly that stream (1.6 MB of split weights per 32 nodes, 42-46 B/clk/CU against an L2 -> CU
// ceiling of ~50-56: profiles/r03_phase_stamps_qm9.txt), and its 608 tiles were 3 rounds for 2.4 rounds of work on 256 CUs.
//
// LDS (160 KB) does not hold the 32-node layout twice, so the operand images are re-used over the layer:
//   IMG  [38 groups][65] x 16 B hi + the same lo'   79.0 KB   agg.s (groups 0-31) for the first 16 k-blocks of the feed-forward GEMM, then h (0-31) +
//                                                              the norm / frame-scalar rows (32-35) for its other 18; the hidden activations; the
//                                                              new h (+ the position GCP2's rows 32-35, or the projection's 32-37)
//   VV   [192 rows][65] fp32                          49.9 KB   rows 0-95 agg.v, rows 96-191 chi (row = channel * 3 + x)
//   VH   [96 rows][65] fp32                           25.0 KB   hidden vectors of the running GCP2 (57 / 33 / 96 rows)
//   PG   gate partials, 4 slots x [32][65] fp32: slots 0-2 ALIAS VV rows 0-95 (agg.v is dead once the feed-forward pre-phase has contracted it),
//        slot 3 the tail of VH (rows 57-88: the feed-forward and position GCP2s use 57 / 33 rows)
//   FR, XP, the feed-forward vector_up table + gate bias (2.2 KB: 68 registers per thread in the 32-node kernel)   5.3 KB
// The k-blocks of every contraction run in the order of the 32-node kernel ([agg.s | h | extended rows] etc.), every accumulator sees the same
// MFMAs in the same order: the results are bit-identical to k_node_x3<false> (tests: test_node_tile_sizes_agree_bitwise).
#pragma once
#include "gcdm_node_x3.hip.h"

constexpr int NW_T = 64, NW_TP = 65;                 // nodes per tile, LDS row pitch
constexpr int NW_IMG_GROUPS = 38;
constexpr int NW_OFF_XH = 0;
constexpr int NW_OFF_XL = NW_OFF_XH + NW_IMG_GROUPS * NW_TP * 16;
constexpr int NW_OFF_VV = NW_OFF_XL + NW_IMG_GROUPS * NW_TP * 16;
constexpr int NW_OFF_VH = NW_OFF_VV + 192 * NW_TP * 4;
constexpr int NW_OFF_FR = NW_OFF_VH + 96 * NW_TP * 4;
constexpr int NW_OFF_XP = NW_OFF_FR + 9 * NW_TP * 4;
constexpr int NW_OFF_WUP = NW_OFF_XP + 3 * NW_TP * 4;             // feed-forward vector_up [32][16] + gate bias [32], staged once per workgroup
constexpr int NW_LDS_BYTES = NW_OFF_WUP + (GCDM_V * 16 + GCDM_V) * 4;
static_assert(NW_LDS_BYTES <= 160 * 1024, "k_node_x3w: LDS budget");
constexpr int NW_PG3_ROW = 57;                        // first VH row of gate-partial slot 3

// gate-partial slot base (element (c, e) at base[c * NW_TP + e]): slots 0-2 in VV rows 0-95, slot 3 in VH rows 57-88
__device__ __forceinline__ float* nw_pg_slot(float* VV, float* VH, int slot) {
    return slot < 3 ? VV + slot * 32 * NW_TP : VH + NW_PG3_ROW * NW_TP;
}

template <int NT>
__device__ __forceinline__ void nw_put_gate_partial(float* VV, float* VH, const f32x16 (&gm)[NT], const f32x16 (&gl)[NT], int slot, int lane, bool add) {
    const int half = lane >> 5, l31 = lane & 31;
    float* base = nw_pg_slot(VV, VH, slot);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* p = base + c * NW_TP + 32 * n + l31;
            const float v = gm[n][r] + gl[n][r] * X3_INV_SCALE;
            *p = add ? *p + v : v;
        }
}

// vec_finish_w of gcdm_node_x3.hip.h with the gate partials at their aliased places (same sums in the same order)
template <int H, int NC, typename WFn, typename BFn, typename StoreFn>
__device__ __forceinline__ void nw_vec_finish(float* VV, float* VHm, WFn wup, BFn bgf, int V_out, int e, int part, StoreFn store) {
    constexpr int TP = NW_TP, PARTS = 8;
    float hx[H], hy[H], hz[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        hx[h] = VHm[(h * 3 + 0) * TP + e];
        hy[h] = VHm[(h * 3 + 1) * TP + e];
        hz[h] = VHm[(h * 3 + 2) * TP + e];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = part + PARTS * i;
        if (c < V_out) {
            float g = bgf(i, c);
#pragma unroll
            for (int w = 0; w < 4; ++w) g += nw_pg_slot(VV, VHm, w)[c * TP + e];
            const float sg = fast_sigmoid(g);
            float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float wv = wup(i, c, h);
                ox += wv * hx[h]; oy += wv * hy[h]; oz += wv * hz[h];
            }
            store(c, ox * sg, oy * sg, oz * sg);
        }
    }
}

// vecmat_mfma of gcdm_node_x3.hip.h for one (component x, group of 16 nodes) unit: 64 nodes are 12 units, dealt out over the 8 waves
template <int MT, int KB, typename StoreFn>
__device__ __forceinline__ void nw_vecmat_unit(const VecMatW<MT, KB>& w, const float* VV, int vch0, int unit, int lane, float& amax, StoreFn store) {
    constexpr int TP = NW_TP;
    const int x = unit % 3, g = unit / 3, q = lane >> 4, n = lane & 15;
    f32x4 am[MT], al[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) { am[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; al[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float v0[4], v1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = VV[((vch0 + 32 * kb + 8 * q + j) * 3 + x) * TP + 16 * g + n];
            v1[j] = VV[((vch0 + 32 * kb + 8 * q + 4 + j) * 3 + x) * TP + 16 * g + n];
        }
        h4 h0, l0, h1, l1;
        split4(v0, h0, l0, amax);
        split4(v1, h1, l1, amax);
        h8 bh = cat44(h0, h1), bl = cat44(l0, l1);
        x3_settle(bh, bl);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            am[m] = MFMA1632(w.aH[m * KB + kb], bh, am[m]);
            al[m] = MFMA1632(w.aH[m * KB + kb], bl, al[m]);
            al[m] = MFMA1632(w.aL[m * KB + kb], bh, al[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) store(16 * m + 4 * q + i, x, 16 * g + n, am[m][i] + al[m][i] * X3_INV_SCALE);
}

// (a device function since round 6, like node_tile_x3: the fused layer kernel's tail role runs 64-node tiles through it)
__device__ __forceinline__ void node_tile_x3w(const NodeX3Args& ax, char* smem, const int tile_index, const int tid_in) {
    const NodeArgs& a = ax.base;
    char* XH = smem + NW_OFF_XH;
    char* XL = smem + NW_OFF_XL;
    float* VV = (float*)(smem + NW_OFF_VV);
    float* VH = (float*)(smem + NW_OFF_VH);
    float* FR = (float*)(smem + NW_OFF_FR);
    float* XP = (float*)(smem + NW_OFF_XP);
    float* WUP = (float*)(smem + NW_OFF_WUP);              // [32][16] vector_up of the feed-forward GCP2, then its 32 gate biases
    constexpr int TP = NW_TP, PARTS = 8, CB = 32;      // CB: channel base of chi inside VV
    constexpr int PD = 2;

    const int tid = tid_in, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane, part = wave;                   // VALU phases: lane = node of the tile, 8 threads (one per wave) share a node
    const int half = lane >> 5, l31 = lane & 31;
    const int N = a.N;
    const int n0 = tile_index * NW_T;
    const int nid = min(n0 + e, N - 1);
    const bool valid = (n0 + e) < N;
    bool over = false;
    float amax = 0.f;
    const uint64_t t_start = ax.prof ? __builtin_amdgcn_s_memtime() : 0;
    (void)t_start;

    for (int i = tid; i < GCDM_V * 16 + GCDM_V; i += 512) WUP[i] = i < GCDM_V * 16 ? a.ff.wup[i] : a.ff.bg[i - GCDM_V * 16];
    for (int r = part; r < 9; r += PARTS) FR[r * TP + e] = a.FBAR[(size_t)r * N + nid];
    if (part < 3) XP[part * TP + e] = a.XC[(size_t)part * N + nid];

    f32x16 hst[1][2];            // fp32 master of h: channels 32 * wave .. + 31 of the two N-tiles (C layout)
    f32x16 am[1][2], al[1][2];
    f32x16 gm[2], gl[2];
    X3Ring<1, PD> ring;
    const h8* xh8 = (const h8*)XH;
    const h8* xl8 = (const h8*)XL;
    using std::integral_constant;

    // k-blocks [kb0, kb0 + KBC) of a packed matrix with KBT blocks per M-tile against image groups g8base ..
    auto gemm = [&](auto kbc, const h8* wH, const h8* wL, int KBT, int kb0, int g8base) {
        constexpr int KBC = decltype(kbc)::value;
        const h8* wh = wH + ((size_t)wave * KBT + kb0) * 64;
        const h8* wl = wL + ((size_t)wave * KBT + kb0) * 64;
        x3_prefetch<1, PD>(ring, wh, wl, KBC, lane);
        tile_gemm_x3<1, 2, PD, KBC>(am, al, ring, wh, wl, KBC, xh8 + g8base * TP, xl8 + g8base * TP, TP, lane);
    };
    auto acc_bias = [&](const float* b) {
        acc_init_bias<1, 2>(am, b, wave, lane);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) al[0][n][r] = 0.f;
    };
    struct GateWn { h8 aH[2], aL[2]; };
    auto load_gate = [&](const h8* wgH, const h8* wgL) {
        GateWn g;
#pragma unroll
        for (int j = 0; j < 2; ++j) { g.aH[j] = wgH[(wave * 2 + j) * 64 + lane]; g.aL[j] = wgL[(wave * 2 + j) * 64 + lane]; }
        return g;
    };
    auto fold_gate_w = [&](const GateWn& g, const f32x16 (&act)[1][2]) {   // two-stage fold of the 8 gate partials
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                h8 bh, bl;
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                    h2 hi, lo;
                    split16x2(act[0][n][8 * j + t], act[0][n][8 * j + t + 1], hi, lo);
                    bh[t] = hi[0]; bh[t + 1] = hi[1];
                    bl[t] = lo[0]; bl[t + 1] = lo[1];
                }
                x3_settle(bh, bl);
                gm[n] = MFMA16(g.aH[j], bh, j == 0 ? zero : gm[n]);
                gl[n] = MFMA16(g.aH[j], bl, j == 0 ? zero : gl[n]);
                gl[n] = MFMA16(g.aL[j], bh, gl[n]);
            }
        if (wave < 4) nw_put_gate_partial<2>(VV, VH, gm, gl, wave, lane, false);
        __syncthreads();
        if (wave >= 4) nw_put_gate_partial<2>(VV, VH, gm, gl, wave - 4, lane, true);
    };

    VecFinW<8, 1> fw_pos;
    VecMatW<2, 2> vm_ff;
    VecMatW<1, 1> vm_pos;

    // ---- load phase: agg.s -> images (groups 0..31), agg.v -> VV channels 0..31, h -> registers, chi -> VV channels 32..63 ---------------------
    {
        const AggSrc& sg = a.agg;
        int nidl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) nidl[n] = min(n0 + 32 * n + l31, N - 1);          // node of this lane in the MFMA (C-layout) phases
        int rs_l[2], n_l[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { rs_l[n] = sg.ROWSTART[nidl[n]]; n_l[n] = sg.NCNT[nidl[n]]; }
        v4f hv[2][4];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[n][q] = a.H4[(size_t)(8 * wave + 2 * q + half) * N + nidl[n]];
        static_assert(96 % PARTS == 0, "chi rows per thread");
        float cv[96 / PARTS];
#pragma unroll
        for (int k = 0; k < 96 / PARTS; ++k) cv[k] = a.CHI[(size_t)(part + PARTS * k) * N + nid];
        vm_ff.load(ax.ff.vmH, ax.ff.vmL, lane);
        f32x16 t[1][2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const AggRow2 src = agg_row2(sg, nidl[n], rs_l[n], n_l[n]);
            v4f f4[4], g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * wave + 4 * half + 8 * q;
                f4[q] = *(const v4f*)(src.first + col);
                g4[q] = *(const v4f*)(src.next + col);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = f4[q] + g4[q];
                for (int m = 1; m <= src.more; ++m) v += *(const v4f*)(src.next + (size_t)m * 2 * GCDM_AGGW + 32 * wave + 4 * half + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[0][n][4 * q + k] = v[k];
            }
        }
        over |= store_state_x3<1, 2>(XH, XL, 0, t, TP, wave, lane);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) hst[0][n][4 * q + k] = hv[n][q][k];
        // vector rows of agg: node x = wave + 8 i of the tile, lane = channel (two batches of four nodes: bounded registers)
        const int colb = GCDM_S + 64 + (lane & 31);        // channels 64..95: lanes 32..63 repeat the loads of lanes 0..31, only those store
#pragma unroll
        for (int b4 = 0; b4 < 2; ++b4) {
            AggRow2 sx[4];
            float fa[4], ga[4], fb[4], gb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nd = min(n0 + wave + 8 * (4 * b4 + i), N - 1);
                sx[i] = agg_row2(sg, nd, sg.ROWSTART[nd], sg.NCNT[nd]);
                fa[i] = sx[i].first[GCDM_S + lane];
                ga[i] = sx[i].next[GCDM_S + lane];
                fb[i] = sx[i].first[colb];
                gb[i] = sx[i].next[colb];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = wave + 8 * (4 * b4 + i);
                float va = fa[i] + ga[i], vb = fb[i] + gb[i];
                for (int m = 1; m <= sx[i].more; ++m) {
                    va += sx[i].next[(size_t)m * 2 * GCDM_AGGW + GCDM_S + lane];
                    vb += sx[i].next[(size_t)m * 2 * GCDM_AGGW + colb];
                }
                VV[lane * TP + x] = va;
                if (lane < 32) VV[(64 + lane) * TP + x] = vb;
            }
        }
#pragma unroll
        for (int k = 0; k < 96 / PARTS; ++k) VV[(CB * 3 + part + PARTS * k) * TP + e] = cv[k];
    }
    NSTAMP(1);
    __syncthreads();
    NSTAMP(2);
    // ---- feed-forward GCP2 ----------------------------------------------------------------------------------------------------------------------
    {
        const GcpW& w = a.ff;
        // pre-phase: [W_down (16); W_frames (3)] x 64 input vectors on the matrix pipe -> VH, then norms / frame scalars -> image groups 32..35
        for (int unit = wave; unit < 12; unit += 8)
            nw_vecmat_unit<2, 2>(vm_ff, VV, 0, unit, lane, amax, [&](int row, int x, int nd, float v) {
                if (row < 19) VH[(row * 3 + x) * TP + nd] = v;
            });
        NSTAMP(3);
        __syncthreads();
        over |= gcp2_pre_tail_x3<NW_T, 16, 512>(VH, FR, XH, XL, 32, 34, 36, tid);
        NSTAMP(4);
        __syncthreads();
        NSTAMP(5);
        acc_bias(w.b);
        gemm(integral_constant<int, 16>{}, ax.ff.wH, ax.ff.wL, 34, 0, 0);                 // K' blocks 0..15: agg.s
        __syncthreads();                                                                 // every wave is done reading the agg.s images
        over |= store_state_x3<1, 2>(XH, XL, 0, hst, TP, wave, lane);                    // h images over them
        __syncthreads();
        gemm(integral_constant<int, 18>{}, ax.ff.wH, ax.ff.wL, 34, 16, 0);                // K' blocks 16..33: h, norms, frame scalars
        NSTAMP(6);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            fast_silu_merge16(am[0][n], am[0][n], al[0][n], X3_INV_SCALE);
        __syncthreads();                                                                 // every wave is done reading the h images
        over |= store_state_x3<1, 2>(XH, XL, 0, am, TP, wave, lane);                     // hidden activations of Linear-SiLU-Linear
        __syncthreads();
        NSTAMP(7);
        acc_bias(w.b2);
        const GateWn gw = load_gate(ax.ff.wgH, ax.ff.wgL);
        vm_pos.load(ax.pos.vmH, ax.pos.vmL, lane);
        gemm(integral_constant<int, 16>{}, ax.ff.w2H, ax.ff.w2L, 16, 0, 0);
        NSTAMP(8);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            merge16(am[0][n], al[0][n], X3_INV_SCALE);       // nonlinearities (None, None)
        fold_gate_w(gw, am);
        __syncthreads();
        NSTAMP(9);
        float ml[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) ml[n] = a.mask ? a.mask[min(n0 + 32 * n + l31, N - 1)] : 1.f;   // masked nodes: h, chi, x <- 0 after the layer (gcpnet.py:914-928)
        const float me = a.mask ? a.mask[nid] : 1.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) hst[0][n][r] = (hst[0][n][r] + am[0][n][r]) * ml[n];     // h <- h + ff.s (gcpnet.py:907), fp32
        over |= store_state_x3<1, 2>(XH, XL, 0, hst, TP, wave, lane);                    // (the hidden images are dead: GEMM 2 is behind the fold's barrier)
        nw_vec_finish<16, GCDM_V / PARTS>(VV, VH, [&](int, int c, int h) { return WUP[c * 16 + h]; }, [&](int, int c) { return WUP[GCDM_V * 16 + c]; }, GCDM_V, e, part,
                                          [&](int c, float ox, float oy, float oz) {
            VV[((CB + c) * 3 + 0) * TP + e] = (VV[((CB + c) * 3 + 0) * TP + e] + ox) * me;
            VV[((CB + c) * 3 + 1) * TP + e] = (VV[((CB + c) * 3 + 1) * TP + e] + oy) * me;
            VV[((CB + c) * 3 + 2) * TP + e] = (VV[((CB + c) * 3 + 2) * TP + e] + oz) * me;
        });
        NSTAMP(10);
        __syncthreads();
        NSTAMP(11);
    }
    // ---- position update GCP2 -------------------------------------------------------------------------------------------------------------------
    {
        const GcpW& w = a.pos;
        fw_pos.load(a.pos.bg, a.pos.wup, 1, part, PARTS);
        for (int unit = wave; unit < 12; unit += 8)
            nw_vecmat_unit<1, 1>(vm_pos, VV, CB, unit, lane, amax, [&](int row, int x, int nd, float v) {
                if (row < 11) VH[(row * 3 + x) * TP + nd] = v;
            });
        __syncthreads();
        over |= gcp2_pre_tail_x3<NW_T, 8, 512>(VH, FR, XH, XL, 32, 33, 36, tid);
        __syncthreads();
        NSTAMP(12);
        acc_bias(w.b);
        const GateWn gw = load_gate(ax.pos.wgH, ax.pos.wgL);
        gemm(integral_constant<int, 18>{}, ax.pos.wH, ax.pos.wL, 18, 0, 0);              // K' = 256 + 8 + 16 -> 288
        NSTAMP(13);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            fast_silu_merge16(am[0][n], am[0][n], al[0][n], X3_INV_SCALE);
        fold_gate_w(gw, am);
        __syncthreads();
        nw_vec_finish<8, 1>(VV, VH, [&](int i, int, int h) { return fw_pos.w[i][h >> 2][h & 3]; }, [&](int i, int) { return fw_pos.bg[i]; }, 1, e, part,
                            [&](int c, float ox, float oy, float oz) {
            const float mp = a.mask ? a.mask[nid] : 1.f;
            XP[0 * TP + e] = (XP[0 * TP + e] + ox * a.pos_weight) * mp;
            XP[1 * TP + e] = (XP[1 * TP + e] + oy * a.pos_weight) * mp;
            XP[2 * TP + e] = (XP[2 * TP + e] + oz * a.pos_weight) * mp;
        });
        __syncthreads();
        if (part < 3 && valid) a.XC[(size_t)part * N + nid] = XP[part * TP + e];
        NSTAMP(14);
    }
    // ---- write the node state back (h from the register master, chi from LDS) ---------------------------------------------------------------------
#pragma unroll
    for (int n = 0; n < 2; ++n)
        if (n0 + 32 * n + l31 < N) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a.H4[(size_t)(8 * wave + 2 * q + half) * N + n0 + 32 * n + l31] =
                    (v4f){hst[0][n][4 * q], hst[0][n][4 * q + 1], hst[0][n][4 * q + 2], hst[0][n][4 * q + 3]};
        }
    if (valid) {
        for (int r = part; r < 96; r += PARTS) a.CHI[(size_t)r * N + nid] = VV[(CB * 3 + r) * TP + e];
    }

    NSTAMP(15);
    if (a.has_next) {
        // ---- node-level halves of the next layer's msg0 ([P | Q], 16 M-tiles): wave w computes M-tiles 2w and 2w + 1, one after the other ----------
        VecMatW<3, 1> vm_next;
        vm_next.load(ax.vdH, ax.vdL, lane);
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int mt = 2 * wave + mm;
            acc_init_bias<1, 2>(am, ax.bpqx, mt, lane);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) al[0][n][r] = 0.f;
            const h8* wh = ax.wpqH + (size_t)mt * 16 * 64;
            const h8* wl = ax.wpqL + (size_t)mt * 16 * 64;
            x3_prefetch<1, PD>(ring, wh, wl, 16, lane);
            tile_gemm_x3<1, 2, PD, 16>(am, al, ring, wh, wl, 16, xh8, xl8, TP, lane);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                if (n0 + 32 * n + l31 < N) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = 8 * mt + 2 * q + half;
                        v4f o;
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = am[0][n][4 * q + t] + al[0][n][4 * q + t] * X3_INV_SCALE;
                        a.PQ4[(size_t)g * N + n0 + 32 * n + l31] = o;
                    }
                }
        }
        NSTAMP(16);
        // vector halves of the next layer's msg0: [W_down; W_frames][:, block] . chi for the row (I) and col (J) block -- 2 x (H0 + 3) rows
        {
            const int rows = a.H0 + 3;
            for (int unit = wave; unit < 12; unit += 8)
                nw_vecmat_unit<3, 1>(vm_next, VV, CB, unit, lane, amax, [&](int row, int x, int nd, float v) {
                    if (row < 2 * rows && n0 + nd < N) {
                        const int side = row >= rows, hh = side ? row - rows : row;
                        (side ? a.VDJ : a.VDI)[(size_t)(hh * 3 + x) * N + n0 + nd] = v;
                    }
                });
        }
    } else {
        // ---- scalar projection GCP2 (S, V) -> (F+1+C, 0), bottleneck 1, no activation (gcpnet.py:1191-1197) ---------------------------------------
        const GcpW& w = a.proj;
        over |= gcp2_pre_x3g<NW_T, 32, GCDM_V, 512>(w.wdd, VV, CB, FR, XH, XL, 32, 36, 38, VH, e, part);
        __syncthreads();
        if (wave == 0) {
            acc_bias(w.b);
            gemm(integral_constant<int, 19>{}, ax.proj.wH, ax.proj.wL, 19, 0, 0);         // K' = 256 + 32 + 16
#pragma unroll
            for (int n = 0; n < 2; ++n)
                if (n0 + 32 * n + l31 < N) {
                    float* dst = a.OUT + (size_t)(n0 + 32 * n + l31) * a.Dout + 3;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c < a.F) {
                            const float hv = am[0][n][r] + al[0][n][r] * X3_INV_SCALE;
                            dst[c] = hv;
                            over |= !(fabsf(hv) <= 3.0e38f);          // range guard: a non-finite network output (see X3_RANGE)
                        }
                    }
                }
        }
        if (part < 3 && valid) {
            const float v = XP[part * TP + e] - a.X0[(size_t)part * N + nid];
            a.VEL[(size_t)part * N + nid] = v;
            if (v != v) atomicOr(a.flags_dev, 1u);
            over |= !(fabsf(v) <= 3.0e38f);
        }
    }
    NSTAMP(17);
    over |= amax > X3_RANGE;
    if (__any(over) && lane == 0) atomicOr(a.flags_dev, GCDM_FLAG_F16_RANGE_BIT);
}

__global__ __launch_bounds__(512) void k_node_x3w(NodeX3Args ax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    node_tile_x3w(ax, smem, blockIdx.x, threadIdx.x);
}
