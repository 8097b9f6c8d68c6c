// gcdm_edge_x3v.hip.h -- fused edge-message kernel, split-precision scalars AND vectors on the matrix pipe (gfx950).
//
// k_edge_msg_x3 (gcdm_edge_x3.hip.h) still evaluates the small vector contractions of every GCP2 on the VALU
//     pre   :  vh = W_down v,  u = W_frames v         (11 x 32 per edge and xyz)
//     finish:  v' = (W_up vh) * sigmoid(gate)          (32 x 8  per edge and xyz)
// which, once the big GEMMs run on f16 MFMA, is ~36 % of a tile's time.  Here they are MFMA contractions too: the 3 x 64 (xyz, edge)
// pairs of a tile are 192 columns = six N-tiles, the vector channels are the contraction axis.
//   * message vectors: fp32 master in registers of waves 0..5 (wave t owns N-tile t = (xyz = t >> 1, edge half = t & 1), all 32
//     channels, accumulator layout); LDS holds their hi / lo' images VIH / VIL in 8-channel groups [4][193] x 16 B;
//   * hidden vectors vh (and the 3 frame projections u, parked in the unused k-slots 8..10) as images VHH / VHL [4][193] x 16 B,
//     written straight from the accumulator layout, read back as the B operand of the finish contraction;
//   * two small VALU phases remain: norms / frame scalars from the vh images (needs the 3 xyz tiles of an edge together), and the
//     fold of the 4 gate partials + sigmoid.
#pragma once
#include "gcdm_edge_x3.hip.h"

struct EdgeMsgX3VArgs {
    EdgeMsgX3Args x3;
    const h8 *wup0H, *wup0L;           // msg0 vector_up [32][H0 -> 32] packed [1][2][64]
    const h8 *wddH[3], *wddL[3];       // msg1..3 [W_down; W_frames] (11 -> 32 rows) x 32, packed [1][2][64]
    const h8 *wupH[3], *wupL[3];       // msg1..3 vector_up [32][8 -> 16], packed [1][1][64]
};

constexpr int VTP = 193;               // columns (xyz * 64 + edge) + 1: odd row stride of the vector images

__device__ __forceinline__ bool putv16(char* H, char* L, int g8, int slot, int col, float x) {
    _Float16 hi, lo;
    split16(x, hi, lo);
    const int off = (g8 * VTP + col) * 16 + 2 * slot;
    *(_Float16*)(H + off) = hi;
    *(_Float16*)(L + off) = lo;
    return fabsf(x) > X3_RANGE;
}

__device__ __forceinline__ float getv16(const char* H, const char* L, int g8, int slot, int col) {
    const int off = (g8 * VTP + col) * 16 + 2 * slot;
    return (float)*(const _Float16*)(H + off) + (float)*(const _Float16*)(L + off) * X3_INV_SCALE;
}

// 4 consecutive accumulator registers (rows 8q + 4 half + {0..3}) -> half of an 8-group of the image at column `col`
__device__ __forceinline__ bool putv_quad(char* H, char* L, int g8, int half, int col, float x0, float x1, float x2, float x3) {
    h4 vh, vl;
    _Float16 hi, lo;
    split16(x0, hi, lo); vh[0] = hi; vl[0] = lo;
    split16(x1, hi, lo); vh[1] = hi; vl[1] = lo;
    split16(x2, hi, lo); vh[2] = hi; vl[2] = lo;
    split16(x3, hi, lo); vh[3] = hi; vl[3] = lo;
    const int off = (g8 * VTP + col) * 16 + 8 * half;
    *(h4*)(H + off) = vh;
    *(h4*)(L + off) = vl;
    return fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(x2), fabsf(x3))) > X3_RANGE;
}

template <int ET_>
struct EdgeGeoV {
    static constexpr int TP = ET_ + 1;
    static constexpr int OFF_XS = 0;                                   // XH8 | XL8 : 2 x 36 x 65 x 16
    static constexpr int OFF_VI = OFF_XS + 2 * 36 * TP * 16;           // VIH | VIL : 2 x 4 x 193 x 16   (fp32 VV [96][65] aliases from here)
    static constexpr int OFF_VH = OFF_VI + 2 * 4 * VTP * 16;           // VHH | VHL : 2 x 4 x 193 x 16
    static constexpr int OFF_PG = OFF_VH + 2 * 4 * VTP * 16;           // gate partials [4][32][65] fp32 (slot 0 later holds sigmoid(gate))
    static constexpr int OFF_FR = OFF_PG + 4 * 32 * TP * 4;
    static constexpr int OFF_META = OFF_FR + 9 * TP * 4;
    static constexpr int LDS_BYTES = OFF_META + (ET_ + ET_ + (ET_ + 2) + ET_ + 4) * 4;
};
static_assert(EdgeGeoV<64>::LDS_BYTES <= 163840, "LDS budget");
static_assert(96 * 65 * 4 <= 2 * 4 * VTP * 16 + 2 * 4 * VTP * 16, "fp32 VV alias must fit VI + VH");

template <int SE, int VE>
__global__ __launch_bounds__(512) void k_edge_msg_x3v(EdgeMsgX3VArgs av) {
    constexpr int ET = 64;
    const EdgeMsgX3Args& ax = av.x3;
    const EdgeMsgArgs& a = ax.base;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Geo = EdgeGeoV<ET>;
    constexpr int ETP = Geo::TP, EK_THREADS = 512, PARTS = 8;
    char* XH = smem + Geo::OFF_XS;
    char* XL = XH + 36 * ETP * 16;
    v4f* XS4 = (v4f*)(smem + Geo::OFF_XS);              // fp32 alias after the last GEMM
    char* VIH = smem + Geo::OFF_VI;
    char* VIL = VIH + 4 * VTP * 16;
    float* VV = (float*)(smem + Geo::OFF_VI);           // fp32 alias [96][65] after the last finish
    char* VHH = smem + Geo::OFF_VH;
    char* VHL = VHH + 4 * VTP * 16;
    float* PG = (float*)(smem + Geo::OFF_PG);
    float* FR = (float*)(smem + Geo::OFF_FR);
    int* m_row = (int*)(smem + Geo::OFF_META);
    int* m_col = m_row + ET;
    int* m_seg = m_col + ET;
    float* m_att = (float*)(m_seg + ET + 2);
    int* m_misc = (int*)(m_att + ET);

    constexpr int H0 = (2 * GCDM_V + VE) / 4;
    constexpr int SEG = SE / 4;
    constexpr int N8 = SE / 8;
    constexpr int H0G8 = (H0 + 7) / 8;
    constexpr int Q8 = N8 + H0G8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane, part = wave;
    const int E = a.E, N = a.N;
    const int e0 = blockIdx.x * ET;
    const int nvalid = min(ET, E - e0);
    const int eid = min(e0 + e, E - 1);
    const int ni = a.EROW[eid], nj = a.ECOL[eid];
    const uint64_t t_start = a.prof ? __builtin_amdgcn_s_memtime() : 0;
    bool over = false;
    constexpr int PD = 2;
    X3Ring<1, PD> ring;   // one A-operand ring, reused by every contraction of the tile

    if (wave == 0) {
        m_row[e] = ni;
        m_col[e] = nj;
        const int prev = __shfl_up(ni, 1);
        const bool start = (e < nvalid) && (e == 0 || prev != ni);
        const unsigned long long mask = __ballot(start);
        const int sid = __popcll(mask & ((2ull << lane) - 1ull)) - 1;
        if (start) m_seg[sid] = e;
        if (lane == 0) {
            const int ns = __popcll(mask);
            m_seg[ns] = nvalid;
            m_misc[0] = ns;
        }
    }
    // ---- P1: msg0 pre-phase (node-side gathers; VALU) ------------------------------------------------------------------------
    {
        float fr[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) fr[r] = a.FR[(size_t)r * E + eid];
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < 9; ++r) FR[r * ETP + e] = fr[r];
        }
        for (int g = part; g < SEG; g += PARTS) {
            const v4f v = a.EP4[(size_t)g * E + eid];
            h4 vh, vl;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                _Float16 hi, lo;
                split16(v[t], hi, lo);
                vh[t] = hi; vl[t] = lo;
                over |= fabsf(v[t]) > X3_RANGE;
            }
            const int off = ((g >> 1) * ETP + e) * 16 + 8 * (g & 1);
            *(h4*)(XH + off) = vh;
            *(h4*)(XL + off) = vl;
        }
        float al[VE];
#pragma unroll
        for (int c = 0; c < VE; ++c) al[c] = a.AL[(size_t)c * E + eid];
        const float u0 = a.U[eid], u1 = a.U[(size_t)E + eid], u2 = a.U[2 * (size_t)E + eid];
        constexpr int ROWS0 = H0 + 3, NH0 = (ROWS0 + PARTS - 1) / PARTS;
        float gi[NH0][3], gj[NH0][3], beta[NH0];
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            const int hh = min(part + PARTS * i, ROWS0 - 1);
            const size_t r0 = (size_t)(hh * 3) * N;
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                gi[i][x] = a.VDI[r0 + (size_t)x * N + ni];
                gj[i][x] = a.VDJ[r0 + (size_t)x * N + nj];
            }
            const float* w = a.wddE + hh * VE;
            float bsum = 0.f;
#pragma unroll
            for (int c = 0; c < VE; ++c) bsum += w[c] * al[c];
            beta[i] = bsum;
        }
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            const int hh = part + PARTS * i;
            const float vx = gi[i][0] + beta[i] * u0 + gj[i][0];
            const float vy = gi[i][1] + beta[i] * u1 + gj[i][1];
            const float vz = gi[i][2] + beta[i] * u2 + gj[i][2];
            if (hh < H0) {
                over |= put16(XH, XL, ETP, N8 + (hh >> 3), hh & 7, e, sqrtf(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f);
                over |= putv16(VHH, VHL, hh >> 3, hh & 7, e, vx);            // hidden vectors: images for the finish contraction
                over |= putv16(VHH, VHL, hh >> 3, hh & 7, 64 + e, vy);
                over |= putv16(VHH, VHL, hh >> 3, hh & 7, 128 + e, vz);
            } else if (hh < ROWS0) {
                const int k = hh - H0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int idx = 3 * k + r;
                    over |= put16(XH, XL, ETP, Q8 + (idx >> 3), idx & 7, e, fr[3 * r] * vx + fr[3 * r + 1] * vy + fr[3 * r + 2] * vz);
                }
            }
        }
        if (part == PARTS - 1) {
            for (int hh = H0; hh < 8 * H0G8; ++hh) put16(XH, XL, ETP, N8 + (hh >> 3), hh & 7, e, 0.f);
            for (int idx = 9; idx < 16; ++idx) put16(XH, XL, ETP, Q8 + (idx >> 3), idx & 7, e, 0.f);
            for (int g = Q8 + 2; g < 2 * ax.KB0; ++g) {
                *(v4f*)(XH + (g * ETP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
                *(v4f*)(XL + (g * ETP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (part == PARTS - 2) {   // vh k-slots H0..31 of the finish contraction (its weights are zero there, LDS is not)
            for (int hh = H0; hh < 32; ++hh)
#pragma unroll
                for (int x = 0; x < 3; ++x) putv16(VHH, VHL, hh >> 3, hh & 7, 64 * x + e, 0.f);
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    const int half = lane >> 5, l31 = lane & 31;
    const int mt0 = wave;
    f32x16 st[1][2];
    f32x16 am[1][2], al2[1][2];
    f32x16 gm[2], gl[2];
    f32x16 vmst;                        // fp32 message vectors of N-tile `wave` (waves 0..5): 32 channels x 32 (xyz, edge) columns
    const int vcol = 32 * wave + l31;   // this lane's column in the vector contractions (waves 0..5)
    const int ve = 32 * (wave & 1) + l31;   // ... and the edge it belongs to
    const h8* xh8 = (const h8*)XH;
    const h8* xl8 = (const h8*)XL;

    // gate fold: partials -> sigmoid(gate) in PG slot 0 (one thread per (edge, 4 channels))
    auto gate_sigmoid = [&](const float* bg) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * part + k;
            float g = bg[c];
#pragma unroll
            for (int w = 0; w < 4; ++w) g += PG[(w * 32 + c) * ETP + e];
            PG[c * ETP + e] = fast_sigmoid(g);
        }
    };
    // finish contraction of N-tile `wave`: v' = (W_up vh) * sigmoid(gate); returns the gated update in accumulator layout
    auto finish_mfma = [&](const h8* wupH, const h8* wupL, int KBv, f32x16& outv) {
        f32x16 om[1][1], ol[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) { om[0][0][r] = 0.f; ol[0][0][r] = 0.f; }
        x3_prefetch<1, PD>(ring, wupH, wupL, KBv, lane);
        tile_gemm_x3<1, 1, PD>(om, ol, ring, wupH, wupL, KBv, (const h8*)VHH + 32 * wave, (const h8*)VHL + 32 * wave, VTP, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            outv[r] = (om[0][0][r] + ol[0][0][r] * X3_INV_SCALE) * PG[c * ETP + ve];
        }
    };
    auto store_vimages = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) over |= putv_quad(VIH, VIL, q, half, vcol, vmst[4 * q], vmst[4 * q + 1], vmst[4 * q + 2], vmst[4 * q + 3]);
    };

    // ---- P2: msg0 GEMM -------------------------------------------------------------------------------------------------------------
    {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ri = m_row[32 * n + l31], cj = m_col[32 * n + l31];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = 8 * mt0 + 2 * q + half;
                const v4f p = a.PQ4[(size_t)g * N + ri];
                const v4f qq = a.PQ4[(size_t)(64 + g) * N + cj];
#pragma unroll
                for (int t = 0; t < 4; ++t) { am[0][n][4 * q + t] = p[t] + qq[t]; al2[0][n][4 * q + t] = 0.f; }
            }
        }
        STAMP(3);
        x3_prefetch<1, PD>(ring, ax.w0H + (size_t)mt0 * ax.KB0 * 64, ax.w0L + (size_t)mt0 * ax.KB0 * 64, ax.KB0, lane);
        tile_gemm_x3<1, 2, PD>(am, al2, ring, ax.w0H + (size_t)mt0 * ax.KB0 * 64, ax.w0L + (size_t)mt0 * ax.KB0 * 64, ax.KB0, xh8, xl8, ETP, lane);
        STAMP(4);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[0][n][r] = fast_silu(am[0][n][r] + al2[0][n][r] * X3_INV_SCALE);
        STAMP(5);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gm[n][r] = 0.f; gl[n][r] = 0.f; }
        gate_partial_x3<1, 2>(gm, gl, st, ax.wg0H, ax.wg0L, mt0, lane);
        if (wave < 4) put_gate_partial<2>(PG, gm, gl, ETP, wave, lane, false);
        __syncthreads();
        if (wave >= 4) put_gate_partial<2>(PG, gm, gl, ETP, wave - 4, lane, true);
        STAMP(6);
    }
    __syncthreads();
    STAMP(7);
    // ---- P3: gate fold + state images; then the msg0 vector finish on the matrix pipe -------------------------------------------------
    gate_sigmoid(a.bg0);
    over |= store_state_x3<1, 2>(XH, XL, 0, st, ETP, mt0, lane);
    __syncthreads();
    if (wave < 6) {
        finish_mfma(av.wup0H, av.wup0L, 2, vmst);
        store_vimages();
    }
    STAMP(8);
    __syncthreads();
    STAMP(9);

    // ---- residual message GCP2s k = 1..3 ---------------------------------------------------------------------------------------------
    for (int k = 0; k < 3; ++k) {
        const GcpW& w = a.mk[k];
        // A1: [vh ; u] = [W_down ; W_frames] v  for N-tile `wave`
        if (wave < 6) {
            f32x16 pm[1][1], pl[1][1];
#pragma unroll
            for (int r = 0; r < 16; ++r) { pm[0][0][r] = 0.f; pl[0][0][r] = 0.f; }
            x3_prefetch<1, PD>(ring, av.wddH[k], av.wddL[k], 2, lane);
            tile_gemm_x3<1, 1, PD>(pm, pl, ring, av.wddH[k], av.wddL[k], 2, (const h8*)VIH + 32 * wave, (const h8*)VIL + 32 * wave, VTP, lane);
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = pm[0][0][r] + pl[0][0][r] * X3_INV_SCALE;
            over |= putv_quad(VHH, VHL, 0, half, vcol, v[0], v[1], v[2], v[3]);      // rows 0..7  : hidden vectors
            over |= putv_quad(VHH, VHL, 1, half, vcol, v[4], v[5], v[6], v[7]);      // rows 8..10 : frame projections u (rows 11..15 are 0)
        }
        __syncthreads();
        // A2: norms and frame scalars need the three xyz columns of an edge together (VALU, one output per thread)
        {
            const int h = part;                                            // 8 hidden channels <-> 8 parts
            const float vx = getv16(VHH, VHL, 0, h, e), vy = getv16(VHH, VHL, 0, h, 64 + e), vz = getv16(VHH, VHL, 0, h, 128 + e);
            over |= put16(XH, XL, ETP, 32, h, e, sqrtf(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f);
            for (int idx = part; idx < 9; idx += PARTS) {
                const int kk = idx / 3, r = idx - 3 * kk;
                const float ux = getv16(VHH, VHL, 1, kk, e), uy = getv16(VHH, VHL, 1, kk, 64 + e), uz = getv16(VHH, VHL, 1, kk, 128 + e);
                over |= put16(XH, XL, ETP, 33 + (idx >> 3), idx & 7, e, FR[(3 * r) * ETP + e] * ux + FR[(3 * r + 1) * ETP + e] * uy + FR[(3 * r + 2) * ETP + e] * uz);
            }
            if (part == PARTS - 1) {
                for (int idx = 9; idx < 16; ++idx) put16(XH, XL, ETP, 33 + (idx >> 3), idx & 7, e, 0.f);
                for (int g = 35; g < 2 * ax.KB; ++g) {
                    *(v4f*)(XH + (g * ETP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
                    *(v4f*)(XL + (g * ETP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (k == 0) STAMP(10);
        __syncthreads();
        if (k == 0) STAMP(11);
        acc_init_bias<1, 2>(am, w.b, mt0, lane);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) al2[0][n][r] = 0.f;
        x3_prefetch<1, PD>(ring, ax.wH[k] + (size_t)mt0 * ax.KB * 64, ax.wL[k] + (size_t)mt0 * ax.KB * 64, ax.KB, lane);
        tile_gemm_x3<1, 2, PD>(am, al2, ring, ax.wH[k] + (size_t)mt0 * ax.KB * 64, ax.wL[k] + (size_t)mt0 * ax.KB * 64, ax.KB, xh8, xl8, ETP, lane);
        if (k == 0) STAMP(12);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) am[0][n][r] = fast_silu(am[0][n][r] + al2[0][n][r] * X3_INV_SCALE);
        if (k == 0) STAMP(13);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { gm[n][r] = 0.f; gl[n][r] = 0.f; }
        gate_partial_x3<1, 2>(gm, gl, am, ax.wgH[k], ax.wgL[k], mt0, lane);
        if (wave < 4) put_gate_partial<2>(PG, gm, gl, ETP, wave, lane, false);
        __syncthreads();
        if (wave >= 4) put_gate_partial<2>(PG, gm, gl, ETP, wave - 4, lane, true);
        if (k == 0) STAMP(14);
        __syncthreads();
        if (k == 0) STAMP(15);
        // C: gate fold + residual state (registers) + state images
        gate_sigmoid(w.bg);
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[0][n][r] += am[0][n][r];
        if (k < 2) over |= store_state_x3<1, 2>(XH, XL, 0, st, ETP, mt0, lane);
        else store_state<1, 2, false>(XS4, 0, st, ETP, mt0, lane, 0);
        __syncthreads();
        // D: vector finish on the matrix pipe, residual add in the register master
        if (wave < 6) {
            f32x16 upd;
            finish_mfma(av.wupH[k], av.wupL[k], 1, upd);
#pragma unroll
            for (int r = 0; r < 16; ++r) vmst[r] += upd[r];
            if (k < 2) store_vimages();
        }
        if (k == 0) STAMP(16);
        __syncthreads();
        if (k == 0) STAMP(17);
    }
    STAMP(18);
    if (__any(over) && lane == 0) atomicOr(ax.flags_dev, GCDM_FLAG_F16_RANGE_BIT);

    // fp32 message vectors for the segment sums: VV[(c*3 + xyz)][edge]  (aliases the vector images, which are dead now)
    if (wave < 6) {
        const int xyz = wave >> 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            VV[(c * 3 + xyz) * ETP + ve] = vmst[r];
        }
    }
    // ---- scalar message attention + aggregation (fp32) ------------------------------------------------------------------------------
    {
        float s = 0.f;
        constexpr int GPP = GCDM_SG / PARTS;
        for (int g = part * GPP; g < part * GPP + GPP; ++g) {
            const v4f wv = *(const v4f*)(a.wa + 4 * g);
            const v4f x = XS4[g * ETP + e];
            s += wv[0] * x[0] + wv[1] * x[1] + wv[2] * x[2] + wv[3] * x[3];
        }
        PG[part * ETP + e] = s;
        __syncthreads();
        if (part == 0) {
            float s2 = a.ba;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) s2 += PG[q * ETP + e];
            m_att[e] = fast_sigmoid(s2);
        }
        __syncthreads();
    }
    STAMP(19);
    {
        const int nseg = m_misc[0];
        constexpr int UNITS = GCDM_SG + 3 * GCDM_V;
        for (int wk = tid; wk < nseg * UNITS; wk += EK_THREADS) {
            const int sg = wk / UNITS, un = wk - sg * UNITS;
            const int sb = m_seg[sg], en = m_seg[sg + 1];
            const int node = m_row[sb];
            const bool whole = (en - sb) == a.NCNT[node];
            float* dst = a.AGG + (size_t)node * GCDM_AGGW;
            if (un < GCDM_SG) {
                v4f s = {0.f, 0.f, 0.f, 0.f};
                for (int x = sb; x < en; ++x) s += XS4[un * ETP + x] * m_att[x];
                if (whole) {
                    *(v4f*)(dst + 4 * un) = s;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) atomicAdd(dst + 4 * un + t, s[t]);
                }
            } else {
                const int r = un - GCDM_SG;
                float s = 0.f;
                for (int x = sb; x < en; ++x) s += VV[r * ETP + x];
                if (whole) dst[GCDM_S + r] = s;
                else atomicAdd(dst + GCDM_S + r, s);
            }
        }
    }
    STAMP(20);
}
