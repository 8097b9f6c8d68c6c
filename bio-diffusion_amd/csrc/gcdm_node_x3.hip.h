// gcdm_node_x3.hip.h -- split-precision (f16 x3) variant of the node kernels (gfx950).
//
// Same work as k_node (gcdm_kernels.hip.h): node embedding, or feed-forward GCP2 + residual + position update, followed by the
// node-level halves of the next layer's msg0 or the output projection -- with the dense contractions on v_mfma_f32_32x32x16_f16 and
// operands split as x = hi + 2^-11 lo' (see gcdm_edge_x3.hip.h).  One workgroup = 32 nodes, 512 threads: wave w owns M-tile w
// (32 output channels); the fp32 node scalars h of those channels stay in its registers, LDS holds the hi / lo' images.
#pragma once
#include <type_traits>
#include "gcdm_edge_x3.hip.h"

struct GcpX3 {
    const h8 *wH, *wL; int KB;     // scalar_out (first Linear) packed [M'/32][KB][64]
    const h8 *w2H, *w2L;           // feed-forward second Linear packed [8][16][64] (or null)
    const h8 *wgH, *wgL;           // vector gate packed [8][2][64] (or null)
    const h8 *vmH, *vmL;           // [W_down; W_frames] as 16x16x32 A operands, packed [M-tiles][K-blocks][64] (pack_vecmat, gcdm_api.hip)
};

struct NodeX3Args {
    X3Const x3c;                   // MUST stay the first member (X3_KARG)
    NodeArgs base;
    GcpX3 emb, ff, pos, proj;
    const h8 *wpqH, *wpqL;         // next layer's msg0 node halves, packed [16][16][64]
    const float* bpqx;             // c * bias of the next layer's msg0 node halves (scaled units of the edge kernel, X3_C)
    const h8 *vdH, *vdL;           // next layer's msg0 vector halves [wddI; wddJ] (2 x (H0+3) rows x 32) as 16x16x32 A operands, [3][1][64]
    float* prof;                   // optional [tiles][8 waves][24] phase time stamps (gcdm_profile_enable(h, 3))
};

#ifdef GCDM_STAMPS          // (see STAMP in gcdm_kernels.hip.h)
#define NSTAMP(i)                                                                                       \
    do {                                                                                                \
        if (ax.prof && lane == 0) ax.prof[((size_t)tile_index * 8 + wave) * 24 + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_start); \
    } while (0)
#else
#define NSTAMP(i) ((void)0)
#endif

// generic GCP2 pre-phase writing the extended-K rows as hi / lo' images (rows of [W_down; W_frames] split over the PARTS threads of an entity)
template <int T, int H, int V_IN, int NTHR>
__device__ __forceinline__ bool gcp2_pre_x3g(const float* __restrict__ wdd, const float* VV, int vch0, const float* FR, char* XH, char* XL,
                                             int gN8, int gQ8, int gEnd8, float* VH, int e, int part) {
    constexpr int TP = T + 1, PARTS = NTHR / T, ROWS = H + 3, NH = (ROWS + PARTS - 1) / PARTS;
    float ax[NH], ay[NH], az[NH];
    const float* wrow[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        ax[i] = ay[i] = az[i] = 0.f;
        const int hh = part + PARTS * i;
        wrow[i] = wdd + (hh < ROWS ? hh : ROWS - 1) * V_IN;
    }
    const float* vp = VV + (vch0 * 3) * TP + e;
#pragma unroll 8
    for (int c = 0; c < V_IN; ++c) {
        const float vx = vp[0], vy = vp[TP], vz = vp[2 * TP];
        vp += 3 * TP;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const float wc = wrow[i][c];
            ax[i] += wc * vx; ay[i] += wc * vy; az[i] += wc * vz;
        }
    }
    float f[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) f[r] = FR[r * TP + e];
    bool over = false;
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int hh = part + PARTS * i;
        const float vx = ax[i], vy = ay[i], vz = az[i];
        if (hh < H) {
            over |= put16(XH, XL, TP, gN8 + (hh >> 3), hh & 7, e, fast_sqrt(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f);
            VH[(hh * 3 + 0) * TP + e] = vx;
            VH[(hh * 3 + 1) * TP + e] = vy;
            VH[(hh * 3 + 2) * TP + e] = vz;
        } else if (hh < ROWS) {
            const int k = hh - H;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = 3 * k + r;
                over |= put16(XH, XL, TP, gQ8 + (idx >> 3), idx & 7, e, f[3 * r] * vx + f[3 * r + 1] * vy + f[3 * r + 2] * vz);
            }
        }
    }
    if (part == PARTS - 1) {
        for (int hh = H; hh < 8 * (gQ8 - gN8); ++hh) put16(XH, XL, TP, gN8 + (hh >> 3), hh & 7, e, 0.f);
        for (int idx = 9; idx < 16; ++idx) put16(XH, XL, TP, gQ8 + (idx >> 3), idx & 7, e, 0.f);
        for (int g = gQ8 + 2; g < gEnd8; ++g) {
            *(v4f*)(XH + (g * TP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
            *(v4f*)(XL + (g * TP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    return over;
}

// ---- small-M vector contractions of the node kernels on the matrix pipe --------------------------------------------------------
// OUT[r][x][node] = sum_c W[r][c] * V[c][x][node]  (r < 16 MT rows, c < 32 KB channels; V = fp32 rows VV[(vch0 + c) * 3 + x][node]).
// One wave per (component x, group of 16 nodes): waves 0..5 of the 8 (v_mfma_f32_16x16x32_f16, split precision; lane l: q = l >> 4,
// n = l & 15 -- A[row n][k = 8q + j], B[k = 8q + j][col n], D[row 4q + i][col n]).  `store(row, x, node_in_tile, value)` receives
// the results.  As VALU FMAs (16 threads per node, every thread re-reading all vector components) the same contractions cost
// 700 (feed-forward pre-phase) and 670 (next layer's msg0 halves) instructions per thread.
template <int MT, int KB>
struct VecMatW {                 // the A operands of one vecmat_mfma, requested ahead of the phase that uses them
    h8 aH[MT * KB], aL[MT * KB];
    __device__ __forceinline__ void load(const h8* __restrict__ wH, const h8* __restrict__ wL, int lane) {
#pragma unroll
        for (int i = 0; i < MT * KB; ++i) { aH[i] = wH[i * 64 + lane]; aL[i] = wL[i * 64 + lane]; }
    }
};
template <int MT, int KB, int TP, typename StoreFn>
__device__ __forceinline__ void vecmat_mfma(const VecMatW<MT, KB>& w, const float* VV, int vch0, int wave, int lane, float& amax, StoreFn store) {
    if (wave >= 6) return;
    const int x = wave % 3, g = wave / 3, q = lane >> 4, n = lane & 15;
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

// second half of a GCP2 pre-phase: hidden vectors VH[(h*3 + x)][node] (rows 0..H-1) and frame vectors (rows H..H+2) -> norms and frame
// scalars as hi / lo' images of the extended-K rows (gcpnet.py:442-459, scalarize: components/__init__.py:174-219)
template <int T, int H, int NTHR>
__device__ __forceinline__ bool gcp2_pre_tail_x3(const float* VH, const float* FR, char* XH, char* XL, int gN8, int gQ8, int gEnd8, int tid) {
    constexpr int TP = T + 1, ROWS = H + 3;
    bool over = false;
    for (int it = tid; it < ROWS * T; it += NTHR) {
        const int hh = it / T, e = it - hh * T;
        const float vx = VH[(hh * 3 + 0) * TP + e], vy = VH[(hh * 3 + 1) * TP + e], vz = VH[(hh * 3 + 2) * TP + e];
        if (hh < H) {
            over |= put16(XH, XL, TP, gN8 + (hh >> 3), hh & 7, e, fast_sqrt(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f);
        } else {
            const int k = hh - H;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = 3 * k + r;
                over |= put16(XH, XL, TP, gQ8 + (idx >> 3), idx & 7, e, FR[(3 * r) * TP + e] * vx + FR[(3 * r + 1) * TP + e] * vy + FR[(3 * r + 2) * TP + e] * vz);
            }
        }
    }
    for (int it = tid; it < T; it += NTHR) {           // zero the padding slots (weights there are zero, LDS is not)
        const int e = it;
        for (int hh = H; hh < 8 * (gQ8 - gN8); ++hh) put16(XH, XL, TP, gN8 + (hh >> 3), hh & 7, e, 0.f);
        for (int idx = 9; idx < 16; ++idx) put16(XH, XL, TP, gQ8 + (idx >> 3), idx & 7, e, 0.f);
        for (int g = gQ8 + 2; g < gEnd8; ++g) {
            *(v4f*)(XH + (g * TP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
            *(v4f*)(XL + (g * TP + e) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    return over;
}

// vector_up weights and gate bias of the channels a thread finishes (c = part + PARTS * i), requested long before vec_finish needs them:
// loaded at the point of use they cost the phase an exposed L2 round trip (two waves per SIMD, all of them in the same phase)
template <int H, int NC>
struct VecFinW {
    float bg[NC];
    v4f w[NC][H / 4];
    __device__ __forceinline__ void load(const float* __restrict__ bgp, const float* __restrict__ wup, int V_out, int part, int parts) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = min(part + parts * i, V_out - 1);
            bg[i] = bgp[c];
#pragma unroll
            for (int k = 0; k < H / 4; ++k) w[i][k] = *(const v4f*)(wup + c * H + 4 * k);
        }
    }
};
template <int T, int H, int NC, int NTHR, typename StoreFn>
__device__ __forceinline__ void vec_finish_w(const float* PG, const VecFinW<H, NC>& fw, int V_out, const float* VH, int e, int part, StoreFn store) {
    constexpr int TP = T + 1, PARTS = NTHR / T;
    float hx[H], hy[H], hz[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        hx[h] = VH[(h * 3 + 0) * TP + e];
        hy[h] = VH[(h * 3 + 1) * TP + e];
        hz[h] = VH[(h * 3 + 2) * TP + e];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = part + PARTS * i;
        if (c < V_out) {
            float g = fw.bg[i];
#pragma unroll
            for (int w = 0; w < 4; ++w) g += PG[(w * 32 + c) * TP + e];
            const float sg = fast_sigmoid(g);
            float ox = 0.f, oy = 0.f, oz = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const float wv = fw.w[i][h >> 2][h & 3];
                ox += wv * hx[h]; oy += wv * hy[h]; oz += wv * hz[h];
            }
            store(c, ox * sg, oy * sg, oz * sg);
        }
    }
}

// hi / lo' images of a C-layout register block (one M-tile, one N-tile) -> 8-groups gbase8 .. gbase8+3
__device__ __forceinline__ bool store_block_x3(char* XH, char* XL, int gbase8, const f32x16& v, int TP, int lane) {
    f32x16 t[1][1] = {{v}};
    return store_state_x3<1, 1>(XH, XL, gbase8, t, TP, 0, lane);
}

constexpr int NX_GROUPS8 = 70;
constexpr int NX_THREADS = 512;
static_assert(2 * NX_GROUPS8 * NTP * 16 == NK_XS_GROUPS * NTP * 16, "XH8/XL8 must exactly fill the fp32 XS4 region of k_node");

// One 32-node tile of the node kernel.  A device function since round 6: the stand-alone kernel k_node_x3 calls it with tile = blockIdx.x, the fused layer
// kernel (k_edge_msg_x3<..., TAIL = true>, gcdm_edge_x3.hip.h) calls it from the TAIL ROLE of its persistent workgroups with tiles taken from a queue.
template <bool EMBED, int VIN0 = 2>
__device__ __forceinline__ void node_tile_x3(const NodeX3Args& ax, char* smem, const int tile_index, const int tid_in) {
    const NodeArgs& a = ax.base;
    char* XH = smem + NK_OFF_XS;
    char* XL = XH + NX_GROUPS8 * NTP * 16;
    float* VV = (float*)(smem + NK_OFF_VV);
    float* VH = (float*)(smem + NK_OFF_VH);
    float* PG = (float*)(smem + NK_OFF_PG);
    float* FR = (float*)(smem + NK_OFF_FR);
    float* XP = (float*)(smem + NK_OFF_XP);

    const int tid = tid_in, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = tid & 31, part = tid >> 5;       // 16 threads share an entity in the VALU phases
    constexpr int PARTS = NX_THREADS / NT_;
    const int half = lane >> 5, l31 = lane & 31;
    const int N = a.N;
    const int n0 = tile_index * NT_;
    const int nid = min(n0 + e, N - 1);
    const bool valid = (n0 + e) < N;
    const int nidl = min(n0 + l31, N - 1);           // node of this lane in the MFMA (C-layout) phases
    const bool validl = (n0 + l31) < N;
    constexpr int HB8 = 32;   // 8-group base of h inside XH8/XL8 (LAYER: groups 0..31 hold agg.s, then the ff hidden activations)
    constexpr int CB = 32;    // channel base of chi inside VV
    constexpr int PD = 2;     // weight prefetch distance (3 / 4 / 6 / 8: +-0.2 % -- the GEMM phases are bound by the weight stream's bandwidth, not its latency)
    bool over = false;
    float amax = 0.f;
    const uint64_t t_start = ax.prof ? __builtin_amdgcn_s_memtime() : 0;

    for (int r = part; r < 9; r += PARTS) FR[r * NTP + e] = a.FBAR[(size_t)r * N + nid];
    if (part < 3) XP[part * NTP + e] = a.XC[(size_t)part * N + nid];

    f32x16 hst;                  // fp32 master of h: channels 32*wave .. +31 of the 32 nodes (C layout)
    f32x16 am[1][1], al[1][1];
    f32x16 gm[1], gl[1];
    X3Ring<1, PD> ring;
    const h8* xh8 = (const h8*)XH;
    const h8* xl8 = (const h8*)XL;

    // kbc: compile-time k-block count (0 = use KBrt); the production widths fix all of them except the embedding's
    auto gemm = [&](auto kbc, const h8* wH, const h8* wL, int KBrt, int g8base) {
        constexpr int KBC = decltype(kbc)::value;
        const int KB = KBC ? KBC : KBrt;
        const h8* wh = wH + (size_t)wave * KB * 64;
        const h8* wl = wL + (size_t)wave * KB * 64;
        x3_prefetch<1, PD>(ring, wh, wl, KB, lane);
        tile_gemm_x3<1, 1, PD, KBC>(am, al, ring, wh, wl, KB, xh8 + g8base * NTP, xl8 + g8base * NTP, NTP, lane);
    };
    using std::integral_constant;
    auto acc_bias = [&](const float* b) {
        acc_init_bias<1, 1>(am, b, wave, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) al[0][0][r] = 0.f;
    };
    struct GateW { h8 aH[2], aL[2]; };
    auto load_gate = [&](const h8* wgH, const h8* wgL) {      // called before the GEMM whose output the gate contracts
        GateW g;
#pragma unroll
        for (int j = 0; j < 2; ++j) { g.aH[j] = wgH[(wave * 2 + j) * 64 + lane]; g.aL[j] = wgL[(wave * 2 + j) * 64 + lane]; }
        return g;
    };
    auto fold_gate_w = [&](const GateW& g, const f32x16 (&act)[1][1]) {   // two-stage fold of the 8 gate partials
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h8 bh, bl;
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                h2 hi, lo;
                split16x2(act[0][0][8 * j + t], act[0][0][8 * j + t + 1], hi, lo);
                bh[t] = hi[0]; bh[t + 1] = hi[1];
                bl[t] = lo[0]; bl[t + 1] = lo[1];
            }
            x3_settle(bh, bl);
            gm[0] = MFMA16(g.aH[j], bh, j == 0 ? zero : gm[0]);
            gl[0] = MFMA16(g.aH[j], bl, j == 0 ? zero : gl[0]);
            gl[0] = MFMA16(g.aL[j], bh, gl[0]);
        }
        if (wave < 4) put_gate_partial<1>(PG, gm, gl, NTP, wave, lane, false);
        __syncthreads();
        if (wave >= 4) put_gate_partial<1>(PG, gm, gl, NTP, wave - 4, lane, true);
    };
    auto fold_gate = [&](const h8* wgH, const h8* wgL, const f32x16 (&act)[1][1]) {   // (embedding: weights loaded at the point of use)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gm[0][r] = 0.f; gl[0][r] = 0.f; }
        gate_partial_x3<1, 1>(gm, gl, act, wgH, wgL, wave, lane);
        if (wave < 4) put_gate_partial<1>(PG, gm, gl, NTP, wave, lane, false);
        __syncthreads();
        if (wave >= 4) put_gate_partial<1>(PG, gm, gl, NTP, wave - 4, lane, true);
    };

    VecFinW<16, GCDM_V / PARTS> fw_ff;
    VecFinW<8, 1> fw_pos;
    VecMatW<2, 2> vm_ff;
    VecMatW<1, 1> vm_pos;
    if (EMBED) {
        // h_in (fp32 float4 groups in HBM) -> images; zero the unused half of the last 8-group
        const int G8in = (a.FinG + 1) >> 1;
        for (int g = part; g < 2 * G8in; g += PARTS) {
            const v4f v = g < a.FinG ? a.HIN4[(size_t)g * N + nid] : (v4f){0.f, 0.f, 0.f, 0.f};
            h4 vh, vl;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
                h2 hi, lo;
                split16x2(v[t], v[t + 1], hi, lo);
                vh[t] = hi[0]; vh[t + 1] = hi[1];
                vl[t] = lo[0]; vl[t + 1] = lo[1];
                over |= X3_OVER(fmaxf(fabsf(v[t]), fabsf(v[t + 1])) > X3_RANGE);
            }
            const int off = ((g >> 1) * NTP + e) * 16 + 8 * (g & 1);
            *(h4*)(XH + off) = vh;
            *(h4*)(XL + off) = vl;
        }
        for (int r = part; r < 3 * VIN0; r += PARTS) VV[r * NTP + e] = a.CHI0[(size_t)r * N + nid];
        __syncthreads();
        const GcpW& w = a.emb;
        over |= gcp2_pre_x3g<NT_, 32, VIN0, NX_THREADS>(w.wdd, VV, 0, FR, XH, XL, G8in, G8in + 4, 2 * ax.emb.KB, VH, e, part);
        __syncthreads();
        acc_bias(w.b);
        gemm(integral_constant<int, 0>{}, ax.emb.wH, ax.emb.wL, ax.emb.KB, 0);
        merge16(am[0][0], al[0][0], X3_INV_SCALE);     // nonlinearities (None, None): h = p
        hst = am[0][0];
        fold_gate(ax.emb.wgH, ax.emb.wgL, am);
        __syncthreads();
        over |= store_block_x3(XH, XL, HB8 + 4 * wave, hst, NTP, lane);
        vec_finish<NT_, 32, NX_THREADS>(PG, w.bg, w.wup, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
            VV[((CB + c) * 3 + 0) * NTP + e] = ox;
            VV[((CB + c) * 3 + 1) * NTP + e] = oy;
            VV[((CB + c) * 3 + 2) * NTP + e] = oz;
        });
        __syncthreads();
    } else {
        // agg.s -> images (8-groups 0..31), agg.v -> VV channels 0..31, h -> registers + images (8-groups 32..63), chi -> VV channels 32..63.
        // Every global load of the phase is requested before the first one is consumed (row descriptors, then all row pieces, h and chi
        // at once): three dependent round trips instead of one per piece.
        {
            const AggSrc& sg = a.agg;
            const int rs_l = sg.ROWSTART[nidl], n_l = sg.NCNT[nidl];
            int ndx[4], rsx[4], ncx[4];                        // the 4 nodes whose vector rows this wave fetches (lane = channel)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ndx[i] = min(n0 + wave + 8 * i, N - 1);
                rsx[i] = sg.ROWSTART[ndx[i]];
                ncx[i] = sg.NCNT[ndx[i]];
            }
            v4f hv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[q] = a.H4[(size_t)(8 * wave + 2 * q + half) * N + nidl];
            static_assert(96 % PARTS == 0, "chi rows per thread");
            float cv[96 / PARTS];
#pragma unroll
            for (int k = 0; k < 96 / PARTS; ++k) cv[k] = a.CHI[(size_t)(part + PARTS * k) * N + nid];
            vm_ff.load(ax.ff.vmH, ax.ff.vmL, lane);
            fw_ff.load(a.ff.bg, a.ff.wup, GCDM_V, part, PARTS);
            fw_pos.load(a.pos.bg, a.pos.wup, 1, part, PARTS);
            const AggRow2 src = agg_row2(sg, nidl, rs_l, n_l);
            v4f f4[4], g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * wave + 4 * half + 8 * q;
                f4[q] = *(const v4f*)(src.first + col);
                g4[q] = *(const v4f*)(src.next + col);
            }
            AggRow2 sx[4];
            float fa[4], ga[4], fb[4], gb[4];
            const int colb = GCDM_S + 64 + (lane & 31);        // channels 64..95: lanes 32..63 repeat the loads of lanes 0..31, only those store
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sx[i] = agg_row2(sg, ndx[i], rsx[i], ncx[i]);
                fa[i] = sx[i].first[GCDM_S + lane];
                ga[i] = sx[i].next[GCDM_S + lane];
                fb[i] = sx[i].first[colb];
                gb[i] = sx[i].next[colb];
            }
            f32x16 t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = f4[q] + g4[q];
                for (int m = 1; m <= src.more; ++m) v += *(const v4f*)(src.next + (size_t)m * 2 * GCDM_AGGW + 32 * wave + 4 * half + 8 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[4 * q + k] = v[k];
            }
            over |= store_block_x3(XH, XL, 4 * wave, t, NTP, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) hst[4 * q + k] = hv[q][k];
            over |= store_block_x3(XH, XL, HB8 + 4 * wave, hst, NTP, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = wave + 8 * i;
                float va = fa[i] + ga[i], vb = fb[i] + gb[i];
                for (int m = 1; m <= sx[i].more; ++m) {
                    va += sx[i].next[(size_t)m * 2 * GCDM_AGGW + GCDM_S + lane];
                    vb += sx[i].next[(size_t)m * 2 * GCDM_AGGW + colb];
                }
                VV[lane * NTP + x] = va;
                if (lane < 32) VV[(64 + lane) * NTP + x] = vb;
            }
#pragma unroll
            for (int k = 0; k < 96 / PARTS; ++k) VV[(CB * 3 + part + PARTS * k) * NTP + e] = cv[k];
        }
        NSTAMP(1);
        __syncthreads();
        NSTAMP(2);
        // ---- feed-forward GCP2 ------------------------------------------------------------------------------------------------
        {
            const GcpW& w = a.ff;
            // pre-phase: [W_down (16); W_frames (3)] x 64 input vectors on the matrix pipe -> VH, then norms / frame scalars
            vecmat_mfma<2, 2, NTP>(vm_ff, VV, 0, wave, lane, amax, [&](int row, int x, int nd, float v) {
                if (row < 19) VH[(row * 3 + x) * NTP + nd] = v;
            });
            NSTAMP(3);
            __syncthreads();
            over |= gcp2_pre_tail_x3<NT_, 16, NX_THREADS>(VH, FR, XH, XL, 64, 66, 2 * ax.ff.KB, tid);
            NSTAMP(4);
            __syncthreads();
            NSTAMP(5);
            acc_bias(w.b);
            gemm(integral_constant<int, 34>{}, ax.ff.wH, ax.ff.wL, ax.ff.KB, 0);        // K' = 512 + 16 + 16
            NSTAMP(6);
            fast_silu_merge16(am[0][0], am[0][0], al[0][0], X3_INV_SCALE);
            __syncthreads();                                      // every wave is done reading agg.s (8-groups 0..31)
            over |= store_block_x3(XH, XL, 4 * wave, am[0][0], NTP, lane);   // hidden activations of Linear-SiLU-Linear
            __syncthreads();
            NSTAMP(7);
            acc_bias(w.b2);
            const GateW gw = load_gate(ax.ff.wgH, ax.ff.wgL);
            vm_pos.load(ax.pos.vmH, ax.pos.vmL, lane);
            gemm(integral_constant<int, 16>{}, ax.ff.w2H, ax.ff.w2L, 16, 0);
            NSTAMP(8);
            merge16(am[0][0], al[0][0], X3_INV_SCALE);   // nonlinearities (None, None)
            fold_gate_w(gw, am);
            __syncthreads();
            NSTAMP(9);
            const float ml = a.mask ? a.mask[nidl] : 1.f, me = a.mask ? a.mask[nid] : 1.f;     // masked nodes: h, chi, x <- 0 after the layer (gcpnet.py:914-928)
#pragma unroll
            for (int r = 0; r < 16; ++r) hst[r] = (hst[r] + am[0][0][r]) * ml;     // h <- h + ff.s (gcpnet.py:907), fp32
            over |= store_block_x3(XH, XL, HB8 + 4 * wave, hst, NTP, lane);
            vec_finish_w<NT_, 16, GCDM_V / PARTS, NX_THREADS>(PG, fw_ff, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
                VV[((CB + c) * 3 + 0) * NTP + e] = (VV[((CB + c) * 3 + 0) * NTP + e] + ox) * me;
                VV[((CB + c) * 3 + 1) * NTP + e] = (VV[((CB + c) * 3 + 1) * NTP + e] + oy) * me;
                VV[((CB + c) * 3 + 2) * NTP + e] = (VV[((CB + c) * 3 + 2) * NTP + e] + oz) * me;
            });
            NSTAMP(10);
            __syncthreads();
            NSTAMP(11);
        }
        // ---- position update GCP2 -------------------------------------------------------------------------------------------------
        {
            const GcpW& w = a.pos;
            vecmat_mfma<1, 1, NTP>(vm_pos, VV, CB, wave, lane, amax, [&](int row, int x, int nd, float v) {
                if (row < 11) VH[(row * 3 + x) * NTP + nd] = v;
            });
            __syncthreads();
            over |= gcp2_pre_tail_x3<NT_, 8, NX_THREADS>(VH, FR, XH, XL, 64, 65, HB8 + 2 * ax.pos.KB, tid);
            __syncthreads();
            NSTAMP(12);
            acc_bias(w.b);
            const GateW gw = load_gate(ax.pos.wgH, ax.pos.wgL);
            gemm(integral_constant<int, 18>{}, ax.pos.wH, ax.pos.wL, ax.pos.KB, HB8);    // K' = 256 + 8 + 16 -> 288
            NSTAMP(13);
            fast_silu_merge16(am[0][0], am[0][0], al[0][0], X3_INV_SCALE);
            fold_gate_w(gw, am);
            __syncthreads();
            vec_finish_w<NT_, 8, 1, NX_THREADS>(PG, fw_pos, 1, VH, e, part, [&](int c, float ox, float oy, float oz) {
                const float mp = a.mask ? a.mask[nid] : 1.f;
                XP[0 * NTP + e] = (XP[0 * NTP + e] + ox * a.pos_weight) * mp;
                XP[1 * NTP + e] = (XP[1 * NTP + e] + oy * a.pos_weight) * mp;
                XP[2 * NTP + e] = (XP[2 * NTP + e] + oz * a.pos_weight) * mp;
            });
            __syncthreads();
            if (part < 3 && valid) a.XC[(size_t)part * N + nid] = XP[part * NTP + e];
            NSTAMP(14);
        }
    }
    // ---- write the node state back (h from the register master, chi from LDS) ----------------------------------------------------
    if (validl) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            a.H4[(size_t)(8 * wave + 2 * q + half) * N + n0 + l31] = (v4f){hst[4 * q], hst[4 * q + 1], hst[4 * q + 2], hst[4 * q + 3]};
    }
    if (valid) {
        for (int r = part; r < 96; r += PARTS) a.CHI[(size_t)r * N + nid] = VV[(CB * 3 + r) * NTP + e];
    }

    NSTAMP(15);
    if (a.has_next) {
        // ---- node-level halves of the next layer's msg0 ([P | Q], 16 M-tiles): wave w computes M-tiles 2w and 2w + 1 -------------
        // (one GEMM with two M-tiles per wave: the activations are read from LDS once and twice as many weight blocks are in flight)
        VecMatW<3, 1> vm_next;
        vm_next.load(ax.vdH, ax.vdL, lane);
        {
            X3Ring<2, PD> r2;
            f32x16 pm[2][1], pl[2][1];
            const int mt0 = 2 * wave;
            acc_init_bias<2, 1>(pm, ax.bpqx, mt0, lane);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) pl[m][0][r] = 0.f;
            const h8* wh = ax.wpqH + (size_t)mt0 * 16 * 64;
            const h8* wl = ax.wpqL + (size_t)mt0 * 16 * 64;
            x3_prefetch<2, PD>(r2, wh, wl, 16, lane);
            tile_gemm_x3<2, 1, PD, 16>(pm, pl, r2, wh, wl, 16, xh8 + HB8 * NTP, xl8 + HB8 * NTP, NTP, lane);
            if (validl) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int g = 8 * (mt0 + m) + 2 * q + half;
                        v4f o;
#pragma unroll
                        for (int t = 0; t < 4; ++t) o[t] = pm[m][0][4 * q + t] + pl[m][0][4 * q + t] * X3_INV_SCALE;
                        a.PQ4[(size_t)g * N + n0 + l31] = o;
                    }
            }
        }
        NSTAMP(16);
        // vector halves of the next layer's msg0: [W_down; W_frames][:, block] . chi for the row (I) and col (J) block -- 2 x (H0 + 3) rows
        {
            const int rows = a.H0 + 3;
            vecmat_mfma<3, 1, NTP>(vm_next, VV, CB, wave, lane, amax, [&](int row, int x, int nd, float v) {
                if (row < 2 * rows && n0 + nd < N) {
                    const int side = row >= rows, hh = side ? row - rows : row;
                    (side ? a.VDJ : a.VDI)[(size_t)(hh * 3 + x) * N + n0 + nd] = v;
                }
            });
        }
    } else {
        // ---- scalar projection GCP2 (S, V) -> (F+1+C, 0), bottleneck 1, no activation (gcpnet.py:1191-1197) -----------------------
        const GcpW& w = a.proj;
        over |= gcp2_pre_x3g<NT_, 32, GCDM_V, NX_THREADS>(w.wdd, VV, CB, FR, XH, XL, 64, 68, HB8 + 2 * ax.proj.KB, VH, e, part);
        __syncthreads();
        if (wave == 0) {
            acc_bias(w.b);
            gemm(integral_constant<int, 19>{}, ax.proj.wH, ax.proj.wL, ax.proj.KB, HB8);  // K' = 256 + 32 + 16
            if (validl) {
                float* dst = a.OUT + (size_t)(n0 + l31) * a.Dout + 3;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < a.F) {
                        const float hv = am[0][0][r] + al[0][0][r] * X3_INV_SCALE;
                        dst[c] = hv;
                        over |= !(fabsf(hv) <= 3.0e38f);          // range guard: a non-finite network output (see X3_RANGE)
                    }
                }
            }
        }
        if (part < 3 && valid) {
            const float v = XP[part * NTP + e] - a.X0[(size_t)part * N + nid];
            a.VEL[(size_t)part * N + nid] = v;
            if (v != v) atomicOr(a.flags_dev, 1u);
            over |= !(fabsf(v) <= 3.0e38f);
        }
    }
    NSTAMP(17);
    over |= amax > X3_RANGE;
    if (__any(over) && lane == 0) atomicOr(a.flags_dev, GCDM_FLAG_F16_RANGE_BIT);
}

template <bool EMBED, int VIN0 = 2>
__global__ __launch_bounds__(NX_THREADS) void k_node_x3(NodeX3Args ax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    node_tile_x3<EMBED, VIN0>(ax, smem, blockIdx.x, threadIdx.x);
}
