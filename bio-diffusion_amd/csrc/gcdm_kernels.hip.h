// gcdm_kernels.hip.h -- device code of libgcdm_hip.so (gfx950 / CDNA4 only).
//
// Formulation.  Every dense map of the network is evaluated TRANSPOSED,
//     Out^T[M, T] = W[M, K] . X[K, T]
// with the T "entities" of a tile (64 flat edges, or 32 nodes) on the MFMA column / lane axis and the
// feature channels on the row axis, using v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD):
//   * X lives in LDS as float4 "channel groups":  XS4[g][e] = channels 4g..4g+3 of entity e;
//   * a wave's B operand for 8 consecutive channels is one ds_read_b128 per N-tile
//     (lanes 0-31 take group 2g, lanes 32-63 group 2g+1 -- the contraction order is free);
//   * W is pre-packed on the host so that a lane's A operands for the same 4 MFMAs are one
//     global_load_dwordx4 (each wave owns its own M-tiles: weights go L2 -> VGPR, no LDS staging);
//   * the 32x32 accumulator layout (lane = entity, regs = channels 8q+4*half+{0..3}) is exactly a
//     float4 channel group, so results go back to LDS with ds_write_b128 and can also be fed straight
//     back as a B operand (the vector-gate GEMM contracts over the channels a wave already holds).
//
// Reference semantics follow oracle/gcdm_oracle.py <-> src/models/components/gcpnet.py (cited per kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GCDM_S 256      // h_hidden_dim
#define GCDM_V 32       // chi_hidden_dim
#define GCDM_SG 64      // scalar channel groups (S/4)
#define GCDM_AGGW 352   // S + 3V : width of one aggregated message row

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_silu(float x) { return x * fast_sigmoid(x); }

// ------------------------------------------------------------------------------------------------
// packed weights of one GCP2 (gcpnet.py:265-491, production config) as the kernels consume them
// ------------------------------------------------------------------------------------------------
struct GcpW {
    const v4f* w;      // scalar_out packed [M'/32][G][64] float4   (K' = 8G)
    const float* b;    // scalar_out bias [M'] (zero padded)
    const v4f* w2;     // feed-forward second Linear packed [8][32][64] (or null)
    const float* b2;   // [256]
    const float* wdd;  // [(H+3)][V_in]: vector_down rows, then the 3 vector_down_frames rows
    const v4f* wg;     // vector_out_scale packed [1][32][64] (K = 256), rows >= V_out are zero
    const float* bg;   // [32]
    const float* wup;  // vector_up [V_out][H]
    int G;             // k-groups (of 8 channels) of the scalar_out GEMM
    int H;             // hidden vector channels
    int V_in;
    int V_out;
};

// ------------------------------------------------------------------------------------------------
// tile GEMM:  acc[m][n] += Wpacked(M-tiles mt0..mt0+MT-1) . XS4(groups gbase .. gbase+2G-1)
// ------------------------------------------------------------------------------------------------
template <int MT, int NT>
__device__ __forceinline__ void gemm_group(f32x16 (&acc)[MT][NT], const v4f (&a)[MT], const v4f (&b)[NT]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = MFMA32(a[m][t], b[n][t], acc[m][n]);
}

template <int MT, int NT, int PD = 4>
__device__ __forceinline__ void tile_gemm(f32x16 (&acc)[MT][NT], const v4f* __restrict__ wp, int G,
                                          const v4f* xs4, int TP, int lane) {
    // A (weights) streams L2 -> VGPR through a ring of R = PD+1 statically indexed register sets: the load of k-group
    // g+PD is issued before the MFMAs of group g, so the compiler's counted s_waitcnt leaves PD groups in flight.
    // B (activations) is read from LDS one group ahead (two alternating register sets).  The main loop is branch-free
    // (whole rings only); the remainder runs as a straight-line tail.
    constexpr int R = PD + 1;
    static_assert(R % 2 == 1 || true, "");
    const v4f* wl = wp + lane;
    const v4f* sl = xs4 + (lane >> 5) * TP + (lane & 31);
    const int wstride = G * 64;
    const int last = G - 1;
    v4f a[R][MT];
    v4f b[2][NT];
#pragma unroll
    for (int r = 0; r < PD; ++r) {
        const int gl = min(r, last);
#pragma unroll
        for (int m = 0; m < MT; ++m) a[r][m] = wl[m * wstride + gl * 64];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) b[0][n] = sl[n * 32];
    int g0 = 0;
    for (; g0 + 2 * R <= G; g0 += 2 * R) {   // two rings per trip so that the B double buffer index is static too
#pragma unroll
        for (int r = 0; r < 2 * R; ++r) {
            const int g = g0 + r;
            const int gl = min(g + PD, last);
#pragma unroll
            for (int m = 0; m < MT; ++m) a[(r + PD) % R][m] = wl[m * wstride + gl * 64];
            const int gb = min(g + 1, last);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[(r + 1) & 1][n] = sl[(2 * gb) * TP + n * 32];
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetches ABOVE this group's MFMAs (the scheduler sinks them otherwise)
            gemm_group<MT, NT>(acc, a[r % R], b[r & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 2 * R - 1; ++r) {
        const int g = g0 + r;
        if (g < G) {
            if (r + PD < 2 * R - 1) {
                const int gl = min(g + PD, last);
#pragma unroll
                for (int m = 0; m < MT; ++m) a[(r + PD) % R][m] = wl[m * wstride + gl * 64];
            }
            const int gb = min(g + 1, last);
#pragma unroll
            for (int n = 0; n < NT; ++n) b[(r + 1) & 1][n] = sl[(2 * gb) * TP + n * 32];
            __builtin_amdgcn_sched_barrier(0);
            gemm_group<MT, NT>(acc, a[r % R], b[r & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc[m][n][r] <- bias[channel(r)]   (channel = 32*(mt0+m) + 8*(r>>2) + 4*half + (r&3))
template <int MT, int NT>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MT][NT], const float* __restrict__ bias, int mt0, int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f bv = *(const v4f*)(bias + 32 * (mt0 + m) + 8 * q + 4 * half);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[m][n][4 * q + t] = bv[t];
        }
}

// partial vector-gate GEMM over the channels this wave holds: gacc[n] += Wg[:, own channels] . act[own channels, :]
template <int MT, int NT>
__device__ __forceinline__ void gate_partial(f32x16 (&gacc)[NT], const f32x16 (&act)[MT][NT],
                                             const v4f* __restrict__ wg, int mt0, int lane) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v4f a = wg[(4 * (mt0 + m) + q) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int n = 0; n < NT; ++n) gacc[n] = MFMA32(a[t], act[m][n][4 * q + t], gacc[n]);
        }
}

// PG[wave][c][e] <- gacc  (c = 8*(r>>2) + 4*half + (r&3))
template <int NT>
__device__ __forceinline__ void store_gate_partial(float* PG, const f32x16 (&gacc)[NT], int TP, int slot, int lane, int col0 = 0) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
            PG[(slot * 32 + c) * TP + col0 + 32 * n + l31] = gacc[n][r];
        }
}

// XS4[gbase + own groups][e] (+)= acc
template <int MT, int NT, bool ADD>
__device__ __forceinline__ void store_state(v4f* xs4, int gbase, const f32x16 (&acc)[MT][NT], int TP, int mt0, int lane, int col0 = 0) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = (gbase + 8 * (mt0 + m) + 2 * q + half) * TP + col0 + 32 * n + l31;
                v4f v = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
                if (ADD) v += xs4[idx];
                xs4[idx] = v;
            }
}

template <int MT, int NT>
__device__ __forceinline__ void apply_silu(f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = fast_silu(acc[m][n][r]);
}

// ------------------------------------------------------------------------------------------------
// GCP2 pre-phase for entity e (gcpnet.py:442-459 + scalarize components/__init__.py:174-219):
//   vh = W_down v ; n = sqrt(sum_xyz vh^2 + 1e-8) + 1e-8 ; u = W_frames v ; q[3k+r] = F[r,:].u[:,k]
// rows hh of [W_down; W_frames] are split over the PARTS threads that share an entity.
// ------------------------------------------------------------------------------------------------
template <int T, int H, int V_IN, int NTHR = 256>
__device__ __forceinline__ void gcp2_pre(const float* __restrict__ wdd, const float* VV, int vch0, const float* FR, float* XSf,
                                         int gN, int gQ, int gEnd, float* VH, int e, int part) {
    constexpr int TP = T + 1, PARTS = NTHR / T, ROWS = H + 3, NH = (ROWS + PARTS - 1) / PARTS;
    // each thread owns rows hh = part + PARTS*i of [W_down; W_frames]; the V_IN input vectors are read from LDS once
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
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int hh = part + PARTS * i;
        const float vx = ax[i], vy = ay[i], vz = az[i];
        if (hh < H) {
            XSf[((gN + (hh >> 2)) * TP + e) * 4 + (hh & 3)] = sqrtf(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f;
            VH[(hh * 3 + 0) * TP + e] = vx;
            VH[(hh * 3 + 1) * TP + e] = vy;
            VH[(hh * 3 + 2) * TP + e] = vz;
        } else if (hh < ROWS) {
            const int k = hh - H;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = 3 * k + r;
                XSf[((gQ + (idx >> 2)) * TP + e) * 4 + (idx & 3)] = f[3 * r] * vx + f[3 * r + 1] * vy + f[3 * r + 2] * vz;
            }
        }
    }
    if (part == 0) {  // zero the padding slots of the extended-K rows (weights there are zero, LDS is not)
        for (int hh = H; hh < 4 * (gQ - gN); ++hh) XSf[((gN + (hh >> 2)) * TP + e) * 4 + (hh & 3)] = 0.f;
        for (int idx = 9; idx < 12; ++idx) XSf[((gQ + (idx >> 2)) * TP + e) * 4 + (idx & 3)] = 0.f;
        for (int g = gQ + 3; g < gEnd; ++g) ((v4f*)XSf)[g * TP + e] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
}

// vector output: v'[c] = (W_up vh)[c] * sigmoid(gate[c])  (gcpnet.py:388-411); channels split over PARTS threads
template <int T, int H, int NTHR = 256, typename StoreFn>
__device__ __forceinline__ void vec_finish(const float* PG, const float* __restrict__ bg, const float* __restrict__ wup,
                                           int V_out, const float* VH, int e, int part, StoreFn store) {
    constexpr int TP = T + 1, PARTS = NTHR / T;
    float hx[H], hy[H], hz[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        hx[h] = VH[(h * 3 + 0) * TP + e];
        hy[h] = VH[(h * 3 + 1) * TP + e];
        hz[h] = VH[(h * 3 + 2) * TP + e];
    }
    for (int c = part; c < V_out; c += PARTS) {
        float g = bg[c];
#pragma unroll
        for (int w = 0; w < 4; ++w) g += PG[(w * 32 + c) * TP + e];
        const float sg = fast_sigmoid(g);
        float ox = 0.f, oy = 0.f, oz = 0.f;
        const float* wu = wup + c * H;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const float wv = wu[h];
            ox += wv * hx[h]; oy += wv * hy[h]; oz += wv * hz[h];
        }
        store(c, ox * sg, oy * sg, oz * sg);
    }
}

// ------------------------------------------------------------------------------------------------
// Aggregated messages, deterministic for any molecule size.  A node's row of n flat edges that lies inside ONE edge tile is summed
// there and stored to AGG[node]; a row cut by tile boundaries leaves one partial sum per tile it touches in PART[tile][slot]
// (slot 0: the piece touching the tile's first edge, slot 1: a piece that starts inside the tile and runs to its end), and the node
// kernel adds the pieces in tile order.  No atomics, no zero-fill of AGG, bit-reproducible for rows spanning any number of tiles.
// ------------------------------------------------------------------------------------------------
struct AggSrc {
    const float* AGG;      // [N][352] rows of nodes whose edge row lies inside one tile
    const float* PART;     // [tiles][2][352] partial sums of cut rows
    const int* ROWSTART;   // [N] flat index of the node's first edge
    const int* NCNT;       // [N] row length (atoms of the node's molecule)
    int tile_shift;        // log2(edges per tile) of the edge kernel that produced PART
    const float* ZROW;     // 352 zeros (the branch-free form below adds them where a row has no second piece)
};

struct AggRow {            // where to find node's aggregated row
    const float* first;    // AGG row, or the first partial
    const float* next;     // partial (slot 0) of tile t0 + 1
    int extra;             // number of further tiles; -1: the node has no edges (a masked node): the row is zero
};

__device__ __forceinline__ AggRow agg_row(const AggSrc& s, int node) {
    const int rs = s.ROWSTART[node], n = s.NCNT[node];
    const int t0 = rs >> s.tile_shift, t1 = (rs + n - 1) >> s.tile_shift;
    AggRow r;
    r.extra = t1 - t0;
    if (n == 0) {
        r.extra = -1;
        r.first = r.next = s.AGG;
    } else if (r.extra == 0) {
        r.first = s.AGG + (size_t)node * GCDM_AGGW;
        r.next = r.first;
    } else {
        const int slot = (rs & ((1 << s.tile_shift) - 1)) ? 1 : 0;
        r.first = s.PART + ((size_t)t0 * 2 + slot) * GCDM_AGGW;
        r.next = s.PART + ((size_t)(t0 + 1) * 2) * GCDM_AGGW;
    }
    return r;
}
__device__ __forceinline__ v4f agg_load4(const AggRow& r, int col) {
    if (r.extra < 0) return (v4f){0.f, 0.f, 0.f, 0.f};
    v4f v = *(const v4f*)(r.first + col);
    for (int t = 0; t < r.extra; ++t) v += *(const v4f*)(r.next + (size_t)t * 2 * GCDM_AGGW + col);
    return v;
}
// Branch-free form for kernels that want all their loads in flight at once: row = first + next (+ `more` further partials, rare: only rows
// longer than one tile).  Same summation order as agg_load4 / agg_load1 (adding the zero row changes no bits but the sign of a zero).
struct AggRow2 {
    const float* first;
    const float* next;
    int more;
};
__device__ __forceinline__ AggRow2 agg_row2(const AggSrc& s, int node, int rs, int n) {
    const int t0 = rs >> s.tile_shift, t1 = (rs + n - 1) >> s.tile_shift;
    const bool none = n == 0, whole = t1 == t0;
    const int slot = (rs & ((1 << s.tile_shift) - 1)) ? 1 : 0;
    AggRow2 r;
    r.first = none ? s.ZROW : whole ? s.AGG + (size_t)node * GCDM_AGGW : s.PART + ((size_t)t0 * 2 + slot) * GCDM_AGGW;
    r.next = (none || whole) ? s.ZROW : s.PART + ((size_t)(t0 + 1) * 2) * GCDM_AGGW;
    r.more = (none || whole) ? 0 : t1 - t0 - 1;
    return r;
}

__device__ __forceinline__ float agg_load1(const AggRow& r, int col) {
    if (r.extra < 0) return 0.f;
    float v = r.first[col];
    for (int t = 0; t < r.extra; ++t) v += r.next[(size_t)t * 2 * GCDM_AGGW + col];
    return v;
}

__device__ __forceinline__ void frame_of(float xi0, float xi1, float xi2, float xj0, float xj1, float xj2, float (&f)[9]) {
    // localize, components/__init__.py:122-171 (norm_x_diff=True)
    float d0 = xi0 - xj0, d1 = xi1 - xj1, d2 = xi2 - xj2;
    float c0 = xi1 * xj2 - xi2 * xj1, c1 = xi2 * xj0 - xi0 * xj2, c2 = xi0 * xj1 - xi1 * xj0;
    const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2) + 1.0f;
    const float cn = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + 1.0f;
    d0 /= dn; d1 /= dn; d2 /= dn;
    c0 /= cn; c1 /= cn; c2 /= cn;
    f[0] = d0; f[1] = d1; f[2] = d2;
    f[3] = c0; f[4] = c1; f[5] = c2;
    f[6] = d1 * c2 - d2 * c1; f[7] = d2 * c0 - d0 * c2; f[8] = d0 * c1 - d1 * c0;
}

// ================================================================================================
// K1  prep: one workgroup per molecule (gcpnet.py:1081-1109, 1142-1166; scalarize node-mode mean)
// ================================================================================================
// Step-dependent scalars of one ancestral transition, one row per step index, resident on the device: the kernels of a CAPTURED step (hipGraph,
// gcdm_api.hip: StepGraph) read row [*cursor] instead of taking the values as launch arguments, so one instantiated graph serves every step.
struct StepRow {
    float t;                            // network time of the step (k_prep puts it into the time column of h_in)
    float alpha_coef, c_eps, sigma;     // as in StepArgs
    uint32_t draw;                      // Philox draw index of the step's noise
    uint32_t pad[3];
};

struct PrepArgs {
    const float* xh;   // [N][3+F]
    const float* t;    // [N]
    const float* ctx;  // [N][C] or null
    const int* noff;   // [B+1]
    int N, F, C, FinG; // FinG = ceil((F+1+C)/4)
    float* X0;         // [3][N] un-centralised positions
    float* XC;         // [3][N] centralised positions (updated in place by the layers)
    float* FBAR;       // [9][N] mean_j f_ij
    float* CHI0;       // [6][N] orientations (flat-batch adjacency)
    v4f* HIN4;         // [FinG][N]
    int has_prev, has_next;   // this plan is a slice of a larger flat batch: rows -1 / N of xh exist and are its flat neighbours
    // self-conditioning (gcpnet.py:1112-1139): h_in = [h0 | h_sc | t | context], CHI0 gets the orientations of x_sc as vectors 2, 3,
    // X0SC the un-centralised x_sc for the second edge scalar / vector.  xh_sc may be null (= zeros)
    int sc; const float* xh_sc; float* X0SC;
    // masked nodes (batch.mask with False entries; gcpnet.py:1081, 1094-1099, components/__init__.py:53-92): 1 / 0 per node, or null (all True).
    // Masked nodes enter with zero positions and features, have no edges, are left out of the centroid and keep time / context inputs.
    const float* mask;
    // the sampler's steps (gcdm_api.hip: transition): t is one value for the whole batch -- rows[*cursor].t of a captured step, else t_value -- when t is null;
    // flags_dev: the handle's flag word, cleared here for the kernels behind (what a memset node did before round 6)
    const StepRow* t_rows; int* t_cursor; float t_value;       // t_cursor: the two slots of the step cursor (k_cursor_set)
    uint32_t* flags_dev;
};

__global__ __launch_bounds__(64) void k_prep(PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [3][n] centralised
    const int b = blockIdx.x, o = a.noff[b], n = a.noff[b + 1] - o, D = 3 + a.F;
    if (a.flags_dev && b == 0 && threadIdx.x == 0) *a.flags_dev = 0u;
    float t_all = a.t_value;
    if (a.t_rows) {
        const int cur = a.t_cursor[0];
        t_all = a.t_rows[cur].t;
        if (b == 0 && threadIdx.x == 0) a.t_cursor[1] = cur;
    }
    // the molecule's (masked) positions through LDS: the loads in parallel, the sum below in the serial order every thread repeats
    for (int i = threadIdx.x; i < n; i += 64) {
        const float* p = a.xh + (size_t)(o + i) * D;
        const float mk = a.mask ? a.mask[o + i] : 1.f;
        xs[i] = p[0] * mk; xs[n + i] = p[1] * mk; xs[2 * n + i] = p[2] * mk;
    }
    __syncthreads();
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, cnt = 0.f;
    for (int i = 0; i < n; ++i) {  // same summation order as scatter(sum) over a sorted index
        m0 += xs[i]; m1 += xs[n + i]; m2 += xs[2 * n + i];
        cnt += a.mask ? a.mask[o + i] : 1.f;
    }
    m0 /= cnt; m1 /= cnt; m2 /= cnt;
    __syncthreads();                 // (xs is overwritten with the centralised positions below)
    for (int i = threadIdx.x; i < n; i += 64) {
        const int g = o + i;
        const float* p = a.xh + (size_t)g * D;
        const float mk = a.mask ? a.mask[g] : 1.f;
        const float x0 = p[0] * mk, x1 = p[1] * mk, x2 = p[2] * mk;
        a.X0[g] = x0; a.X0[a.N + g] = x1; a.X0[2 * a.N + g] = x2;
        const float c0 = x0 - m0 * mk, c1 = x1 - m1 * mk, c2 = x2 - m2 * mk;
        a.XC[g] = c0; a.XC[a.N + g] = c1; a.XC[2 * a.N + g] = c2;
        xs[i] = c0; xs[n + i] = c1; xs[2 * n + i] = c2;
        // orientations (protein_graph_dataset.py:217-225): flat neighbours, zero padded at the global ends
        float fw[3] = {0.f, 0.f, 0.f}, bw[3] = {0.f, 0.f, 0.f};
        if (g + 1 < a.N || a.has_next) {
            const float* q = p + D;
            const float mq = (a.mask && g + 1 < a.N) ? a.mask[g + 1] : 1.f;
            const float e0 = q[0] * mq - x0, e1 = q[1] * mq - x1, e2 = q[2] * mq - x2, nr = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
            if (nr > 0.f) { fw[0] = e0 / nr; fw[1] = e1 / nr; fw[2] = e2 / nr; }
        }
        if (g > 0 || a.has_prev) {
            const float* q = p - D;
            const float mq = (a.mask && g > 0) ? a.mask[g - 1] : 1.f;
            const float e0 = q[0] * mq - x0, e1 = q[1] * mq - x1, e2 = q[2] * mq - x2, nr = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
            if (nr > 0.f) { bw[0] = e0 / nr; bw[1] = e1 / nr; bw[2] = e2 / nr; }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) { a.CHI0[x * a.N + g] = fw[x]; a.CHI0[(3 + x) * a.N + g] = bw[x]; }
        const int Fsc = a.sc ? a.F : 0;
        if (a.sc) {
            const float* ps = a.xh_sc ? a.xh_sc + (size_t)g * D : nullptr;
            const float s0 = ps ? ps[0] : 0.f, s1 = ps ? ps[1] : 0.f, s2 = ps ? ps[2] : 0.f;
            a.X0SC[g] = s0; a.X0SC[a.N + g] = s1; a.X0SC[2 * a.N + g] = s2;
            float fs[3] = {0.f, 0.f, 0.f}, bs_[3] = {0.f, 0.f, 0.f};
            if (ps && (g + 1 < a.N || a.has_next)) {
                const float* q = ps + D;
                const float e0 = q[0] - s0, e1 = q[1] - s1, e2 = q[2] - s2, nr = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
                if (nr > 0.f) { fs[0] = e0 / nr; fs[1] = e1 / nr; fs[2] = e2 / nr; }
            }
            if (ps && (g > 0 || a.has_prev)) {
                const float* q = ps - D;
                const float e0 = q[0] - s0, e1 = q[1] - s1, e2 = q[2] - s2, nr = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
                if (nr > 0.f) { bs_[0] = e0 / nr; bs_[1] = e1 / nr; bs_[2] = e2 / nr; }
            }
#pragma unroll
            for (int x = 0; x < 3; ++x) { a.CHI0[(6 + x) * a.N + g] = fs[x]; a.CHI0[(9 + x) * a.N + g] = bs_[x]; }
        }
        // h_in = [h0 | h_sc | t | context], zero padded to a whole number of float4 groups
        const int Fin = a.F + Fsc + 1 + a.C;
        for (int gg = 0; gg < a.FinG; ++gg) {
            v4f v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * gg + k;
                float val = 0.f;
                if (c < a.F) val = p[3 + c] * mk;
                else if (c < a.F + Fsc) val = a.xh_sc ? a.xh_sc[(size_t)g * D + 3 + (c - a.F)] : 0.f;
                else if (c == a.F + Fsc) val = a.t ? a.t[g] : t_all;
                else if (c < Fin) val = a.ctx[(size_t)g * a.C + (c - a.F - Fsc - 1)];
                v[k] = val;
            }
            a.HIN4[(size_t)gg * a.N + g] = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) {
        float s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, f[9];
        const float xi0 = xs[i], xi1 = xs[n + i], xi2 = xs[2 * n + i];
        const float mi = a.mask ? a.mask[o + i] : 1.f;
        for (int j = 0; j < n; ++j) {
            if (a.mask && (mi == 0.f || a.mask[o + j] == 0.f)) continue;       // no edge (i, j)
            frame_of(xi0, xi1, xi2, xs[j], xs[n + j], xs[2 * n + j], f);
#pragma unroll
            for (int r = 0; r < 9; ++r) s[r] += f[r];
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) a.FBAR[r * a.N + o + i] = mi != 0.f ? s[r] / cnt : 0.f;
    }
}

// ================================================================================================
// K2  edge geometry + edge embedding GCP2 (1,1)->(Se,Ve): one thread per edge
//     (edm_dataset.py:22-38, components/__init__.py:122-171, gcpnet.py:584-590)
// ================================================================================================
struct EdgeEmbedArgs {
    const float* X0; const float* XC; int N;
    const int* EROW; const int* ECOL; int E;
    const float* ws;    // scalar_out.weight [Se][1+Ve+9]
    const float* bs;    // [Se]
    const float* wd;    // vector_down.weight [Ve]  (Ve x 1)
    const float* wdf;   // vector_down_frames.weight [3]
    const float* kappa; // [Ve] = vector_up.weight @ vector_down.weight
    const float* wg;    // vector_out_scale.weight [Ve][Se]
    const float* bg;    // [Ve]
    v4f* EP4;           // [Se/4][E]   e' (SiLU'd scalar edge features)
    float* AL;          // [Ve][E]     xi'_c = AL[c] * U
    float* U;           // [3][E]      unit vector (0 for self loops)
    float* FR;          // [9][E]      frames
    // self-conditioning: the edge GCP2 is (2,2) -> (Se,Ve) (gcpnet.py:961-975): second scalar |x_sc_i - x_sc_j|^2, second vector u_sc;
    // scalar_out.weight rows are [e | e_sc | norms | q] (row stride `kin`), vector_down / _frames have a second column (wd1, wdf1), the
    // embedded vectors are rank 2: xi'_c = AL[c] * U + BL[c] * USC with kappa1 = vector_up @ vector_down[:, 1]
    int sc, kin; const float* X0SC; const float* wd1; const float* wdf1; const float* kappa1; float* BL; float* USC;
};

template <int SE, int VE>
__global__ __launch_bounds__(256) void k_edge_embed(EdgeEmbedArgs a) {
    const int eid = blockIdx.x * 256 + threadIdx.x;
    if (eid >= a.E) return;
    const int i = a.EROW[eid], j = a.ECOL[eid], N = a.N, E = a.E;
    // e = |x_i - x_j|^2, xi = unit vector, both from the UN-centralised positions (gcpnet.py:1102,1109)
    const float d0 = a.X0[i] - a.X0[j], d1 = a.X0[N + i] - a.X0[N + j], d2 = a.X0[2 * N + i] - a.X0[2 * N + j];
    const float es = d0 * d0 + d1 * d1 + d2 * d2;
    const float nr = sqrtf(es);
    float u[3] = {0.f, 0.f, 0.f};
    if (nr > 0.f) { u[0] = d0 / nr; u[1] = d1 / nr; u[2] = d2 / nr; }
    float f[9];
    frame_of(a.XC[i], a.XC[N + i], a.XC[2 * N + i], a.XC[j], a.XC[N + j], a.XC[2 * N + j], f);
#pragma unroll
    for (int r = 0; r < 9; ++r) a.FR[(size_t)r * E + eid] = f[r];
#pragma unroll
    for (int x = 0; x < 3; ++x) a.U[(size_t)x * E + eid] = u[x];
    float usc[3] = {0.f, 0.f, 0.f}, es_sc = 0.f;
    if (a.sc) {
        const float s0 = a.X0SC[i] - a.X0SC[j], s1 = a.X0SC[N + i] - a.X0SC[N + j], s2 = a.X0SC[2 * N + i] - a.X0SC[2 * N + j];
        es_sc = s0 * s0 + s1 * s1 + s2 * s2;
        const float ns = sqrtf(es_sc);
        if (ns > 0.f) { usc[0] = s0 / ns; usc[1] = s1 / ns; usc[2] = s2 / ns; }
#pragma unroll
        for (int x = 0; x < 3; ++x) a.USC[(size_t)x * E + eid] = usc[x];
    }
    constexpr int KIN = 1 + VE + 9;      // [e | norms | q]; the second edge scalar of the self-conditioning mode is kept apart (es_sc)
    float in[KIN];
    in[0] = es;
#pragma unroll
    for (int h = 0; h < VE; ++h) {
        const float w = a.wd[h], w1 = a.sc ? a.wd1[h] : 0.f;
        const float v0 = u[0] * w + usc[0] * w1, v1 = u[1] * w + usc[1] * w1, v2 = u[2] * w + usc[2] * w1;
        in[1 + h] = sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + 1e-8f) + 1e-8f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float w = a.wdf[k], w1 = a.sc ? a.wdf1[k] : 0.f;
        const float v0 = u[0] * w + usc[0] * w1, v1 = u[1] * w + usc[1] * w1, v2 = u[2] * w + usc[2] * w1;
#pragma unroll
        for (int r = 0; r < 3; ++r) in[1 + VE + 3 * k + r] = f[3 * r] * v0 + f[3 * r + 1] * v1 + f[3 * r + 2] * v2;
    }
    float gate[VE];
#pragma unroll
    for (int c = 0; c < VE; ++c) gate[c] = a.bg[c];
    const int ne = a.sc ? 2 : 1;         // columns of the edge scalars in scalar_out.weight
    for (int g = 0; g < SE / 4; ++g) {
        v4f o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = 4 * g + k;
            const float* w = a.ws + ch * a.kin;
            float p = a.bs[ch] + w[0] * es;
            if (a.sc) p += w[1] * es_sc;
#pragma unroll
            for (int q = 1; q < KIN; ++q) p += w[ne - 1 + q] * in[q];
            p = fast_silu(p);
            o[k] = p;
#pragma unroll
            for (int c = 0; c < VE; ++c) gate[c] += a.wg[c * SE + ch] * p;
        }
        a.EP4[(size_t)g * E + eid] = o;
    }
#pragma unroll
    for (int c = 0; c < VE; ++c) {
        const float sg = fast_sigmoid(gate[c]);
        a.AL[(size_t)c * E + eid] = a.kappa[c] * sg;
        if (a.sc) a.BL[(size_t)c * E + eid] = a.kappa1[c] * sg;
    }
}

// ================================================================================================
// K4  fused edge-message kernel: one workgroup = 64 consecutive flat edges, all four message GCP2s,
//     scalar message attention and the per-row segment sum (gcpnet.py:676-737)
// ================================================================================================
struct EdgeMsgArgs {
    // per-edge (constant over layers)
    const v4f* EP4; const float* AL; const float* U; const float* FR; const int* EROW; const int* ECOL;
    const float* BL; const float* USC;   // self-conditioning only (else null): second coefficient set / unit vector of the rank-2 embedded edge vectors
    const int* NCNT;  // [N] atoms in the node's molecule (row length)
    int E, N;
    // per-node, for this layer (written by the previous node kernel)
    const v4f* PQ4;   // [128][N]: groups 0..63 = W_i h_i + b, groups 64..127 = W_j h_j  (msg0 scalar_out split)
    const float* VDI; // [(H0+3)*3][N] rows hh*3+x: [W_down;W_frames][:, 0:V] chi_i
    const float* VDJ; // same with the column block of chi_j
    float* AGG;       // [N][352] rows that lie inside one tile
    float* PART;      // [tiles][2][352] partial sums of rows cut by tile boundaries (see AggSrc)
    // msg0
    const v4f* w0; int G0;       // packed [8][G0][64]; K' = [e'(Se) | n(H0 pad 4) | q(9 pad 12)] padded to 8
    const float* wddE;           // [(H0+3)][Ve]: columns V..V+Ve of [W_down; W_frames]
    const v4f* wg0; const float* bg0; const float* wup0;  // gate / vector_up [32][H0] of msg0
    // msg1..3
    GcpW mk[3];
    const float* wa; float ba;   // scalar_message_attention
    float* prof;                 // optional [tiles][8 waves][24] phase time stamps (shader cycles since kernel start)
};

// In-kernel phase time stamps are compiled in only with -DGCDM_STAMPS (tools/build_variants.sh; gcdm_profile_enable(h, 2 | 3) fails
// without it): even switched off at run time each stamp costs the wave an exec-mask save, a branch and a restore -- 20 of them are
// ~2 % of the edge kernel's issue slots, and they pin the instruction schedule around them.
// STAMP_END: the END-OF-TILE stamp alone is part of every build (one uniform branch per tile; the -DGCDM_STAMPS builds pin the instruction schedule at 20 places
// and differ from the shipped kernel by +-1.5 % tile cycles from build to build, profiles/r05_edge_kernel_cycles.md -- the shipped kernel is measured by this one)
#define STAMP_END(i)                                                                                    \
    do {                                                                                                \
        if (a.prof && lane == 0) a.prof[((size_t)prof_tile * 8 + wave) * 24 + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_start); \
    } while (0)
#ifdef GCDM_STAMPS
#define STAMP(i)                                                                                        \
    do {                                                                                                \
        if (a.prof && lane == 0) a.prof[((size_t)prof_tile * 8 + wave) * 24 + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_start); \
    } while (0)
#define GCDM_HAVE_STAMPS 1
#else
#define STAMP(i) ((void)0)
#define GCDM_HAVE_STAMPS 0
#endif

// Tile geometry.  T = 64: one 8-wave workgroup per CU (152 KB LDS).  T = 32: 4-wave workgroups of 77 KB, two per CU,
// which run out of phase so that one's VALU phases overlap the other's MFMA phases (weights are then streamed twice).
template <int T>
struct EdgeGeo {
    static constexpr int TP = T + 1;
    static constexpr int THREADS = (T == 64) ? 512 : 256;
    static constexpr int PARTS = THREADS / T;      // threads sharing one entity in the VALU phases (8)
    static constexpr int XS_GROUPS = 72;
    static constexpr int OFF_XS = 0;
    static constexpr int OFF_VV = OFF_XS + XS_GROUPS * TP * 16;
    static constexpr int OFF_VH = OFF_VV + 96 * TP * 4;
    static constexpr int OFF_PG = OFF_VH + 60 * TP * 4;
    static constexpr int OFF_FR = OFF_PG + 4 * 32 * TP * 4;
    static constexpr int OFF_META = OFF_FR + 9 * TP * 4;
    static constexpr int OFF_VHB = (OFF_META + (T + T + (T + 2) + T + 4) * 4 + 15) & ~15;   // split-precision kernel: hidden-vector images [3][3][T] x 16 B
    static constexpr int LDS_BYTES = OFF_VHB + 3 * 3 * T * 16;
    static constexpr int OFF_WAX = LDS_BYTES;                      // split-precision kernel (persistent): attention weights, staged once per workgroup
    static constexpr int OFF_US = OFF_WAX + GCDM_S * 4;            // split-precision kernel: unit vectors u of the tile's edges [3][TP], staged by one wave (round 5)
    static constexpr int LDS_BYTES_X3 = OFF_US + 3 * TP * 4;
    static_assert(T != 64 || LDS_BYTES_X3 <= 160 * 1024, "64-edge tile: one workgroup per CU, 160 KB of LDS");
};

template <int SE, int VE, int ET>
__global__ __launch_bounds__(EdgeGeo<ET>::THREADS) void k_edge_msg(EdgeMsgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Geo = EdgeGeo<ET>;
    constexpr int ETP = Geo::TP, EK_THREADS = Geo::THREADS, PARTS = Geo::PARTS;
    v4f* XS4 = (v4f*)(smem + Geo::OFF_XS);
    float* XSf = (float*)XS4;
    float* VV = (float*)(smem + Geo::OFF_VV);
    float* VH = (float*)(smem + Geo::OFF_VH);
    float* PG = (float*)(smem + Geo::OFF_PG);
    float* FR = (float*)(smem + Geo::OFF_FR);
    int* m_row = (int*)(smem + Geo::OFF_META);
    int* m_col = m_row + ET;
    int* m_seg = m_col + ET;   // [ET+2] segment starts (+ end sentinel)
    float* m_att = (float*)(m_seg + ET + 2);
    int* m_misc = (int*)(m_att + ET);  // [0] = nseg

    constexpr int H0 = (2 * GCDM_V + VE) / 4;       // bottleneck 4
    constexpr int H0G = (H0 + 3) / 4;
    constexpr int SEG = SE / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // VALU phases: entity e, and which of the PARTS co-operating threads this is (T = 64: entity = lane, part = wave)
    const int e = (ET == 64) ? lane : (tid & (ET - 1));
    const int part = (ET == 64) ? wave : (tid / ET);
    const int E = a.E, N = a.N;
    const int e0 = blockIdx.x * ET;
    const int nvalid = min(ET, E - e0);
    const int eid = min(e0 + e, E - 1);
    const int ni = a.EROW[eid], nj = a.ECOL[eid];
    [[maybe_unused]] const int prof_tile = blockIdx.x;
    const uint64_t t_start = a.prof ? __builtin_amdgcn_s_memtime() : 0;

    // ---- P0: tile metadata + row segments (wave 0) -------------------------------------------
    if (wave == 0) {
        const bool own = lane < ET;    // T = 32: lanes 32..63 of wave 0 are part 1 of the same entities
        if (own) { m_row[e] = ni; m_col[e] = nj; }
        const int prev = __shfl_up(ni, 1);
        const bool start = own && (e < nvalid) && (e == 0 || prev != ni);
        const unsigned long long mask = __ballot(start);
        const int sid = __popcll(mask & ((2ull << lane) - 1ull)) - 1;
        if (start) m_seg[sid] = e;
        if (lane == 0) {
            const int ns = __popcll(mask);
            m_seg[ns] = nvalid;
            m_misc[0] = ns;
        }
    }
    // ---- P1: msg0 pre-phase ---------------------------------------------------------------------
    {
        float fr[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) fr[r] = a.FR[(size_t)r * E + eid];
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < 9; ++r) FR[r * ETP + e] = fr[r];
        }
        for (int g = part; g < SEG; g += PARTS) XS4[g * ETP + e] = a.EP4[(size_t)g * E + eid];
        float al[VE];
#pragma unroll
        for (int c = 0; c < VE; ++c) al[c] = a.AL[(size_t)c * E + eid];
        const float u0 = a.U[eid], u1 = a.U[(size_t)E + eid], u2 = a.U[2 * (size_t)E + eid];
        constexpr int gN = SEG, gQ = SEG + H0G;
        constexpr int ROWS0 = H0 + 3, NH0 = (ROWS0 + PARTS - 1) / PARTS;
        // all node-side gathers of this thread's rows are issued before the first use (static unroll)
        float gi[NH0][3], gj[NH0][3], beta[NH0], beta2[NH0];
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
            beta2[i] = 0.f;
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        if (a.BL) {                      // self-conditioning: rank-2 embedded edge vectors, xi'_c = AL_c u + BL_c u_sc (uniform branch)
            s0 = a.USC[eid]; s1 = a.USC[(size_t)E + eid]; s2 = a.USC[2 * (size_t)E + eid];
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
#pragma unroll
        for (int i = 0; i < NH0; ++i) {
            const int hh = part + PARTS * i;
            const float vx = gi[i][0] + beta[i] * u0 + beta2[i] * s0 + gj[i][0];
            const float vy = gi[i][1] + beta[i] * u1 + beta2[i] * s1 + gj[i][1];
            const float vz = gi[i][2] + beta[i] * u2 + beta2[i] * s2 + gj[i][2];
            if (hh < H0) {
                XSf[((gN + (hh >> 2)) * ETP + e) * 4 + (hh & 3)] = sqrtf(vx * vx + vy * vy + vz * vz + 1e-8f) + 1e-8f;
                VH[(hh * 3 + 0) * ETP + e] = vx;
                VH[(hh * 3 + 1) * ETP + e] = vy;
                VH[(hh * 3 + 2) * ETP + e] = vz;
            } else if (hh < ROWS0) {
                const int k = hh - H0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int idx = 3 * k + r;
                    XSf[((gQ + (idx >> 2)) * ETP + e) * 4 + (idx & 3)] = fr[3 * r] * vx + fr[3 * r + 1] * vy + fr[3 * r + 2] * vz;
                }
            }
        }
        if (part == 0) {
            for (int hh = H0; hh < 4 * H0G; ++hh) XSf[((gN + (hh >> 2)) * ETP + e) * 4 + (hh & 3)] = 0.f;
            for (int idx = 9; idx < 12; ++idx) XSf[((gQ + (idx >> 2)) * ETP + e) * 4 + (idx & 3)] = 0.f;
            for (int g = gQ + 3; g < 2 * a.G0; ++g) XS4[g * ETP + e] = (v4f){0.f, 0.f, 0.f, 0.f};
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    const int half = lane >> 5, l31 = lane & 31;
    const int mt0 = 2 * (wave & 3);   // this wave's two M-tiles (64 output channels) ...
    const int col0 = 32 * (wave >> 2);  // ... of one N-tile (32 of the 64 edges); waves w and w+4 share the weight stream via L1
    f32x16 acc[2][1];
    f32x16 gacc[1];

    // ---- P2: msg0 GEMM.  acc starts from the node-level halves P_i + Q_j of scalar_out -----------
    {
        {
            const int ri = m_row[col0 + l31], cj = m_col[col0 + l31];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = 8 * (mt0 + m) + 2 * q + half;
                    const v4f p = a.PQ4[(size_t)g * N + ri];
                    const v4f qq = a.PQ4[(size_t)(64 + g) * N + cj];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[m][0][4 * q + t] = p[t] + qq[t];
                }
        }
        STAMP(3);
        tile_gemm<2, 1>(acc, a.w0 + (size_t)mt0 * a.G0 * 64, a.G0, XS4 + col0, ETP, lane);
        STAMP(4);
        apply_silu<2, 1>(acc);
        STAMP(5);
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[0][r] = 0.f;
        gate_partial<2, 1>(gacc, acc, a.wg0, mt0, lane);
        store_gate_partial<1>(PG, gacc, ETP, wave & 3, lane, col0);
        STAMP(6);
    }
    __syncthreads();
    STAMP(7);
    // ---- P3: m.s = silu(p) ; m.v = (W_up vh) * sigmoid(gate) -------------------------------------
    store_state<2, 1, false>(XS4, 0, acc, ETP, mt0, lane, col0);
    vec_finish<ET, H0, EK_THREADS>(PG, a.bg0, a.wup0, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
        VV[(c * 3 + 0) * ETP + e] = ox;
        VV[(c * 3 + 1) * ETP + e] = oy;
        VV[(c * 3 + 2) * ETP + e] = oz;
    });
    STAMP(8);
    __syncthreads();
    STAMP(9);

    // ---- residual message GCP2s k = 1..3 (gcpnet.py:698-701) ------------------------------------
    for (int k = 0; k < 3; ++k) {
        const GcpW& w = a.mk[k];
        gcp2_pre<ET, 8, GCDM_V, EK_THREADS>(w.wdd, VV, 0, FR, XSf, GCDM_SG, GCDM_SG + 2, 2 * w.G, VH, e, part);
        if (k == 0) STAMP(10);
        __syncthreads();
        if (k == 0) STAMP(11);
        acc_init_bias<2, 1>(acc, w.b, mt0, lane);
        tile_gemm<2, 1>(acc, w.w + (size_t)mt0 * w.G * 64, w.G, XS4 + col0, ETP, lane);
        if (k == 0) STAMP(12);
        apply_silu<2, 1>(acc);
        if (k == 0) STAMP(13);
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[0][r] = 0.f;
        gate_partial<2, 1>(gacc, acc, w.wg, mt0, lane);
        store_gate_partial<1>(PG, gacc, ETP, wave & 3, lane, col0);
        if (k == 0) STAMP(14);
        __syncthreads();
        if (k == 0) STAMP(15);
        store_state<2, 1, true>(XS4, 0, acc, ETP, mt0, lane, col0);
        vec_finish<ET, 8, EK_THREADS>(PG, w.bg, w.wup, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
            VV[(c * 3 + 0) * ETP + e] += ox;
            VV[(c * 3 + 1) * ETP + e] += oy;
            VV[(c * 3 + 2) * ETP + e] += oz;
        });
        if (k == 0) STAMP(16);
        __syncthreads();
        if (k == 0) STAMP(17);
    }
    STAMP(18);

    // ---- scalar message attention (gcpnet.py:709-711): att = sigmoid(w_a . m.s + b_a) -----------
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
    // ---- aggregation: agg_i = sum_j [m.s * att | m.v]  over the row segments of this tile --------
    {
        const int nseg = m_misc[0];
        constexpr int UNITS = GCDM_SG + 3 * GCDM_V;  // 64 float4 scalar groups + 96 vector floats
        for (int wk = tid; wk < nseg * UNITS; wk += EK_THREADS) {
            const int sg = wk / UNITS, un = wk - sg * UNITS;
            const int st = m_seg[sg], en = m_seg[sg + 1];
            const int node = m_row[st];
            const bool whole = (en - st) == a.NCNT[node];
            float* dst = whole ? a.AGG + (size_t)node * GCDM_AGGW : a.PART + ((size_t)blockIdx.x * 2 + (st == 0 ? 0 : 1)) * GCDM_AGGW;
            if (un < GCDM_SG) {
                v4f s = {0.f, 0.f, 0.f, 0.f};
                for (int x = st; x < en; ++x) s += XS4[un * ETP + x] * m_att[x];
                *(v4f*)(dst + 4 * un) = s;
            } else {
                const int r = un - GCDM_SG;
                float s = 0.f;
                for (int x = st; x < en; ++x) s += VV[r * ETP + x];
                dst[GCDM_S + r] = s;
            }
        }
    }
    STAMP(20);
}

// ================================================================================================
// K3/K5  node kernels: one workgroup = 32 consecutive nodes.
//   EMBED : node embedding GCP2 (gcpnet.py:591-597)                      -> h, chi
//   LAYER : feed-forward GCP2 + residual + position-update GCP2 (gcpnet.py:894-928) -> h, chi, x
//   then either the node-level halves of the NEXT layer's msg0 (PQ4, VDI, VDJ) or, after the last layer,
//   the scalar projection GCP2 + vel (gcpnet.py:1191-1216).
// ================================================================================================
struct NodeArgs {
    int N, F, C, FinG, Dout;   // Dout = 3 + F
    float pos_weight;
    // embed inputs
    const v4f* HIN4; const float* CHI0;
    GcpW emb;
    // layer inputs
    AggSrc agg;
    GcpW ff; GcpW pos;
    // shared state
    v4f* H4;        // [64][N]
    float* CHI;     // [96][N]
    float* XC;      // [3][N]
    const float* X0;
    const float* FBAR;
    // next-layer pre (has_next) ...
    int has_next;
    const v4f* wpq; const float* bpq;           // packed [16][32][64], bias [512]
    const float* wddI; const float* wddJ; int H0; // [(H0+3)][32] each
    v4f* PQ4; float* VDI; float* VDJ;
    // ... or projection (last layer)
    GcpW proj;
    float* OUT;     // [N][Dout] : columns 3.. get h_final
    float* VEL;     // [3][N]
    uint32_t* flags_dev;
    const float* mask;   // masked nodes (or null): h, chi and x of a masked node are zeroed after every interaction layer (gcpnet.py:914-928)
};

constexpr int NT_ = 32, NTP = 33;
constexpr int NK_XS_GROUPS = 140;
constexpr int NK_OFF_XS = 0;
constexpr int NK_OFF_VV = NK_OFF_XS + NK_XS_GROUPS * NTP * 16;
constexpr int NK_OFF_VH = NK_OFF_VV + 192 * NTP * 4;
constexpr int NK_OFF_PG = NK_OFF_VH + 96 * NTP * 4;
constexpr int NK_OFF_FR = NK_OFF_PG + 4 * 32 * NTP * 4;
constexpr int NK_OFF_XP = NK_OFF_FR + 9 * NTP * 4;
constexpr int NK_LDS_BYTES = NK_OFF_XP + 3 * NTP * 4;

// VIN0: input vectors of the node embedding (2 orientations; 4 with self-conditioning, gcpnet.py:966-970)
template <bool EMBED, int VIN0 = 2>
__global__ __launch_bounds__(256) void k_node(NodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f* XS4 = (v4f*)(smem + NK_OFF_XS);
    float* XSf = (float*)XS4;
    float* VV = (float*)(smem + NK_OFF_VV);
    float* VH = (float*)(smem + NK_OFF_VH);
    float* PG = (float*)(smem + NK_OFF_PG);
    float* FR = (float*)(smem + NK_OFF_FR);
    float* XP = (float*)(smem + NK_OFF_XP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = tid & 31, part = tid >> 5;   // T = 32: 8 threads share an entity
    const int half = lane >> 5, l31 = lane & 31;
    const int N = a.N;
    const int n0 = blockIdx.x * NT_;
    const int nid = min(n0 + e, N - 1);
    const bool valid = (n0 + e) < N;
    constexpr int HB = 64;   // group base of h inside XS4 (LAYER: groups 0..63 hold agg.s)
    constexpr int CB = 32;   // channel base of chi inside VV (LAYER: channels 0..31 hold agg.v)

    // ---- load tile --------------------------------------------------------------------------------
    for (int r = part; r < 9; r += 8) FR[r * NTP + e] = a.FBAR[(size_t)r * N + nid];
    if (part < 3) XP[part * NTP + e] = a.XC[(size_t)part * N + nid];
    f32x16 acc[2][1];
    f32x16 gacc[1];
    const int mt0 = 2 * wave;

    if (EMBED) {
        for (int g = part; g < a.FinG; g += 8) XS4[g * NTP + e] = a.HIN4[(size_t)g * N + nid];
        for (int r = part; r < 3 * VIN0; r += 8) VV[r * NTP + e] = a.CHI0[(size_t)r * N + nid];
        __syncthreads();
        const GcpW& w = a.emb;
        const int gN = a.FinG, gQ = gN + (w.H + 3) / 4;
        gcp2_pre<NT_, 32, VIN0>(w.wdd, VV, 0, FR, XSf, gN, gQ, 2 * w.G, VH, e, part);
        __syncthreads();
        acc_init_bias<2, 1>(acc, w.b, mt0, lane);
        tile_gemm<2, 1>(acc, w.w + (size_t)mt0 * w.G * 64, w.G, XS4, NTP, lane);
        // nonlinearities (None, None): h = p, the vector gate sees p (gcpnet.py:539, 410)
        for (int r = 0; r < 16; ++r) gacc[0][r] = 0.f;
        gate_partial<2, 1>(gacc, acc, w.wg, mt0, lane);
        store_gate_partial<1>(PG, gacc, NTP, wave, lane);
        __syncthreads();
        store_state<2, 1, false>(XS4, HB, acc, NTP, mt0, lane);
        vec_finish<NT_, 32>(PG, w.bg, w.wup, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
            VV[((CB + c) * 3 + 0) * NTP + e] = ox;
            VV[((CB + c) * 3 + 1) * NTP + e] = oy;
            VV[((CB + c) * 3 + 2) * NTP + e] = oz;
        });
        __syncthreads();
    } else {
        // agg (node-major rows of 352 floats): one wave reads a node's 64 scalar groups + 96 vector floats coalesced
        for (int x = wave; x < NT_; x += 4) {
            const int nd = min(n0 + x, N - 1);
            const AggRow src = agg_row(a.agg, nd);
            XS4[lane * NTP + x] = agg_load4(src, 4 * lane);
            VV[lane * NTP + x] = agg_load1(src, GCDM_S + lane);
            if (lane < 32) VV[(64 + lane) * NTP + x] = agg_load1(src, GCDM_S + 64 + lane);
        }
        for (int g = part; g < GCDM_SG; g += 8) XS4[(HB + g) * NTP + e] = a.H4[(size_t)g * N + nid];
        for (int r = part; r < 96; r += 8) VV[(CB * 3 + r) * NTP + e] = a.CHI[(size_t)r * N + nid];
        __syncthreads();
        // ---- feed-forward GCP2: ([agg.s | h], [agg.v ; chi]) -> (S, V), scalar_out = Linear-SiLU-Linear
        {
            const GcpW& w = a.ff;
            const int gN = 2 * GCDM_SG, gQ = gN + (w.H + 3) / 4;
            gcp2_pre<NT_, 16, 2 * GCDM_V>(w.wdd, VV, 0, FR, XSf, gN, gQ, 2 * w.G, VH, e, part);
            __syncthreads();
            acc_init_bias<2, 1>(acc, w.b, mt0, lane);
            tile_gemm<2, 1>(acc, w.w + (size_t)mt0 * w.G * 64, w.G, XS4, NTP, lane);
            apply_silu<2, 1>(acc);
            __syncthreads();                                   // all waves done reading groups 0..63 (agg.s)
            store_state<2, 1, false>(XS4, 0, acc, NTP, mt0, lane);  // hidden activations overwrite agg.s
            __syncthreads();
            acc_init_bias<2, 1>(acc, w.b2, mt0, lane);
            tile_gemm<2, 1>(acc, w.w2 + (size_t)mt0 * 32 * 64, 32, XS4, NTP, lane);
            for (int r = 0; r < 16; ++r) gacc[0][r] = 0.f;
            gate_partial<2, 1>(gacc, acc, w.wg, mt0, lane);    // nonlinearities (None, None)
            store_gate_partial<1>(PG, gacc, NTP, wave, lane);
            __syncthreads();
            store_state<2, 1, true>(XS4, HB, acc, NTP, mt0, lane);  // h <- h + ff.s (gcpnet.py:907)
            const float me = a.mask ? a.mask[nid] : 1.f;            // masked nodes: h, chi, x <- 0 after the layer (gcpnet.py:914-928)
            vec_finish<NT_, 16>(PG, w.bg, w.wup, GCDM_V, VH, e, part, [&](int c, float ox, float oy, float oz) {
                VV[((CB + c) * 3 + 0) * NTP + e] = (VV[((CB + c) * 3 + 0) * NTP + e] + ox) * me;
                VV[((CB + c) * 3 + 1) * NTP + e] = (VV[((CB + c) * 3 + 1) * NTP + e] + oy) * me;
                VV[((CB + c) * 3 + 2) * NTP + e] = (VV[((CB + c) * 3 + 2) * NTP + e] + oz) * me;
            });
            __syncthreads();
            if (a.mask) {
                for (int g = part; g < GCDM_SG; g += 8) XS4[(HB + g) * NTP + e] *= me;
                __syncthreads();
            }
        }
        // ---- position update GCP2: (h, chi) -> (S, 1); x += v[0] * weight (gcpnet.py:834-857, 922-928)
        {
            const GcpW& w = a.pos;
            const int gN = 2 * GCDM_SG, gQ = gN + (w.H + 3) / 4;
            gcp2_pre<NT_, 8, GCDM_V>(w.wdd, VV, CB, FR, XSf, gN, gQ, HB + 2 * w.G, VH, e, part);
            __syncthreads();
            acc_init_bias<2, 1>(acc, w.b, mt0, lane);
            tile_gemm<2, 1>(acc, w.w + (size_t)mt0 * w.G * 64, w.G, XS4 + HB * NTP, NTP, lane);
            apply_silu<2, 1>(acc);
            for (int r = 0; r < 16; ++r) gacc[0][r] = 0.f;
            gate_partial<2, 1>(gacc, acc, w.wg, mt0, lane);
            store_gate_partial<1>(PG, gacc, NTP, wave, lane);
            __syncthreads();
            const float me = a.mask ? a.mask[nid] : 1.f;
            vec_finish<NT_, 8>(PG, w.bg, w.wup, 1, VH, e, part, [&](int c, float ox, float oy, float oz) {
                XP[0 * NTP + e] = (XP[0 * NTP + e] + ox * a.pos_weight) * me;
                XP[1 * NTP + e] = (XP[1 * NTP + e] + oy * a.pos_weight) * me;
                XP[2 * NTP + e] = (XP[2 * NTP + e] + oz * a.pos_weight) * me;
            });
            __syncthreads();
            if (part < 3 && valid) a.XC[(size_t)part * N + nid] = XP[part * NTP + e];
        }
    }
    // ---- write the node state back (h, chi) -------------------------------------------------------
    if (valid) {
        for (int g = part; g < GCDM_SG; g += 8) a.H4[(size_t)g * N + nid] = XS4[(HB + g) * NTP + e];
        for (int r = part; r < 96; r += 8) a.CHI[(size_t)r * N + nid] = VV[(CB * 3 + r) * NTP + e];
    }

    if (a.has_next) {
        // ---- node-level halves of the next layer's msg0 --------------------------------------------
        // PQ = [W_s[:, :S] h + b ; W_s[:, S+Se:2S+Se] h]  (scalar_out column split)
        f32x16 pacc[4][1];
        const int pmt0 = 4 * wave;
        acc_init_bias<4, 1>(pacc, a.bpq, pmt0, lane);
        tile_gemm<4, 1>(pacc, a.wpq + (size_t)pmt0 * 32 * 64, 32, XS4 + HB * NTP, NTP, lane);
        if ((n0 + l31) < N) {
            const int nd = n0 + l31;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int g = 8 * (pmt0 + m) + 2 * q + half;
                    a.PQ4[(size_t)g * N + nd] = (v4f){pacc[m][0][4 * q], pacc[m][0][4 * q + 1], pacc[m][0][4 * q + 2], pacc[m][0][4 * q + 3]};
                }
        }
        // VDI / VDJ = [W_down; W_frames][:, block] chi   (3 x (H0+3) per node and side)
        if (valid) {
            const int rows = a.H0 + 3;
            for (int it = part; it < 2 * rows; it += 8) {
                const int side = it >= rows, hh = side ? it - rows : it;
                const float* w = (side ? a.wddJ : a.wddI) + hh * GCDM_V;
                const float* vp = VV + (CB * 3) * NTP + e;
                float vx = 0.f, vy = 0.f, vz = 0.f;
                for (int c = 0; c < GCDM_V; ++c) {
                    const float wc = w[c];
                    vx += wc * vp[0]; vy += wc * vp[NTP]; vz += wc * vp[2 * NTP];
                    vp += 3 * NTP;
                }
                float* dst = (side ? a.VDJ : a.VDI) + (size_t)(hh * 3) * N + nid;
                dst[0] = vx; dst[N] = vy; dst[2 * (size_t)N] = vz;
            }
        }
    } else {
        // ---- scalar projection GCP2 (S, V) -> (F+1+C, 0), bottleneck 1, no activation (gcpnet.py:1191-1197)
        const GcpW& w = a.proj;
        const int gN = 2 * GCDM_SG, gQ = gN + (w.H + 3) / 4;
        gcp2_pre<NT_, 32, GCDM_V>(w.wdd, VV, CB, FR, XSf, gN, gQ, HB + 2 * w.G, VH, e, part);
        __syncthreads();
        if (wave == 0) {
            f32x16 qacc[1][1];
            acc_init_bias<1, 1>(qacc, w.b, 0, lane);
            tile_gemm<1, 1>(qacc, w.w, w.G, XS4 + HB * NTP, NTP, lane);
            if ((n0 + l31) < N) {
                float* dst = a.OUT + (size_t)(n0 + l31) * a.Dout + 3;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < a.F) dst[c] = qacc[0][0][r];   // context and time columns are dropped (gcpnet.py:1208-1211)
                }
            }
        }
        // vel = x - x_init (un-centralised x_init, gcpnet.py:1204); NaN -> flag (whole batch zeroed by k_finish)
        if (part < 3 && valid) {
            const float v = XP[part * NTP + e] - a.X0[(size_t)part * N + nid];
            a.VEL[(size_t)part * N + nid] = v;
            if (v != v) atomicOr(a.flags_dev, 1u);
        }
    }
}

// ================================================================================================
// K6  finish: per molecule CoM projection of vel, NaN guard, assemble net_out (gcpnet.py:1213-1230)
// ================================================================================================
struct FinishArgs {
    const float* VEL; const int* noff; int N, Dout; float* OUT; const uint32_t* flags_dev; uint32_t* user_flags;
    const float* mask;      // masked nodes (or null): vel is zero there and they are left out of the centroid (components/__init__.py:53-92)
};

__global__ __launch_bounds__(64) void k_finish(FinishArgs a) {
    const int b = blockIdx.x, o = a.noff[b], n = a.noff[b + 1] - o;
    const uint32_t fl = *a.flags_dev;
    const bool nan = (fl & 1u) != 0;
    if (fl && a.user_flags && b == 0 && threadIdx.x == 0) atomicOr(a.user_flags, fl);   // NaN-vel / f16-range bits
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (!nan) {
        float cnt = 0.f;
        for (int i = 0; i < n; ++i) {
            const float mk = a.mask ? a.mask[o + i] : 1.f;
            m0 += a.VEL[o + i] * mk; m1 += a.VEL[a.N + o + i] * mk; m2 += a.VEL[2 * (size_t)a.N + o + i] * mk;
            cnt += mk;
        }
        m0 /= cnt; m1 /= cnt; m2 /= cnt;
    }
    for (int i = threadIdx.x; i < n; i += 64) {
        float* dst = a.OUT + (size_t)(o + i) * a.Dout;
        const float mk = a.mask ? a.mask[o + i] : 1.f;
        dst[0] = nan ? 0.f : (a.VEL[o + i] - m0) * mk;
        dst[1] = nan ? 0.f : (a.VEL[a.N + o + i] - m1) * mk;
        dst[2] = nan ? 0.f : (a.VEL[2 * (size_t)a.N + o + i] - m2) * mk;
    }
}

// ================================================================================================
// K7  sampler kernels (variational_diffusion.py:795-907, 1204-1278): one workgroup per molecule
// ================================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// standard normal #col of node `node` for draw `draw` (Box-Muller on Philox4x32-10)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t draw, uint32_t node, uint32_t col) {
    uint32_t c[4] = {node, draw, col >> 1, 0x6c078965u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1));
    float s, co;
    __sincosf(6.283185307179586f * u2, &s, &co);
    return (col & 1) ? r * s : r * co;
}

struct StepArgs {
    const StepRow* rows; const int* cursor;   // captured step: alpha_coef / c_eps / sigma / draw come from rows[*cursor] (else null)
    int* cursor_rw;      // captured step: the cursor's two slots {read by k_prep, read by k_sample}; `cursor` above points at slot 1 (else null)
    // the network's last stage folded in (k_finish: CoM-free velocities; flags to the caller): the x columns of eps are VEL - mean(VEL) per molecule,
    // computed here and also written to eps.  Null -> eps is complete already (gcdm_forward's own k_finish ran)
    const float* VEL;
    float* z;            // [N][D] in/out
    float* z_out;        // step / init: where the new latent goes (null = in place)
    uint32_t node_base;  // flat index of node 0 in the whole batch (Philox counter), for plans that are slices of a flat batch
    const float* gmean;  // fix_noise (variational_diffusion.py:832-834, 1323-1325): x-noise is centred over the WHOLE flat batch -> its mean [3], else null
    const float* eps;    // [N][D] network output (null for init)
    const float* noise;  // [N][D] or null -> Philox
    const int* noff; int N, D;
    float alpha_coef, c_eps, sigma;     // step: z/alpha_coef - c_eps*eps + sigma*noise ; final: alpha_coef*(z - c_eps*eps) + sigma*noise
    uint64_t seed; uint32_t draw;
    int mode;            // 0 = step, 1 = init (z = noise), 2 = final decode, 3 = forward noising alpha_coef*z + sigma*noise, 4 = the same + zero-CoM projection
    // final decode
    float* out; int num_atom_types, include_charges; float nv0, nv1, nv2, nb1, nb2;
    uint32_t* user_flags; uint32_t* flags_dev;
};

// the step cursor of a captured step, two slots: k_prep reads slot 0 and copies it to slot 1, k_sample reads slot 1 and writes slot 0 = slot 1 - 1
// (every reader of a slot is in another launch than its writer: stream order is the only synchronisation)
__global__ void k_cursor_set(int* cursor, int v) { cursor[0] = v; cursor[1] = v; }

// whole-batch CoG re-projection if any molecule drifted (variational_diffusion.py:1389-1402)
__global__ __launch_bounds__(64) void k_cog_fix(float* out, const int* noff, int D, const uint32_t* flags_dev, uint32_t* user_flags) {
    if (!(*flags_dev & 4u)) return;
    const int b = blockIdx.x, o = noff[b], n = noff[b + 1] - o;
    if (user_flags && b == 0 && threadIdx.x == 0) atomicOr(user_flags, 4u);
    float m[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) { m[0] += out[(size_t)(o + i) * D]; m[1] += out[(size_t)(o + i) * D + 1]; m[2] += out[(size_t)(o + i) * D + 2]; }
    m[0] /= (float)n; m[1] /= (float)n; m[2] /= (float)n;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) {
        float* d = out + (size_t)(o + i) * D;
        d[0] -= m[0]; d[1] -= m[1]; d[2] -= m[2];
    }
}

// normalize (variational_diffusion.py:702-732) of caller-supplied samples [x | one-hot | charge] into z, plus the statistics of
// assert_mean_zero_with_mask (:465-474): stat[0] = max_b |sum_i z_x|, stat[1] = max |z_x| (as float bit patterns: both >= 0).
struct EncodeArgs {
    const float* xh; float* z; const int* noff; int D, num_atom_types, include_charges;
    float nv0, nv1, nv2, nb1, nb2;
    uint32_t* stat;
};

__global__ __launch_bounds__(64) void k_encode(EncodeArgs a) {
    const int b = blockIdx.x, o = a.noff[b], n = a.noff[b + 1] - o, D = a.D;
    float s[3] = {0.f, 0.f, 0.f}, mx = 0.f;
    for (int idx = threadIdx.x; idx < n * D; idx += 64) {
        const int i = idx / D, c = idx - i * D;
        const size_t gi = (size_t)(o + i) * D + c;
        const float v = a.xh[gi];
        float r;
        if (c < 3) { r = v / a.nv0; mx = fmaxf(mx, fabsf(r)); s[c] += r; }
        else if (c < 3 + a.num_atom_types) r = (v - a.nb1) / a.nv1;
        else r = (v - a.nb2) / a.nv2;
        a.z[gi] = r;
    }
    // D is not a multiple of 3 in general, so a lane sees all three coordinates: reduce each over the wave
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        s[0] += __shfl_xor(s[0], sh); s[1] += __shfl_xor(s[1], sh); s[2] += __shfl_xor(s[2], sh);
        mx = fmaxf(mx, __shfl_xor(mx, sh));
    }
    if (threadIdx.x == 0) {
        atomicMax(a.stat, __float_as_uint(fmaxf(fabsf(s[0]), fmaxf(fabsf(s[1]), fabsf(s[2])))));
        atomicMax(a.stat + 1, __float_as_uint(mx));
    }
}

// unnormalize_z (variational_diffusion.py:735-792): continuous frame of the chain visualisation, no argmax / rounding
__global__ __launch_bounds__(256) void k_unnormalize(const float* __restrict__ z, float* __restrict__ out, int N, int D, int num_atom_types,
                                                     float nv0, float nv1, float nv2, float nb1, float nb2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * D) return;
    const int c = idx % D;
    const float v = z[idx];
    out[idx] = c < 3 ? v * nv0 : (c < 3 + num_atom_types ? v * nv1 + nb1 : v * nv2 + nb2);
}

__global__ void k_mean_flag(const uint32_t* stat, uint32_t* user_flags) {
    const float err = __uint_as_float(stat[0]), largest = __uint_as_float(stat[1]);
    if (!(err / (largest + 1e-10f) < 1e-2f)) atomicOr(user_flags, 2u);      // GCDM_FLAG_MEAN_NOT_ZERO
}

// fix_noise: mean of the x-part of one noise draw over all N nodes, in a fixed order (one workgroup; thread i takes nodes i, i+1024, ...)
__global__ __launch_bounds__(1024) void k_noise_mean(const float* __restrict__ noise, uint64_t seed, uint32_t draw, uint32_t node_base, int N, int D, float* __restrict__ gmean) {
    __shared__ float red[3][1024];
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < N; i += 1024)
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += noise ? noise[(size_t)i * D + c] : philox_normal(seed, draw, node_base + (uint32_t)i, (uint32_t)c);
#pragma unroll
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
#pragma unroll
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 3) gmean[threadIdx.x] = red[threadIdx.x][0] / (float)N;
}

// RePaint inpainting (variational_diffusion.py:1582-1789).  One wave per molecule; `fixed` marks the nodes taken from the known molecule.
// mean over the fixed nodes of a molecule of the x-part of p (0 if it has none); every lane returns the same value
__device__ inline void fixed_mean(const float* __restrict__ p, const uint8_t* __restrict__ fixed, int o, int n, int D, float m[3]) {
    float s[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
    for (int i = threadIdx.x; i < n; i += 64)
        if (fixed[o + i]) {
            const float* r = p + (size_t)(o + i) * D;
            s[0] += r[0]; s[1] += r[1]; s[2] += r[2]; cnt += 1.f;
        }
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) {
        s[0] += __shfl_xor(s[0], sh); s[1] += __shfl_xor(s[1], sh); s[2] += __shfl_xor(s[2], sh); cnt += __shfl_xor(cnt, sh);
    }
    const float inv = 1.f / fmaxf(cnt, 1.f);
    m[0] = s[0] * inv; m[1] = s[1] * inv; m[2] = s[2] * inv;
}

// the known molecule, shifted so that its fixed nodes have zero CoM (:1625-1633)
__global__ __launch_bounds__(64) void k_inpaint_center(const float* __restrict__ xh, const uint8_t* __restrict__ fixed, const int* __restrict__ noff,
                                                       int D, float* __restrict__ out) {
    const int b = blockIdx.x, o = noff[b], n = noff[b + 1] - o;
    float m[3];
    fixed_mean(xh, fixed, o, n, D, m);
    for (int idx = threadIdx.x; idx < n * D; idx += 64) {
        const int c = idx % D;
        const size_t gi = (size_t)o * D + idx;
        out[gi] = xh[gi] - (c < 3 ? m[c] : 0.f);
    }
}

// z = fixed ? z_known + (CoM_fixed(z_unknown) - CoM_fixed(z_known)) : z_unknown   (:1678-1702)
__global__ __launch_bounds__(64) void k_inpaint_combine(const float* __restrict__ zk, const float* __restrict__ zu, const uint8_t* __restrict__ fixed,
                                                        const int* __restrict__ noff, int D, float* __restrict__ z) {
    const int b = blockIdx.x, o = noff[b], n = noff[b + 1] - o;
    float mk[3], mu[3];
    fixed_mean(zk, fixed, o, n, D, mk);
    fixed_mean(zu, fixed, o, n, D, mu);
    for (int idx = threadIdx.x; idx < n * D; idx += 64) {
        const int i = idx / D, c = idx - i * D;
        const size_t gi = (size_t)o * D + idx;
        z[gi] = fixed[o + i] ? zk[gi] + (c < 3 ? mu[c] - mk[c] : 0.f) : zu[gi];
    }
}

__global__ __launch_bounds__(64) void k_sample(StepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ns[];  // [n][D] noise, then results
    const int b = blockIdx.x, o = a.noff[b], n = a.noff[b + 1] - o, D = a.D;
    if (a.rows) {
        const StepRow r = a.rows[*a.cursor];
        a.alpha_coef = r.alpha_coef; a.c_eps = r.c_eps; a.sigma = r.sigma; a.draw = r.draw;
    }
    // k_finish's arithmetic, in its order (plans of the sampler are unmasked: mk = 1, cnt = n)
    bool vnan = false;
    float vm[3] = {0.f, 0.f, 0.f};
    if (a.VEL) {
        const uint32_t fl = *a.flags_dev & ~4u;       // (bit 2 = CoG drift is this launch's own, raised by other workgroups below: k_cog_fix reports it)
        vnan = (fl & 1u) != 0;
        if (fl && a.user_flags && b == 0 && threadIdx.x == 0) atomicOr(a.user_flags, fl);
        // (through LDS: parallel loads, then the serial sum every thread repeats; ns holds >= 3 n floats)
        for (int i = threadIdx.x; i < n; i += 64) { ns[i] = a.VEL[o + i]; ns[n + i] = a.VEL[a.N + o + i]; ns[2 * n + i] = a.VEL[2 * (size_t)a.N + o + i]; }
        __syncthreads();
        if (!vnan) {
            float cnt = 0.f;
            for (int i = 0; i < n; ++i) { vm[0] += ns[i]; vm[1] += ns[n + i]; vm[2] += ns[2 * n + i]; cnt += 1.f; }
            vm[0] /= cnt; vm[1] /= cnt; vm[2] /= cnt;
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < n * D; idx += 64) {
        const int i = idx / D, c = idx - i * D;
        ns[idx] = a.noise ? a.noise[(size_t)(o + i) * D + c] : philox_normal(a.seed, a.draw, a.node_base + (uint32_t)(o + i), (uint32_t)c);
    }
    __syncthreads();
    // CoM-free x-noise (sample_center_gravity_zero_gaussian_with_mask, :396-420)
    float m[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) { m[0] += ns[i * D]; m[1] += ns[i * D + 1]; m[2] += ns[i * D + 2]; }
    m[0] /= (float)n; m[1] /= (float)n; m[2] /= (float)n;
    if (a.gmean) { m[0] = a.gmean[0]; m[1] = a.gmean[1]; m[2] = a.gmean[2]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * D; idx += 64) {
        const int i = idx / D, c = idx - i * D;
        float e = ns[idx];
        if (c < 3) e -= m[c];
        float v;
        if (a.mode == 1) {
            v = e;
        } else {
            const size_t gi = (size_t)(o + i) * D + c;
            float ep = 0.f;
            if (a.mode < 3) {
                if (a.VEL && c < 3) {
                    ep = vnan ? 0.f : a.VEL[(size_t)c * a.N + o + i] - vm[c];
                    const_cast<float*>(a.eps)[gi] = ep;
                } else {
                    ep = a.eps[gi];
                }
            }
            // mu = z / alpha_ts - (sigma2_ts / alpha_ts / sigma_t) * eps ; zs = mu + sigma * noise   (:1247-1263)
            // final: mu = 1/alpha_0 * (z0 - sigma_0 * eps) ; xh = mu + sigma_x * noise          (:571, :878-886)
            // forward noising: q(z_t | x, h) = alpha_t * xh + sigma_t * noise (:922-929) and q(z_t | z_s) (:1174-1185)
            if (a.mode >= 3) v = a.alpha_coef * a.z[gi] + a.sigma * e;
            else
            v = (a.mode == 0) ? (a.z[gi] / a.alpha_coef - a.c_eps * ep) + a.sigma * e
                              : a.alpha_coef * (a.z[gi] - a.c_eps * ep) + a.sigma * e;
        }
        ns[idx] = v;
    }
    __syncthreads();
    if (a.mode == 0 || a.mode == 4) {  // project x back to zero CoM (:1266-1277, :1187-1194)
        float s[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) { s[0] += ns[i * D]; s[1] += ns[i * D + 1]; s[2] += ns[i * D + 2]; }
        s[0] /= (float)n; s[1] /= (float)n; s[2] /= (float)n;
        float* zo = a.z_out ? a.z_out : a.z;
        for (int idx = threadIdx.x; idx < n * D; idx += 64) {
            const int i = idx / D, c = idx - i * D;
            zo[(size_t)(o + i) * D + c] = ns[idx] - (c < 3 ? s[c] : 0.f);
        }
    } else if (a.mode == 1 || a.mode == 3) {
        float* zo = a.z_out ? a.z_out : a.z;
        for (int idx = threadIdx.x; idx < n * D; idx += 64) {
            const int i = idx / D, c = idx - i * D;
            zo[(size_t)(o + i) * D + c] = ns[idx];
        }
    } else {
        // unnormalize (:735-757), argmax one-hot / rounded charge (:902-905), CoG drift re-projection (:1389-1402)
        float s[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) { s[0] += ns[i * D] * a.nv0; s[1] += ns[i * D + 1] * a.nv0; s[2] += ns[i * D + 2] * a.nv0; }
        const bool drift = fmaxf(fabsf(s[0]), fmaxf(fabsf(s[1]), fabsf(s[2]))) > 5e-2f;
        if (drift && threadIdx.x == 0) atomicOr(a.flags_dev, 4u);
        for (int i = threadIdx.x; i < n; i += 64) {
            float* dst = a.out + (size_t)(o + i) * D;
            const float* src = ns + i * D;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = src[c] * a.nv0;
            int best = 0;
            float bv = src[3] * a.nv1 + a.nb1;
            for (int c = 1; c < a.num_atom_types; ++c) {
                const float v = src[3 + c] * a.nv1 + a.nb1;
                if (v > bv) { bv = v; best = c; }
            }
            for (int c = 0; c < a.num_atom_types; ++c) dst[3 + c] = (c == best) ? 1.f : 0.f;
            if (a.include_charges) dst[3 + a.num_atom_types] = rintf(src[3 + a.num_atom_types] * a.nv2 + a.nb2);
        }
    }
    // captured step: this launch reads slot 1 of the cursor (k_prep copied slot 0 there), so ONE thread may step slot 0 for the next launch of the graph
    // while the other workgroups are still on their way in (what a k_cursor_dec node did before round 6; no atomics, no fence)
    if (a.cursor_rw && b == 0 && threadIdx.x == 0) a.cursor_rw[0] = a.cursor_rw[1] - 1;
}
