// gcdm_api.hip -- host side of libgcdm_hip.so: C ABI (include/gcdm_hip.h), weight re-packing into the MFMA
// fragment layout, batch plan, kernel launches.  gfx950 only; no torch types, no exceptions across the ABI.
#include "gcdm_kernels.hip.h"
#include "gcdm_edge_x3.hip.h"
#include "gcdm_node_x3.hip.h"
#include "gcdm_node_x3w.hip.h"
#include "gcdm_layer_x3.hip.h"
#ifndef GCDM_FUSE_TILE_DEFAULT
#define GCDM_FUSE_TILE_DEFAULT 32
#endif
#include "gcdm_embed_x3.hip.h"
#include "gcdm_stability.hip.h"
#include "../../include/gcdm_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct LayerDev {
    // edge kernel
    const v4f* w0; int G0; const float* wddE; const v4f* wg0; const float* bg0; const float* wup0;
    GcpW mk[3];
    const float* wa; float ba;
    // node kernel
    GcpW ff, pos;
    const v4f* wpq; const float* bpq; const float* wddI; const float* wddJ;
    // split-precision (f16 x3) images of the edge-kernel GEMM weights
    const h8 *w0H, *w0L, *wg0H, *wg0L, *wH[3], *wL[3], *wgH[3], *wgL[3], *wbeH, *wbeL;
    const h8 *vpH[3], *vpL[3], *vf1[3], *vf2[3], *vf0H, *vf0L;   // vector path on the matrix pipe (gcdm_edge_x3.hip.h)
    const h8 *vdH, *vdL;
    const float *bpqx, *wax;       // scaled units of the split-precision edge kernel: c * bias of the msg0 node halves, attention weights / c
    int KB0, KB;
    GcpX3 ffx, posx;
    const h8 *wpqH, *wpqL;
};

}  // namespace

struct gcdm_handle {
    GcdmConfig cfg{};
    std::string err;
    std::map<std::string, std::vector<float>> host_w;
    bool finalized = false;
    // derived dims
    int F = 0, C = 0, Fin = 0, FinG = 0, D = 0, Se = 0, Ve = 0, L = 0, H0 = 0;
    int sc = 0, FinP = 0;            // self-conditioning: Fin = [h | h_sc | t | ctx] feeds the node embedding, FinP = F + 1 + C is what the projection returns
    const float *ee_wd1 = nullptr, *ee_wdf1 = nullptr, *ee_kappa1 = nullptr;
    const h8 *ee_xwH = nullptr, *ee_xwL = nullptr, *ee_xgH = nullptr, *ee_xgL = nullptr;     // split-precision A operands of k_edge_embed_x3
    float *X0SC = nullptr, *BL = nullptr, *USC = nullptr;
    // device weights
    float* wpool = nullptr;
    size_t wpool_bytes = 0;
    std::vector<LayerDev> layers;
    GcpW emb{}, proj{};
    GcpX3 embx{}, projx{};
    const float *ee_ws = nullptr, *ee_bs = nullptr, *ee_wd = nullptr, *ee_wdf = nullptr, *ee_kappa = nullptr, *ee_wg = nullptr, *ee_bg = nullptr;
    std::vector<float> gamma;
    // plan
    int B = 0, N = 0, max_n = 0;
    int64_t E = 0;
    int *d_noff = nullptr, *d_erow = nullptr, *d_ecol = nullptr, *d_ncnt = nullptr, *d_rowstart = nullptr;
    float* d_mask = nullptr;         // 1 / 0 per node when the plan has masked nodes (gcdm_plan_batch_masked), else null
    float* ws = nullptr;  // workspace pool
    size_t ws_floats = 0;
    float *X0 = nullptr, *XC = nullptr, *FBAR = nullptr, *CHI0 = nullptr, *HIN4 = nullptr, *H4 = nullptr, *CHI = nullptr, *PQ4 = nullptr,
          *VDI = nullptr, *VDJ = nullptr, *AGG = nullptr, *PART = nullptr, *VEL = nullptr, *EPS = nullptr, *EP4 = nullptr, *AL = nullptr, *U = nullptr, *FR = nullptr, *PROF = nullptr, *ZK = nullptr, *ZU = nullptr, *ZROW = nullptr;
    float *PQ4b = nullptr, *VDIb = nullptr, *VDJb = nullptr;   // second set of the node-level msg0 halves: layer l gathers set l & 1, its node tiles write set (l + 1) & 1 (round 6:
                                                               // with the node tiles as a tail role of the edge workgroups both happen in ONE launch)
    // Fused layer launch (gcdm_layer_x3.hip.h): tables of the node-tile queue for 32- and 64-node tiles ([0] / [1]); null when the plan does not qualify
    int* d_tail_tab[2] = {nullptr, nullptr};       // qstart[9] | rel_node_end[8] | need[node tiles]
    int* d_tail_ctr = nullptr;                     // ready[node tiles of 32] | qcur[17]  (zeroed at plan time, self-resetting)
    int tail_tiles32 = 0;
    int fuse_node = 1;               // option "fuse_node" / env GCDM_FUSE_NODE: 1 = the layer's node tiles run as a tail role of the persistent edge workgroups (one launch
                                     // per layer) where the plan qualifies; 0 = two launches per layer (rounds 1-5)
    int fuse_tile = 0;               // option "fuse_tile": nodes per node tile of the tail role (32 / 64; 0 = automatic)
    int fuse_active = 0;             // option "fuse_active" (read-only): the last forward used the fused launch
    uint32_t* d_flags = nullptr;
    float* d_gmean = nullptr;
    int flat_prev = 0, flat_next = 0;   // the plan is a slice of a larger flat batch (options "flat_prev" / "flat_next"; include/gcdm_hip.h)
    uint32_t node_base = 0;             // option "node_base": flat index of the slice's first node (Philox counter)
    int fix_noise = 0;                  // option "fix_noise": x-noise centred over the whole flat batch (mol_gen_sample(fix_noise=True))
    int cog_fix = 1;                 // gcdm_sample_final re-projects drifting centres of gravity (off for chain frames, reference :1389)
    int layer_limit = -1;
    int edge_tile = 0;               // 64: one 8-wave workgroup per CU; 32: two 4-wave workgroups per CU; 0: automatic (env GCDM_EDGE_TILE)
    // automatic choice, measured on MI355X: 64 for both kernel families and every batch size since the persistent kernel with
    // next-tile prefetch (round 3) -- round 4, 64 vs 32 edges per tile, QM9 64 / 100 / 200 molecules: 71 / 84 / 114 against 68 / 78 / 100 molecules/s,
    // 1024 molecules +2 %, GEOM 32 molecules 45.5 against 38.3.  Rows cut by tile boundaries are summed from per-tile partials in tile order
    // (AggSrc), so every choice is bit-reproducible for any molecule size.
    int tile() const { return edge_tile ? edge_tile : 64; }
    int cus = 256;                   // compute units of the device (persistent edge-message workgroups: one per CU)
    int persistent = 1;              // option "persistent" / env GCDM_PERSISTENT=0: one workgroup per tile (round 2 schedule; A/B runs)
    int node_tile = 0;               // option "node_tile" / env GCDM_NODE_TILE: nodes per workgroup of the split-precision layer node kernel (64: k_node_x3w,
                                     // 32: k_node_x3, 0 = automatic: whichever needs less time for this plan's node count, see node_tile_for)
    bool x3_weights_ok = true;       // every GEMM weight fits the split-precision images at some exponent split k <= X3_MAX_SHIFT; else mfma_mode 1 is refused
    int x3_shift = 0;                // k: packed weights carry 2^(11-k), activation images 2^(k-11) (X3Const, gcdm_edge_x3.hip.h)
    X3Const x3c() const { const float w = ldexpf(1.0f, 11 - x3_shift); return X3Const{1.0f / w, w, 1.0f / w, 6.0e4f * w}; }
    int mfma_x3 = 1;                 // requested mode -- 1: split-precision f16 x3 kernels (default; env GCDM_MFMA=f16x3|f32), 0: fp32 MFMA
    bool use_x3() const { return mfma_x3 && x3_weights_ok; }   // effective mode: models whose weights do not fit the split images run fp32 MFMA
    bool attr_set = false;
    // profiling (HIP events around the k_edge_msg launches of one forward)
    bool profile = false;
    bool profile_phases = false;     // enable == 2: in-kernel phase time stamps (debug_read("phase"))
    bool profile_node = false;       // enable == 3: phase time stamps of the layer node kernel (debug_read("phase_node"))
    std::vector<hipEvent_t> ev;      // 2 per layer
    int ev_used = 0;
    // One denoise step as a hipGraph (option "step_graph", default on).  A step is ~25 launches (QM9) whose arguments are the same at every
    // step except four scalars and the Philox draw index; those live in a device table (StepRow) indexed by a device cursor, so ONE instantiated
    // graph serves all steps of a sample: gcdm_sample_step enqueues one graph launch instead of 25 kernels (host cost 0.5 ms -> ~0.03 ms per
    // step; what small batches and several slices / batches in flight are bound by).  Same kernels, same arguments: bit-identical results.
    int step_graph = 1;
    bool step_graph_failed = false;  // capture / instantiation failed once on this handle: launch directly from then on
    bool capturing = false;
    struct StepGraph {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        const float* z = nullptr; const float* ctx = nullptr; const uint32_t* flags = nullptr;
        uint64_t seed = 0; int num_steps = 0;
    } sg;
    hipStream_t cap_stream = nullptr;  // capture happens here (the caller's stream may be the null stream, which cannot capture)
    hipEvent_t sg_done = nullptr; bool sg_inflight = false;   // recorded behind every graph launch: an exec (and the row table it reads) is retired only after it
    hipStream_t sg_last_stream = nullptr;                      // ... and a launch on ANOTHER stream first waits for it, so that the one event covers every launch in flight
    StepRow* d_rows = nullptr; int rows_steps = 0;
    int* d_cursor = nullptr; int cursor_expected = -1;
    int64_t graph_launches = 0;        // option "graph_launches" (read-only): steps served by the graph since the handle was created
};

namespace {

int fail(gcdm_handle* h, const std::string& msg) {
    if (h) h->err = msg;
    return -1;
}

// Anything a captured step has baked in changed (plan, weights, gamma table, an option): drop the instantiated graph
// (the last launch of the exec may still be in flight -- the sampler changes options right behind its last step, a restart switches the matrix mode
//  with ~25 launches queued: HIP does not promise to defer the destruction of an exec that is executing, so wait for the event recorded behind it)
void wait_step_graph_idle(gcdm_handle* h) {
    if (h->sg_inflight && h->sg_done) (void)hipEventSynchronize(h->sg_done);
    h->sg_inflight = false;
}
void drop_step_graph(gcdm_handle* h) {
    if (!h) return;
    if (h->sg.exec || h->sg.graph) wait_step_graph_idle(h);
    if (h->sg.exec) (void)hipGraphExecDestroy(h->sg.exec);
    if (h->sg.graph) (void)hipGraphDestroy(h->sg.graph);
    h->sg = gcdm_handle::StepGraph{};
    h->rows_steps = 0;
    h->cursor_expected = -1;
}

#define HIP_OK(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return fail(h, std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)

// Binds the handle's device for the duration of an entry point and restores the caller's current device afterwards: a handle on
// cuda:1 used while device 0 is current must neither launch on the wrong device nor change the caller's (torch's) current device.
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- pool builder: 16-byte aligned sub-arrays of one device allocation ---------------------------
struct Pool {
    std::vector<float> host;
    size_t add(const std::vector<float>& a) {
        size_t off = (host.size() + 3) & ~size_t(3);
        host.resize(off + a.size(), 0.f);
        std::memcpy(host.data() + off, a.data(), a.size() * sizeof(float));
        return off;
    }
};

struct Dense {
    int M, K;
    std::vector<float> a;
    Dense(int m, int k) : M(m), K(k), a((size_t)m * k, 0.f) {}
    float& at(int m, int k) { return a[(size_t)m * K + k]; }
};

// packed[mt][g][lane][t] = W[32 mt + (lane & 31)][8 g + 4 (lane >> 5) + t]
std::vector<float> pack_mfma(Dense& W) {
    const int MT = W.M / 32, G = W.K / 8;
    std::vector<float> out((size_t)W.M * W.K);
    for (int mt = 0; mt < MT; ++mt)
        for (int g = 0; g < G; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < 4; ++t)
                    out[(((size_t)mt * G + g) * 64 + lane) * 4 + t] = W.at(32 * mt + (lane & 31), 8 * g + 4 * (lane >> 5) + t);
    return out;
}

// split-precision images: x = hi + 2^-11 lo', both f16.  packed[mt][kb][lane][8] = W[32 mt + (lane & 31)][16 kb + 8 (lane >> 5) + s]
// Packed split-precision weights carry a factor 2^11 (exact): the kernels keep every activation image pre-multiplied by 2^-11 (X3_PRE,
// gcdm_edge_x3.hip.h), which moves the f16 overflow bound of the activations from 6.5e4 to 1.2e8 at no cost in accuracy (f16
// denormals are honoured by the MFMA, tools/mfma_denorm.hip); weights stay representable while |W| < 31.9.
thread_local float g_split_absmax = 0.f;      // largest |W| seen by split_f16 since gcdm_finalize_weights reset it (NaN counts as too large)
thread_local float g_split_w = 2048.0f;       // 2^(11-k) of the handle being packed
constexpr int X3_MAX_SHIFT = 6;               // k <= 6: |W| < 2047 and the lo' images of ordinary weights (>= 1e-4) stay normal f16 numbers
void split_f16(float x, uint16_t& hi, uint16_t& lo) {
    g_split_absmax = (fabsf(x) <= g_split_absmax) ? g_split_absmax : (x == x ? fabsf(x) : INFINITY);
    x *= g_split_w;
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)((x - (float)h) * g_split_w);
    std::memcpy(&hi, &h, 2);
    std::memcpy(&lo, &l, 2);
}

void pack_x3(Dense& W, std::vector<float>& outH, std::vector<float>& outL) {
    const int MT = W.M / 32, KB = W.K / 16;
    std::vector<uint16_t> H((size_t)W.M * W.K), L((size_t)W.M * W.K);
    for (int mt = 0; mt < MT; ++mt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const size_t o = (((size_t)mt * KB + kb) * 64 + lane) * 8 + s;
                    split_f16(W.at(32 * mt + (lane & 31), 16 * kb + 8 * (lane >> 5) + s), H[o], L[o]);
                }
    outH.assign(H.size() / 2, 0.f);
    outL.assign(L.size() / 2, 0.f);
    std::memcpy(outH.data(), H.data(), H.size() * 2);
    std::memcpy(outL.data(), L.data(), L.size() * 2);
}

// gate weights for the in-register contraction: block (mt, j) slot s of lane l  <->  channel 32 mt + 16 j + 8 (s >> 2) + 4 (l >> 5) + (s & 3)
void pack_gate_x3(Dense& Wg /*[32][256]*/, std::vector<float>& outH, std::vector<float>& outL) {
    std::vector<uint16_t> H((size_t)8 * 2 * 64 * 8), L(H.size());
    for (int mt = 0; mt < 8; ++mt)
        for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const size_t o = (((size_t)mt * 2 + j) * 64 + lane) * 8 + s;
                    split_f16(Wg.at(lane & 31, 32 * mt + 16 * j + 8 * (s >> 2) + 4 * (lane >> 5) + (s & 3)), H[o], L[o]);
                }
    outH.assign(H.size() / 2, 0.f);
    outL.assign(L.size() / 2, 0.f);
    std::memcpy(outH.data(), H.data(), H.size() * 2);
    std::memcpy(outL.data(), L.data(), L.size() * 2);
}

struct WView {
    const std::vector<float>* v;
    int rows, cols;
    float at(int r, int c) const { return (*v)[(size_t)r * cols + c]; }
};

// Scaled units of the split-precision edge kernel (gcdm_edge_x3.hip.h): the message scalars live as c * m.s with c = -log2(e), so that
// SiLU(x) * c = x' / (1 + exp2(x')) for x' = c * x -- one multiply less per element.  Host side: scalar_out weights acting on true-unit
// inputs (and biases) carry c, weights acting on the scaled message scalars whose output is in true units carry 1 / c.
// (the constant itself: X3_C in gcdm_edge_x3.hip.h)

// ---- A operands of the vector path (v_mfma_f32_16x16x32_f16: lane l holds row l & 15, k = 8 (l >> 4) + j), see gcdm_edge_x3.hip.h ----
std::vector<float> f16_words(const std::vector<uint16_t>& v) {
    std::vector<float> o(v.size() / 2);
    std::memcpy(o.data(), v.data(), v.size() * 2);
    return o;
}

// [W_down (H = 8 rows); W_frames (3 rows)] x 32 channels.  MFMA row 4q + i = hidden vector 3q + i (i < 3, real while < 8) or frame vector q
// (i = 3, q < 3), so that every lane class q of the D layout runs the same code; k = 8q + j <-> channel (j < 4 ? 4q + j : 16 + 4q + j - 4)
template <typename WD, typename WF>
void pack_vec_pre(const WD& wd, const WF& wdf, std::vector<float>& outH, std::vector<float>& outL) {
    std::vector<uint16_t> H(64 * 8), L(64 * 8);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int r = lane & 15, rq = r >> 2, ri = r & 3, q = lane >> 4, c = j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4);
            float wv = 0.f;
            if (ri < 3) { if (3 * rq + ri < 8) wv = wd.at(3 * rq + ri, c); }
            else if (rq < 3) wv = wdf.at(rq, c);
            split_f16(wv, H[lane * 8 + j], L[lane * 8 + j]);
        }
    outH = f16_words(H);
    outL = f16_words(L);
}

// vector_up [32][8] against the B image [hi(3) 0 | lo'(3) 0] of hidden channels 3q .. 3q+2:  A1 = [W_hi | 0],  A2 = [W_lo' | W_hi]
template <typename WU>
void pack_vec_fin(const WU& wu, std::vector<float>& out1, std::vector<float>& out2) {
    std::vector<uint16_t> A1(2 * 64 * 8, 0), A2(2 * 64 * 8, 0);
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane) {
            const int c = 16 * m + (lane & 15), q = lane >> 4;
            for (int j = 0; j < 8; ++j) {
                const int hch = 3 * q + (j & 3);
                if ((j & 3) == 3 || hch >= 8) continue;
                uint16_t hi, lo;
                split_f16(wu.at(c, hch), hi, lo);
                const size_t o = ((size_t)m * 64 + lane) * 8 + j;
                A1[o] = j < 4 ? hi : 0;
                A2[o] = j < 4 ? lo : hi;
            }
        }
    out1 = f16_words(A1);
    out2 = f16_words(A2);
}

// generic small-M matrix W [R][K] as 16x16x32 A operands in natural K order: packed [ceil(R/16)][K/32][64 lanes] x 8 f16
// (node kernels: vecmat_mfma, gcdm_node_x3.hip.h)
void pack_vecmat(const std::vector<float>& W, int R, int K, std::vector<float>& outH, std::vector<float>& outL) {
    const int MT = (R + 15) / 16, KBn = K / 32;
    std::vector<uint16_t> H((size_t)MT * KBn * 64 * 8, 0), L(H.size(), 0);
    for (int m = 0; m < MT; ++m)
        for (int kb = 0; kb < KBn; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r = 16 * m + (lane & 15), c = 32 * kb + 8 * (lane >> 4) + j;
                    const size_t o = (((size_t)m * KBn + kb) * 64 + lane) * 8 + j;
                    split_f16(r < R ? W[(size_t)r * K + c] : 0.f, H[o], L[o]);
                }
    outH = f16_words(H);
    outL = f16_words(L);
}

// vector_up of msg0 [32][H0], k = hidden channel (H0 <= 32)
template <typename WU>
void pack_vec_fin0(const WU& wu, int H0, std::vector<float>& outH, std::vector<float>& outL) {
    std::vector<uint16_t> H(2 * 64 * 8, 0), L(2 * 64 * 8, 0);
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * m + (lane & 15), hch = 8 * (lane >> 4) + j;
                if (hch < H0) split_f16(wu.at(c, hch), H[((size_t)m * 64 + lane) * 8 + j], L[((size_t)m * 64 + lane) * 8 + j]);
            }
    outH = f16_words(H);
    outL = f16_words(L);
}

bool get_w(gcdm_handle* h, const std::string& key, int rows, int cols, WView& out) {
    auto it = h->host_w.find(key);
    if (it == h->host_w.end()) {
        h->err = "missing weight: " + key;
        return false;
    }
    if ((int64_t)it->second.size() != (int64_t)rows * cols) {
        h->err = "bad shape for " + key + ": expected " + std::to_string(rows) + "x" + std::to_string(cols) + " got numel " +
                 std::to_string(it->second.size());
        return false;
    }
    out = WView{&it->second, rows, cols};
    return true;
}

std::vector<float> padded(const WView& w, int n) {
    std::vector<float> o(n, 0.f);
    for (int i = 0; i < w.rows * w.cols && i < n; ++i) o[i] = (*w.v)[i];
    return o;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Pending pointer fix-ups (offsets into the pool until it is uploaded)
struct GcpOff {
    size_t w = 0, b = 0, w2 = 0, b2 = 0, wdd = 0, wg = 0, bg = 0, wup = 0;
    bool has_w2 = false, has_gate = false;
    int G = 0, H = 0, V_in = 0, V_out = 0;
    size_t xwH = 0, xwL = 0, xw2H = 0, xw2L = 0, xwgH = 0, xwgL = 0;   // split-precision images
    size_t xvmH = 0, xvmL = 0; bool has_vm = false;
    int KB = 0;
};

// generic GCP2 (scalar input = s_in channels laid out in 4-groups starting at K' = 0)
bool build_gcp(gcdm_handle* h, Pool& pool, const std::string& pre, int s_in, int v_in, int s_out, int v_out, int bottleneck,
               bool ff, GcpOff& o) {
    const int H = bottleneck > 1 ? v_in / bottleneck : std::max(v_in, v_out);
    const int SIg = (s_in + 3) / 4, Hg = (H + 3) / 4;
    const int Kp = round_up(4 * (SIg + Hg + 3), 8), Mp = round_up(s_out, 32);
    WView ws, bs, wd, wdf;
    const std::string so = ff ? "scalar_out.0." : "scalar_out.";
    if (!get_w(h, pre + so + "weight", s_out, s_in + H + 9, ws) || !get_w(h, pre + so + "bias", 1, s_out, bs) ||
        !get_w(h, pre + "vector_down.weight", H, v_in, wd) || !get_w(h, pre + "vector_down_frames.weight", 3, v_in, wdf))
        return false;
    Dense W(Mp, Kp);
    for (int m = 0; m < s_out; ++m) {
        for (int k = 0; k < s_in; ++k) W.at(m, k) = ws.at(m, k);
        for (int k = 0; k < H; ++k) W.at(m, 4 * SIg + k) = ws.at(m, s_in + k);
        for (int k = 0; k < 9; ++k) W.at(m, 4 * (SIg + Hg) + k) = ws.at(m, s_in + H + k);
    }
    o.w = pool.add(pack_mfma(W));
    o.b = pool.add(padded(bs, Mp));
    o.G = Kp / 8; o.H = H; o.V_in = v_in; o.V_out = v_out;
    {   // split-precision image: K' = [s_in -> 8-groups | n (H -> 8-groups) | q (9 -> 16)] padded to a multiple of 16
        const int S8 = (s_in + 7) / 8, H8 = (H + 7) / 8;
        const int Kx = round_up(8 * (S8 + H8 + 2), 16);
        Dense Wx(Mp, Kx);
        for (int m = 0; m < s_out; ++m) {
            for (int k = 0; k < s_in; ++k) Wx.at(m, k) = ws.at(m, k);
            for (int k = 0; k < H; ++k) Wx.at(m, 8 * S8 + k) = ws.at(m, s_in + k);
            for (int k = 0; k < 9; ++k) Wx.at(m, 8 * (S8 + H8) + k) = ws.at(m, s_in + H + k);
        }
        std::vector<float> xh, xl;
        pack_x3(Wx, xh, xl);
        o.xwH = pool.add(xh); o.xwL = pool.add(xl); o.KB = Kx / 16;
    }
    std::vector<float> dd((size_t)(H + 3) * v_in);
    for (int r = 0; r < H; ++r) for (int c = 0; c < v_in; ++c) dd[(size_t)r * v_in + c] = wd.at(r, c);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < v_in; ++c) dd[(size_t)(H + r) * v_in + c] = wdf.at(r, c);
    o.wdd = pool.add(dd);
    if (v_in % 32 == 0) {            // vector pre-phase on the matrix pipe (node kernels)
        std::vector<float> a, b;
        pack_vecmat(dd, H + 3, v_in, a, b);
        o.xvmH = pool.add(a); o.xvmL = pool.add(b); o.has_vm = true;
    }
    if (ff) {
        WView w2, b2;
        if (!get_w(h, pre + "scalar_out.2.weight", s_out, s_out, w2) || !get_w(h, pre + "scalar_out.2.bias", 1, s_out, b2)) return false;
        Dense W2(s_out, s_out);
        for (int m = 0; m < s_out; ++m) for (int k = 0; k < s_out; ++k) W2.at(m, k) = w2.at(m, k);
        o.w2 = pool.add(pack_mfma(W2));
        o.b2 = pool.add(padded(b2, s_out));
        o.has_w2 = true;
        std::vector<float> xh, xl;
        pack_x3(W2, xh, xl);
        o.xw2H = pool.add(xh); o.xw2L = pool.add(xl);
    }
    if (v_out) {
        WView wg, bg, wu;
        if (!get_w(h, pre + "vector_out_scale.weight", v_out, s_out, wg) || !get_w(h, pre + "vector_out_scale.bias", 1, v_out, bg) ||
            !get_w(h, pre + "vector_up.weight", v_out, H, wu))
            return false;
        Dense Wg(32, s_out);
        for (int m = 0; m < v_out; ++m) for (int k = 0; k < s_out; ++k) Wg.at(m, k) = wg.at(m, k);
        o.wg = pool.add(pack_mfma(Wg));
        o.bg = pool.add(padded(bg, 32));
        o.wup = pool.add(padded(wu, v_out * H));
        o.has_gate = true;
        if (s_out == 256) {
            std::vector<float> xh, xl;
            pack_gate_x3(Wg, xh, xl);
            o.xwgH = pool.add(xh); o.xwgL = pool.add(xl);
        }
    }
    return true;
}

GcpW resolve(const GcpOff& o, const float* base) {
    GcpW g{};
    g.w = (const v4f*)(base + o.w);
    g.b = base + o.b;
    g.w2 = o.has_w2 ? (const v4f*)(base + o.w2) : nullptr;
    g.b2 = o.has_w2 ? base + o.b2 : nullptr;
    g.wdd = base + o.wdd;
    g.wg = o.has_gate ? (const v4f*)(base + o.wg) : nullptr;
    g.bg = o.has_gate ? base + o.bg : nullptr;
    g.wup = o.has_gate ? base + o.wup : nullptr;
    g.G = o.G; g.H = o.H; g.V_in = o.V_in; g.V_out = o.V_out;
    return g;
}

GcpX3 resolve_x3(const GcpOff& o, const float* base) {
    GcpX3 g{};
    g.wH = (const h8*)(base + o.xwH); g.wL = (const h8*)(base + o.xwL); g.KB = o.KB;
    g.w2H = o.has_w2 ? (const h8*)(base + o.xw2H) : nullptr; g.w2L = o.has_w2 ? (const h8*)(base + o.xw2L) : nullptr;
    g.wgH = o.has_gate ? (const h8*)(base + o.xwgH) : nullptr; g.wgL = o.has_gate ? (const h8*)(base + o.xwgL) : nullptr;
    g.vmH = o.has_vm ? (const h8*)(base + o.xvmH) : nullptr; g.vmL = o.has_vm ? (const h8*)(base + o.xvmL) : nullptr;
    return g;
}

struct LayerOff {
    size_t w0, wddE, wg0, bg0, wup0, wa, wpq, bpq, wddI, wddJ, bpqx, wax;
    int G0;
    float ba;
    GcpOff mk[3], ff, pos;
    size_t w0H, w0L, wg0H, wg0L, wH[3], wL[3], wgH[3], wgL[3], wbeH, wbeL;
    size_t vpH[3], vpL[3], vf1[3], vf2[3], vf0H, vf0L, vdH, vdL;
    size_t wpqH, wpqL;
    int KB0, KB;
};

void free_plan(gcdm_handle* h) {
    if (h->d_noff) (void)hipFree(h->d_noff);
    if (h->d_erow) (void)hipFree(h->d_erow);
    if (h->d_ecol) (void)hipFree(h->d_ecol);
    if (h->d_ncnt) (void)hipFree(h->d_ncnt);
    if (h->d_rowstart) (void)hipFree(h->d_rowstart);
    if (h->d_mask) (void)hipFree(h->d_mask);
    h->d_mask = nullptr;
    for (int i = 0; i < 2; ++i) { if (h->d_tail_tab[i]) (void)hipFree(h->d_tail_tab[i]); h->d_tail_tab[i] = nullptr; }
    if (h->d_tail_ctr) (void)hipFree(h->d_tail_ctr);
    h->d_tail_ctr = nullptr;
    if (h->ws) (void)hipFree(h->ws);
    h->d_noff = h->d_erow = h->d_ecol = h->d_ncnt = h->d_rowstart = nullptr;
    h->ws = nullptr;
    h->B = h->N = 0;
    h->E = 0;
}

template <typename K>
int set_lds_attr(gcdm_handle* h, K kernel, int bytes) {
    HIP_OK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return 0;
}

}  // namespace

extern "C" {

int gcdm_create(const GcdmConfig* cfg, gcdm_handle** out) {
    if (!cfg || !out) return -1;
    gcdm_handle* h = new (std::nothrow) gcdm_handle();
    if (!h) return -1;
    *out = h;
    h->cfg = *cfg;
    if (cfg->abi_version != GCDM_ABI_VERSION) return fail(h, "abi_version mismatch");
    if (cfg->h_hidden_dim != GCDM_S || cfg->chi_hidden_dim != GCDM_V) return fail(h, "only h_hidden_dim=256, chi_hidden_dim=32 are built");
    if (cfg->bottleneck != 4) return fail(h, "only bottleneck=4 is built");
    if (!cfg->condition_on_time) return fail(h, "condition_on_time must be true");
    if (cfg->num_layers < 1 || cfg->num_layers > 64) return fail(h, "num_layers must be in 1..64");
    if (cfg->num_atom_types < 1 || cfg->num_context < 0 || cfg->num_timesteps < 1) return fail(h, "bad num_atom_types / num_context / num_timesteps");
    if (!((cfg->e_hidden_dim == 64 && cfg->xi_hidden_dim == 16) || (cfg->e_hidden_dim == 16 && cfg->xi_hidden_dim == 8)))
        return fail(h, "edge dims must be (64,16) [QM9] or (16,8) [GEOM]");
    h->F = cfg->num_atom_types + (cfg->include_charges ? 1 : 0);
    h->C = cfg->num_context;
    h->sc = cfg->self_condition ? 1 : 0;
    h->FinP = h->F + 1 + h->C;
    h->Fin = h->FinP + (h->sc ? h->F : 0);
    // the projection returns FinP scalars in one 32-row M-tile; the embedding only contracts over Fin (K dimension)
    if (h->FinP > 32 || h->Fin > 64) return fail(h, "too many node input features (projection <= 32, embedding inputs <= 64)");
    h->FinG = (h->Fin + 3) / 4;
    h->D = 3 + h->F;
    h->Se = cfg->e_hidden_dim;
    h->Ve = cfg->xi_hidden_dim;
    h->L = cfg->num_layers;
    h->H0 = (2 * GCDM_V + h->Ve) / 4;
    if (const char* et = getenv("GCDM_EDGE_TILE")) h->edge_tile = (atoi(et) == 32) ? 32 : 64;
    if (const char* mm = getenv("GCDM_MFMA")) h->mfma_x3 = (std::strcmp(mm, "f16x3") == 0) ? 1 : 0;
    if (const char* pe = getenv("GCDM_PERSISTENT")) h->persistent = atoi(pe) ? 1 : 0;
    if (const char* nt = getenv("GCDM_NODE_TILE")) h->node_tile = atoi(nt) == 32 ? 32 : atoi(nt) == 64 ? 64 : 0;
    if (const char* sg = getenv("GCDM_STEP_GRAPH")) h->step_graph = atoi(sg) ? 1 : 0;
    if (const char* fz = getenv("GCDM_FUSE_NODE")) h->fuse_node = atoi(fz) ? 1 : 0;
    if (const char* fz = getenv("GCDM_FUSE_TILE")) { const int v = atoi(fz); if (v == 32) h->fuse_tile = v; }
    DeviceGuard guard(cfg->device);
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && n > 0) h->cus = n;
    }
    HIP_OK(h, hipMalloc(&h->d_flags, 4 * sizeof(uint32_t)));            // [0] flag word, [1..2] statistics of gcdm_encode_samples
    HIP_OK(h, hipMemset(h->d_flags, 0, 4 * sizeof(uint32_t)));
    return 0;
}

int gcdm_destroy(gcdm_handle* h) {
    if (!h) return 0;
    DeviceGuard guard(h->cfg.device);
    free_plan(h);
    for (auto& e : h->ev) (void)hipEventDestroy(e);
    if (h->wpool) (void)hipFree(h->wpool);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->d_gmean) (void)hipFree(h->d_gmean);
    drop_step_graph(h);
    if (h->d_rows) (void)hipFree(h->d_rows);
    if (h->d_cursor) (void)hipFree(h->d_cursor);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->sg_done) (void)hipEventDestroy(h->sg_done);
    delete h;
    return 0;
}

const char* gcdm_last_error(const gcdm_handle* h) { return h ? h->err.c_str() : "null handle"; }

int gcdm_set_weight(gcdm_handle* h, const char* key, const float* host_data, int64_t numel) {
    if (!h || !key || !host_data || numel <= 0) return fail(h, "gcdm_set_weight: bad argument");
    h->host_w[key] = std::vector<float>(host_data, host_data + numel);
    h->finalized = false;
    return 0;
}

int gcdm_set_gamma(gcdm_handle* h, const float* host_gamma, int64_t numel) {
    if (!h || !host_gamma || numel != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_set_gamma: expected num_timesteps+1 values");
    h->gamma.assign(host_gamma, host_gamma + numel);
    drop_step_graph(h);
    return 0;
}

// One packing pass with the exponent split k (packed weights carry 2^(11-k), X3Const).  Leaves in g_split_absmax the largest magnitude that
// actually went into an f16 image -- matrices AND what the host folds into them (X3_C factors, the scalar_out biases that ride as the weight
// column of the constant-1 slot) -- but NOT the parameters that stay fp32 (node-level biases, vector_up tables, ...).
static int finalize_pass(gcdm_handle* h, int k_shift) {
    const int S = GCDM_S, V = GCDM_V, Se = h->Se, Ve = h->Ve, H0 = h->H0, L = h->L;
    Pool pool;
    g_split_absmax = 0.f;
    h->x3_shift = k_shift;
    g_split_w = ldexpf(1.0f, 11 - k_shift);
    // ---- edge embedding (1,1) -> (Se,Ve), bottleneck 1: H = max(1, Ve) = Ve ------------------------
    size_t o_ws, o_bs, o_wd, o_wdf, o_kap, o_wg, o_bg, o_wd1, o_wdf1, o_kap1, o_exwH, o_exwL, o_exgH, o_exgL;
    {
        const std::string p = "gcp_embedding.edge_embedding.";
        WView ws, bs, wd, wdf, wu, wg, bg;
        const int ne = h->sc ? 2 : 1;     // edge GCP2 is (1,1) -> (Se,Ve), with self-conditioning (2,2) -> (Se,Ve)
        if (!get_w(h, p + "scalar_out.weight", Se, ne + Ve + 9, ws) || !get_w(h, p + "scalar_out.bias", 1, Se, bs) ||
            !get_w(h, p + "vector_down.weight", Ve, ne, wd) || !get_w(h, p + "vector_down_frames.weight", 3, ne, wdf) ||
            !get_w(h, p + "vector_up.weight", Ve, Ve, wu) || !get_w(h, p + "vector_out_scale.weight", Ve, Se, wg) ||
            !get_w(h, p + "vector_out_scale.bias", 1, Ve, bg))
            return -1;
        std::vector<float> kap(Ve, 0.f), kap1(Ve, 0.f), wd0(Ve), wd1(Ve, 0.f), wdf0(3), wdf1(3, 0.f);
        for (int c = 0; c < Ve; ++c) {
            float s = 0.f, s1 = 0.f;
            for (int k = 0; k < Ve; ++k) { s += wu.at(c, k) * wd.at(k, 0); if (h->sc) s1 += wu.at(c, k) * wd.at(k, 1); }
            kap[c] = s; kap1[c] = s1;
            wd0[c] = wd.at(c, 0); if (h->sc) wd1[c] = wd.at(c, 1);
        }
        for (int k = 0; k < 3; ++k) { wdf0[k] = wdf.at(k, 0); if (h->sc) wdf1[k] = wdf.at(k, 1); }
        o_ws = pool.add(*ws.v); o_bs = pool.add(*bs.v); o_wd = pool.add(wd0); o_wdf = pool.add(wdf0);
        o_kap = pool.add(kap); o_wg = pool.add(*wg.v); o_bg = pool.add(*bg.v);
        o_wd1 = pool.add(wd1); o_wdf1 = pool.add(wdf1); o_kap1 = pool.add(kap1);
        // split-precision A operands (k_edge_embed_x3, gcdm_embed_x3.hip.h): scalar_out with the K-slot order of that kernel
        // (k = 16 kb + 8 half + s), the bias as the row of the constant-1 slot; vector_out_scale in the register order of the accumulators
        {
            const int NH = Ve / 2, MTe = (Se + 31) / 32;
            Dense Wp(32 * MTe, 32);
            for (int ch = 0; ch < Se; ++ch) {
                for (int hf = 0; hf < 2; ++hf)
                    for (int sl = 0; sl < NH; ++sl) Wp.at(ch, 8 * hf + sl) = ws.at(ch, ne + hf * NH + sl);
                Wp.at(ch, 16) = ws.at(ch, 0);
                if (h->sc) Wp.at(ch, 24) = ws.at(ch, 1);
                for (int sl = 1; sl <= 5; ++sl) Wp.at(ch, 16 + sl) = ws.at(ch, ne + Ve + sl - 1);
                for (int sl = 1; sl <= 4; ++sl) Wp.at(ch, 24 + sl) = ws.at(ch, ne + Ve + 4 + sl);
                Wp.at(ch, 16 + 7) = bs.at(0, ch);
            }
            std::vector<float> xh, xl;
            pack_x3(Wp, xh, xl);
            o_exwH = pool.add(xh); o_exwL = pool.add(xl);
            const int GB = Se / 16;
            std::vector<uint16_t> H((size_t)GB * 64 * 8), Lo(H.size());
            for (int b = 0; b < GB; ++b)
                for (int lane = 0; lane < 64; ++lane)
                    for (int sl = 0; sl < 8; ++sl) {
                        const int c = lane & 31, ch = 32 * (b >> 1) + 16 * (b & 1) + 8 * (sl >> 2) + 4 * (lane >> 5) + (sl & 3);
                        split_f16(c < Ve ? wg.at(c, ch) : 0.f, H[((size_t)b * 64 + lane) * 8 + sl], Lo[((size_t)b * 64 + lane) * 8 + sl]);
                    }
            o_exgH = pool.add(f16_words(H)); o_exgL = pool.add(f16_words(Lo));
        }
    }
    // ---- node embedding (Fin,2) -> (S,V), bottleneck 1 ------------------------------------------------
    GcpOff emb, proj;
    if (!build_gcp(h, pool, "gcp_embedding.node_embedding.", h->Fin, h->sc ? 4 : 2, S, V, 1, false, emb)) return -1;
    if (!build_gcp(h, pool, "scalar_node_projection_gcp.", S, V, h->FinP, 0, 1, false, proj)) return -1;
    std::vector<LayerOff> lo(L);
    for (int l = 0; l < L; ++l) {
        const std::string lp = "interaction_layers." + std::to_string(l) + ".";
        LayerOff& o = lo[l];
        // msg0: scalar_out split into node-level halves (PQ) and the per-edge part [e' | n | q]
        {
            const std::string p = lp + "interaction.message_fusion.0.";
            const int Vin0 = 2 * V + Ve, Kin = 2 * S + Se + H0 + 9;
            WView ws, bs, wd, wdf, wg, bg, wu;
            if (!get_w(h, p + "scalar_out.weight", S, Kin, ws) || !get_w(h, p + "scalar_out.bias", 1, S, bs) ||
                !get_w(h, p + "vector_down.weight", H0, Vin0, wd) || !get_w(h, p + "vector_down_frames.weight", 3, Vin0, wdf) ||
                !get_w(h, p + "vector_out_scale.weight", V, S, wg) || !get_w(h, p + "vector_out_scale.bias", 1, V, bg) ||
                !get_w(h, p + "vector_up.weight", V, H0, wu))
                return -1;
            const int SEG = Se / 4, H0G = (H0 + 3) / 4;
            const int Kp = round_up(4 * (SEG + H0G + 3), 8);
            Dense W0(S, Kp);
            for (int m = 0; m < S; ++m) {
                for (int k = 0; k < Se; ++k) W0.at(m, k) = ws.at(m, S + k);
                for (int k = 0; k < H0; ++k) W0.at(m, 4 * SEG + k) = ws.at(m, 2 * S + Se + k);
                for (int k = 0; k < 9; ++k) W0.at(m, 4 * (SEG + H0G) + k) = ws.at(m, 2 * S + Se + H0 + k);
            }
            o.w0 = pool.add(pack_mfma(W0));
            o.G0 = Kp / 8;
            Dense PQ(2 * S, S);
            for (int m = 0; m < S; ++m)
                for (int k = 0; k < S; ++k) {
                    PQ.at(m, k) = ws.at(m, k);
                    PQ.at(S + m, k) = ws.at(m, S + Se + k);
                }
            o.wpq = pool.add(pack_mfma(PQ));
            o.bpq = pool.add(padded(bs, 2 * S));
            {   // split-precision path: the node kernel writes c * (P | Q) (scaled units of the edge kernel)
                Dense PQc(2 * S, S);
                for (size_t i = 0; i < PQc.a.size(); ++i) PQc.a[i] = PQ.a[i] * X3_C;
                std::vector<float> xh, xl;
                pack_x3(PQc, xh, xl);
                o.wpqH = pool.add(xh); o.wpqL = pool.add(xl);
                std::vector<float> bc = padded(bs, 2 * S);
                for (auto& v : bc) v *= X3_C;
                o.bpqx = pool.add(bc);
            }
            std::vector<float> dI((size_t)(H0 + 3) * V), dJ((size_t)(H0 + 3) * V), dE((size_t)(H0 + 3) * Ve);
            for (int r = 0; r < H0 + 3; ++r) {
                const WView& src = r < H0 ? wd : wdf;
                const int rr = r < H0 ? r : r - H0;
                for (int c = 0; c < V; ++c) dI[(size_t)r * V + c] = src.at(rr, c);
                for (int c = 0; c < Ve; ++c) dE[(size_t)r * Ve + c] = src.at(rr, V + c);
                for (int c = 0; c < V; ++c) dJ[(size_t)r * V + c] = src.at(rr, V + Ve + c);
            }
            o.wddI = pool.add(dI); o.wddJ = pool.add(dJ); o.wddE = pool.add(dE);
            {   // the edge block as ONE split-precision A operand (rows = hidden / frame vectors, padded 32 x 16): beta on the matrix pipe
                Dense We(32, 16);
                for (int r = 0; r < H0 + 3; ++r)
                    for (int c = 0; c < Ve; ++c) We.at(r, c) = dE[(size_t)r * Ve + c];
                std::vector<float> a, b;
                pack_x3(We, a, b);
                o.wbeH = pool.add(a); o.wbeL = pool.add(b);
            }
            {   // [wddI; wddJ] (2 x (H0 + 3) rows x 32) for the node kernel's matrix-pipe evaluation of VDI / VDJ
                std::vector<float> dIJ(dI);
                dIJ.insert(dIJ.end(), dJ.begin(), dJ.end());
                std::vector<float> a, b;
                pack_vecmat(dIJ, 2 * (H0 + 3), V, a, b);
                o.vdH = pool.add(a); o.vdL = pool.add(b);
            }
            Dense Wg(32, S);
            for (int m = 0; m < V; ++m) for (int k = 0; k < S; ++k) Wg.at(m, k) = wg.at(m, k);
            o.wg0 = pool.add(pack_mfma(Wg));
            o.bg0 = pool.add(padded(bg, 32));
            o.wup0 = pool.add(*wu.v);
            // split-precision images: K' = [e'(Se) | n (H0) | q (9)] compact (round 5), padded to a multiple of 16 (x3_msg0_kb / x3_msg0_qpos)
            const int Kx = 16 * x3_msg0_kb(Se, H0), qpos = x3_msg0_qpos(Se, H0);      // (gcdm_edge_x3.hip.h: the kernel's own layout functions)
            Dense W0x(S, Kx);
            for (int m = 0; m < S; ++m) {
                for (int k = 0; k < Se; ++k) W0x.at(m, k) = ws.at(m, S + k);
                for (int k = 0; k < H0; ++k) W0x.at(m, Se + k) = ws.at(m, 2 * S + Se + k);
                for (int k = 0; k < 9; ++k) W0x.at(m, qpos + k) = ws.at(m, 2 * S + Se + H0 + k);
            }
            for (auto& v : W0x.a) v *= X3_C;                    // true-unit inputs -> scaled pre-activation
            std::vector<float> xh, xl;
            pack_x3(W0x, xh, xl);
            o.w0H = pool.add(xh); o.w0L = pool.add(xl); o.KB0 = Kx / 16;
            {
                Dense Wgc(32, S);
                for (size_t i = 0; i < Wgc.a.size(); ++i) Wgc.a[i] = Wg.a[i] / X3_C;   // acts on the scaled SiLU output
                pack_gate_x3(Wgc, xh, xl);
            }
            o.wg0H = pool.add(xh); o.wg0L = pool.add(xl);
            pack_vec_fin0(wu, H0, xh, xl);
            o.vf0H = pool.add(xh); o.vf0L = pool.add(xl);
        }
        for (int k = 1; k <= 3; ++k) {
            const std::string p = lp + "interaction.message_fusion." + std::to_string(k) + ".";
            if (!build_gcp(h, pool, p, S, V, S, V, 4, false, o.mk[k - 1])) return -1;
            // split-precision images: K' = [m.s (256) | n (8) | q (9 -> 16) | pad] = 288
            WView ws, wg, wb;
            if (!get_w(h, p + "scalar_out.weight", S, S + 8 + 9, ws) || !get_w(h, p + "vector_out_scale.weight", V, S, wg) ||
                !get_w(h, p + "scalar_out.bias", 1, S, wb))
                return -1;
            // extended-K rows as the vector stage writes them (gcdm_edge_x3.hip.h): group 32 + g = [n(3g) n(3g+1) n(3g+2) | q(3g) q(3g+1) q(3g+2) | 0 | *],
            // the last slot of group 35 holds a constant 1 (bias as a weight column)
            Dense Wx(S, 288);
            for (int m = 0; m < S; ++m) {
                for (int kk = 0; kk < S; ++kk) Wx.at(m, kk) = ws.at(m, kk);
                for (int g = 0; g < 3; ++g)
                    for (int t = 0; t < 3; ++t) {          // true-unit inputs: weights carry c (the message scalars are already scaled)
                        if (3 * g + t < 8) Wx.at(m, S + 8 * g + t) = X3_C * ws.at(m, S + 3 * g + t);          // norms of the hidden vectors
                        Wx.at(m, S + 8 * g + 3 + t) = X3_C * ws.at(m, S + 8 + 3 * g + t);                     // frame scalars q[3g + t]
                    }
                Wx.at(m, 287) = X3_C * wb.at(0, m);
            }
            std::vector<float> xh, xl;
            pack_x3(Wx, xh, xl);
            o.wH[k - 1] = pool.add(xh); o.wL[k - 1] = pool.add(xl); o.KB = 18;
            Dense Wgd(32, S);
            for (int m = 0; m < V; ++m) for (int kk = 0; kk < S; ++kk) Wgd.at(m, kk) = wg.at(m, kk) / X3_C;
            pack_gate_x3(Wgd, xh, xl);
            o.wgH[k - 1] = pool.add(xh); o.wgL[k - 1] = pool.add(xl);
            {   // vector path on the matrix pipe: vector_down / vector_down_frames and vector_up as 16x16x32 A operands
                WView wd, wdf, wu;
                if (!get_w(h, p + "vector_down.weight", 8, V, wd) || !get_w(h, p + "vector_down_frames.weight", 3, V, wdf) ||
                    !get_w(h, p + "vector_up.weight", V, 8, wu))
                    return -1;
                std::vector<float> a, b;
                pack_vec_pre(wd, wdf, a, b);
                o.vpH[k - 1] = pool.add(a); o.vpL[k - 1] = pool.add(b);
                pack_vec_fin(wu, a, b);
                o.vf1[k - 1] = pool.add(a); o.vf2[k - 1] = pool.add(b);
            }
        }
        {
            WView wa, ba;
            if (!get_w(h, lp + "interaction.scalar_message_attention.0.weight", 1, S, wa) ||
                !get_w(h, lp + "interaction.scalar_message_attention.0.bias", 1, 1, ba))
                return -1;
            o.wa = pool.add(*wa.v);
            o.ba = ba.at(0, 0);
            std::vector<float> wac(*wa.v);
            for (auto& v : wac) v /= X3_C;
            o.wax = pool.add(wac);
        }
        if (!build_gcp(h, pool, lp + "feedforward_network.0.", 2 * S, 2 * V, S, V, 4, true, o.ff)) return -1;
        if (!build_gcp(h, pool, lp + "node_position_update_gcp.", S, V, S, 1, 4, false, o.pos)) return -1;
    }
    pool.host.resize(pool.host.size() + (size_t)X3_TAIL_BLOCKS * 64 * 4 * 2, 0.f);   // un-clamped weight prefetch of the last packed array
    if (h->wpool) (void)hipFree(h->wpool);
    h->wpool = nullptr;
    HIP_OK(h, hipMalloc(&h->wpool, pool.host.size() * sizeof(float)));
    HIP_OK(h, hipMemcpy(h->wpool, pool.host.data(), pool.host.size() * sizeof(float), hipMemcpyHostToDevice));
    h->wpool_bytes = pool.host.size() * sizeof(float);
    if (h->wpool_bytes >= ((size_t)1 << 32)) return fail(h, "gcdm_finalize_weights: weight pool exceeds 4 GB");
    const float* base = h->wpool;
    h->ee_ws = base + o_ws; h->ee_bs = base + o_bs; h->ee_wd = base + o_wd; h->ee_wdf = base + o_wdf;
    h->ee_kappa = base + o_kap; h->ee_wg = base + o_wg; h->ee_bg = base + o_bg;
    h->ee_wd1 = base + o_wd1; h->ee_wdf1 = base + o_wdf1; h->ee_kappa1 = base + o_kap1;
    h->ee_xwH = (const h8*)(base + o_exwH); h->ee_xwL = (const h8*)(base + o_exwL); h->ee_xgH = (const h8*)(base + o_exgH); h->ee_xgL = (const h8*)(base + o_exgL);
    h->emb = resolve(emb, base);
    h->proj = resolve(proj, base);
    h->embx = resolve_x3(emb, base);
    h->projx = resolve_x3(proj, base);
    h->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        const LayerOff& o = lo[l];
        LayerDev& d = h->layers[l];
        d.w0 = (const v4f*)(base + o.w0); d.G0 = o.G0; d.wddE = base + o.wddE;
        d.wg0 = (const v4f*)(base + o.wg0); d.bg0 = base + o.bg0; d.wup0 = base + o.wup0;
        for (int k = 0; k < 3; ++k) d.mk[k] = resolve(o.mk[k], base);
        d.wa = base + o.wa; d.ba = o.ba;
        d.ff = resolve(o.ff, base); d.pos = resolve(o.pos, base);
        d.wpq = (const v4f*)(base + o.wpq); d.bpq = base + o.bpq; d.wddI = base + o.wddI; d.wddJ = base + o.wddJ;
        d.w0H = (const h8*)(base + o.w0H); d.w0L = (const h8*)(base + o.w0L); d.KB0 = o.KB0; d.KB = o.KB;
        d.ffx = resolve_x3(o.ff, base); d.posx = resolve_x3(o.pos, base);
        d.wpqH = (const h8*)(base + o.wpqH); d.wpqL = (const h8*)(base + o.wpqL);
        d.wg0H = (const h8*)(base + o.wg0H); d.wg0L = (const h8*)(base + o.wg0L);
        d.wbeH = (const h8*)(base + o.wbeH); d.wbeL = (const h8*)(base + o.wbeL);
        d.vf0H = (const h8*)(base + o.vf0H); d.vf0L = (const h8*)(base + o.vf0L);
        d.vdH = (const h8*)(base + o.vdH); d.vdL = (const h8*)(base + o.vdL);
        d.bpqx = base + o.bpqx; d.wax = base + o.wax;
        for (int k = 0; k < 3; ++k) {
            d.wH[k] = (const h8*)(base + o.wH[k]); d.wL[k] = (const h8*)(base + o.wL[k]);
            d.wgH[k] = (const h8*)(base + o.wgH[k]); d.wgL[k] = (const h8*)(base + o.wgL[k]);
            d.vpH[k] = (const h8*)(base + o.vpH[k]); d.vpL[k] = (const h8*)(base + o.vpL[k]);
            d.vf1[k] = (const h8*)(base + o.vf1[k]); d.vf2[k] = (const h8*)(base + o.vf2[k]);
        }
    }
    if (!h->attr_set) {
        if (set_lds_attr(h, k_edge_msg<64, 16, 64>, EdgeGeo<64>::LDS_BYTES) || set_lds_attr(h, k_edge_msg<16, 8, 64>, EdgeGeo<64>::LDS_BYTES) ||
            set_lds_attr(h, k_edge_msg<64, 16, 32>, EdgeGeo<32>::LDS_BYTES) || set_lds_attr(h, k_edge_msg<16, 8, 32>, EdgeGeo<32>::LDS_BYTES) ||
            set_lds_attr(h, k_edge_msg_x3<64, 16, 64>, EdgeGeo<64>::LDS_BYTES_X3) || set_lds_attr(h, k_edge_msg_x3<16, 8, 64>, EdgeGeo<64>::LDS_BYTES_X3) ||
            set_lds_attr(h, k_edge_msg_x3<64, 16, 32>, EdgeGeo<32>::LDS_BYTES_X3) || set_lds_attr(h, k_edge_msg_x3<16, 8, 32>, EdgeGeo<32>::LDS_BYTES_X3) ||
            set_lds_attr(h, k_edge_msg_x3<64, 16, 64, NodeTailRole<32>>, NodeTailRole<32>::LDS_BYTES) || set_lds_attr(h, k_edge_msg_x3<16, 8, 64, NodeTailRole<32>>, NodeTailRole<32>::LDS_BYTES) ||
            set_lds_attr(h, k_node_x3<true>, NK_LDS_BYTES) || set_lds_attr(h, k_node_x3<false>, NK_LDS_BYTES) || set_lds_attr(h, k_node_x3w, NW_LDS_BYTES) ||
            set_lds_attr(h, k_node<true>, NK_LDS_BYTES) || set_lds_attr(h, k_node<false>, NK_LDS_BYTES) ||
            set_lds_attr(h, k_node_x3<true, 4>, NK_LDS_BYTES) || set_lds_attr(h, k_node<true, 4>, NK_LDS_BYTES))
            return -1;
        h->attr_set = true;
    }
    // the packed images hold 2^(11-k) W in f16: if even k = X3_MAX_SHIFT cannot hold the largest packed weight (or one is NaN) -> fp32 MFMA only
    h->x3_weights_ok = g_split_absmax * g_split_w < 65504.0f;
    g_split_w = 2048.0f;
    return 0;
}

int gcdm_finalize_weights(gcdm_handle* h) {
    if (!h) return -1;
    DeviceGuard guard(h->cfg.device);
    // Exponent split of the f16 images: the smallest k whose packed images 2^(11-k) W stay inside f16.  What counts is the largest value that is
    // actually PACKED, which only the packing itself knows (the host folds constants into some matrices, most biases never enter an image): pack once
    // with k = 0 -- every released / synthetic model so far stops there -- and, if that overflowed, once more with the k the measured maximum
    // asks for.  (Round 3 took the maximum over ALL parameters x 1.5: a large fp32-only bias then cost activation range for nothing.)
    int st = finalize_pass(h, 0);
    if (st != 0) return st;
    if (!h->x3_weights_ok) {
        const float packed_max = g_split_absmax;        // (thread-local, set by the pass above)
        int k = 0;
        while (k < X3_MAX_SHIFT && !(packed_max * ldexpf(1.0f, 11 - k) < 65504.0f)) ++k;
        st = finalize_pass(h, k);
        if (st != 0) return st;
    }
    h->finalized = true;
    h->host_w.clear();
    drop_step_graph(h);
    return 0;
}

int gcdm_plan_batch(gcdm_handle* h, int32_t B, const int32_t* nn) { return gcdm_plan_batch_masked(h, B, nn, nullptr); }

int gcdm_plan_batch_masked(gcdm_handle* h, int32_t B, const int32_t* nn, const uint8_t* node_mask) {
    if (!h || B <= 0 || !nn) return fail(h, "gcdm_plan_batch: bad argument");
    DeviceGuard guard(h->cfg.device);
    drop_step_graph(h);
    std::vector<int> noff(B + 1, 0);
    int64_t E = 0;
    int max_n = 0;
    bool masked = false;
    for (int b = 0; b < B; ++b) {
        if (nn[b] <= 0) return fail(h, "gcdm_plan_batch: every molecule needs >= 1 atom");
        noff[b + 1] = noff[b] + nn[b];
        int64_t m = nn[b];
        if (node_mask) {            // edges only between unmasked atoms of a molecule (get_fully_connected_edge_index, gcpnet.py:1062-1065)
            m = 0;
            for (int i = 0; i < nn[b]; ++i) m += node_mask[noff[b] + i] ? 1 : 0;
            if (m == 0) return fail(h, "gcdm_plan_batch_masked: every molecule needs >= 1 unmasked atom (the reference's centroid is 0 / 0 otherwise)");
            masked |= m != nn[b];
        }
        E += m * m;
        max_n = std::max(max_n, (int)nn[b]);
    }
    if (!masked) node_mask = nullptr;
    if (E >= (int64_t)1 << 31) return fail(h, "gcdm_plan_batch: too many edges");
    // k_sample / k_prep stage one molecule in LDS (max_n * (3 + F) floats, 64 KB without an opt-in attribute)
    if (max_n > 4096 || (size_t)max_n * h->D * sizeof(float) > 65536) return fail(h, "gcdm_plan_batch: molecule too large (max_n * (3 + F) floats must fit 64 KB of LDS)");
    free_plan(h);                 // only now: a request rejected by the ARGUMENT checks above leaves the previous plan in place; one that fails
                                  // below (workspace over 4 GB, hipMalloc) leaves NO plan (callers must not assume the old one survives:
                                  // GCPNetDynamics.plan forgets its plan key before the call)
    const int N = noff[B];
    std::vector<int> erow(E), ecol(E), ncnt(N), rowstart(N);
    int64_t p = 0;
    for (int b = 0; b < B; ++b) {
        const int o = noff[b], n = nn[b];
        int m = n;
        if (node_mask) { m = 0; for (int i = 0; i < n; ++i) m += node_mask[o + i] ? 1 : 0; }
        for (int i = 0; i < n; ++i) {
            const bool on = !node_mask || node_mask[o + i];
            ncnt[o + i] = on ? m : 0;           // a masked node has no edges: its aggregated row is zero (AggRow)
            rowstart[o + i] = (int)p;
            if (!on) continue;
            for (int j = 0; j < n; ++j)
                if (!node_mask || node_mask[o + j]) { erow[p] = o + i; ecol[p] = o + j; ++p; }
        }
    }
    if (node_mask) {
        std::vector<float> mf(N);
        for (int i = 0; i < N; ++i) mf[i] = node_mask[i] ? 1.f : 0.f;
        HIP_OK(h, hipMalloc(&h->d_mask, N * sizeof(float)));
        HIP_OK(h, hipMemcpy(h->d_mask, mf.data(), N * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_OK(h, hipMalloc(&h->d_noff, (B + 1) * sizeof(int)));
    HIP_OK(h, hipMalloc(&h->d_erow, E * sizeof(int)));
    HIP_OK(h, hipMalloc(&h->d_ecol, E * sizeof(int)));
    HIP_OK(h, hipMalloc(&h->d_ncnt, N * sizeof(int)));
    HIP_OK(h, hipMalloc(&h->d_rowstart, N * sizeof(int)));
    HIP_OK(h, hipMemcpy(h->d_rowstart, rowstart.data(), N * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(h->d_noff, noff.data(), (B + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(h->d_erow, erow.data(), E * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(h->d_ecol, ecol.data(), E * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(h, hipMemcpy(h->d_ncnt, ncnt.data(), N * sizeof(int), hipMemcpyHostToDevice));
    // workspace (all sub-buffers 16-byte aligned)
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 3) & ~size_t(3); return o; };
    const size_t n = (size_t)N, e = (size_t)E;
    const size_t oX0 = take(3 * n), oXC = take(3 * n), oFB = take(9 * n), oC0 = take(12 * n), oHIN = take(4 * h->FinG * n), oH4 = take(GCDM_S * n),
                 oCHI = take(96 * n), oPQ = take(512 * n), oVDI = take((size_t)(h->H0 + 3) * 3 * n), oVDJ = take((size_t)(h->H0 + 3) * 3 * n),
                 oAGG = take(GCDM_AGGW * n), oPART = take(((e + 31) / 32) * 2 * GCDM_AGGW), oVEL = take(3 * n), oEPS = take((size_t)h->D * n), oEP = take((size_t)h->Se * e),
                 oAL = take((size_t)h->Ve * e), oU = take(3 * e), oFR = take(9 * e), oPROF = take(((e + 31) / 32) * 192),
                 oX0SC = take(h->sc ? 3 * n : 0), oBL = take(h->sc ? (size_t)h->Ve * e : 0), oUSC = take(h->sc ? 3 * e : 0),
                 oZK = take((size_t)h->D * n), oZU = take((size_t)h->D * n), oZROW = take(GCDM_AGGW),
                 oPQb = take(512 * n), oVDIb = take((size_t)(h->H0 + 3) * 3 * n), oVDJb = take((size_t)(h->H0 + 3) * 3 * n);
    h->ws_floats = off;
    if (off * sizeof(float) >= ((size_t)1 << 32)) return fail(h, "gcdm_plan_batch: workspace exceeds 4 GB (buffer-addressed); split the batch");
    HIP_OK(h, hipMalloc(&h->ws, off * sizeof(float)));
    HIP_OK(h, hipMemset(h->ws, 0, off * sizeof(float)));
    float* w = h->ws;
    h->X0 = w + oX0; h->XC = w + oXC; h->FBAR = w + oFB; h->CHI0 = w + oC0; h->HIN4 = w + oHIN; h->H4 = w + oH4; h->CHI = w + oCHI;
    h->PQ4 = w + oPQ; h->VDI = w + oVDI; h->VDJ = w + oVDJ; h->AGG = w + oAGG; h->PART = w + oPART; h->VEL = w + oVEL; h->EPS = w + oEPS; h->EP4 = w + oEP;
    h->AL = w + oAL; h->U = w + oU; h->FR = w + oFR; h->PROF = w + oPROF;
    h->ZK = w + oZK; h->ZU = w + oZU; h->ZROW = w + oZROW;     // ZROW is never written: the workspace starts zeroed
    h->PQ4b = w + oPQb; h->VDIb = w + oVDIb; h->VDJb = w + oVDJb;
    h->X0SC = h->sc ? w + oX0SC : nullptr; h->BL = h->sc ? w + oBL : nullptr; h->USC = h->sc ? w + oUSC : nullptr;
    h->B = B; h->N = N; h->E = E; h->max_n = max_n;
    // ---- node-tile queues of the fused layer launch (gcdm_layer_x3.hip.h): 64-edge tiles, XCD x owns the x-th contiguous eighth of the tile list (the partition of
    // k_edge_msg_x3's persistent loop); node tile t (T nodes) waits for the edge tiles first_tile(t) .. last_tile(t) of its rows and is owned by the XCD whose range
    // contains first_tile(t).  Built for T = 32 and T = 64; a plan qualifies when nothing is masked, the launch is persistent and no node tile spans three XCDs.
    h->tail_tiles32 = 0;
    const int wgs_ = h->cus / 8 * 8;
    if (!node_mask && E > (int64_t)64 * wgs_ && wgs_ >= 8) {
        const int G = (int)((E + 63) / 64), base = G >> 3, rem = G & 7;
        int xs[9];
        for (int x = 0; x <= 8; ++x) xs[x] = x * base + std::min(x, rem);
        auto xcd_of_tile = [&](int g) { int x = 0; while (x < 7 && g >= xs[x + 1]) ++x; return x; };
        for (int which = 0; which < 1; ++which) {          // (T = 32; the builder is generic in T)
            const int T = which ? 64 : 32, NTt = (N + T - 1) / T;
            std::vector<int> tab(64 + (size_t)NTt, 0);
            std::vector<std::vector<std::pair<int, int>>> owned(8);          // per XCD: (edge tile the node item stands behind, node tile)
            bool ok = true;
            for (int t = 0; t < NTt && ok; ++t) {
                const int nf = t * T, nl = std::min(nf + T, N) - 1;
                const int ft = rowstart[nf] >> 6, lt = (rowstart[nl] + ncnt[nl] - 1) >> 6;
                const int owner = xcd_of_tile(ft), xl = xcd_of_tile(lt);
                if (xl > owner + 1 || lt - ft + 1 > 0xffff) ok = false;
                tab[64 + t] = (lt - ft + 1) | ((xl != owner ? 1 : 0) << 16);
                owned[owner].push_back({lt, t});
            }
            for (int x = 0; x < 8 && ok; ++x) {
                tab[8 * x] = owned[x].empty() ? 0 : owned[x][0].second;
                tab[8 * x + 1] = (int)owned[x].size();
                tab[8 * x + 2] = xs[x + 1] - xs[x] + (int)owned[x].size();
                if (x >= 1) {                                                // rows of XCD x's first tiles that belong to the boundary node tile owned by XCD x - 1
                    const int tb = erow[(size_t)xs[x] * 64] / T;
                    if ((rowstart[tb * T] >> 6) < xs[x]) tab[8 * x + 3] = std::min((tb + 1) * T, N);
                }
            }
            if (!ok) continue;
            HIP_OK(h, hipMalloc(&h->d_tail_tab[which], tab.size() * sizeof(int)));
            HIP_OK(h, hipMemcpy(h->d_tail_tab[which], tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        h->tail_tiles32 = (N + 31) / 32;
        HIP_OK(h, hipMalloc(&h->d_tail_ctr, ((size_t)h->tail_tiles32 + TAIL_CTR_WORDS) * sizeof(int)));
        HIP_OK(h, hipMemset(h->d_tail_ctr, 0, ((size_t)h->tail_tiles32 + TAIL_CTR_WORDS) * sizeof(int)));
    }
    return 0;
}

int gcdm_check_stability(const GcdmBondTables* tables, const float* x, int64_t x_row_stride, const int32_t* atom_types,
                         const int32_t* mol_offsets, int32_t num_molecules, int32_t* out, void* stream) {
    if (!tables || tables->num_types < 1 || tables->num_types > GCDM_STABILITY_MAX_TYPES || num_molecules < 0 || x_row_stride < 3) return -1;
    if (num_molecules == 0) return 0;
    if (!x || !atom_types || !mol_offsets || !out) return -1;
    hipLaunchKernelGGL(k_stability, dim3(num_molecules), dim3(64), 0, (hipStream_t)stream, *tables, x, (long)x_row_stride, atom_types,
                       mol_offsets, out, (const int64_t*)nullptr, (uint8_t*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int gcdm_bond_orders(const GcdmBondTables* tables, const float* x, int64_t x_row_stride, const int32_t* atom_types, const int32_t* mol_offsets,
                     int32_t num_molecules, const int64_t* pair_offsets, uint8_t* orders, void* stream) {
    if (!tables || tables->num_types < 1 || tables->num_types > GCDM_STABILITY_MAX_TYPES || num_molecules < 0 || x_row_stride < 3) return -1;
    if (num_molecules == 0) return 0;
    if (!x || !atom_types || !mol_offsets || !pair_offsets || !orders) return -1;
    hipLaunchKernelGGL(k_stability, dim3(num_molecules), dim3(64), 0, (hipStream_t)stream, *tables, x, (long)x_row_stride, atom_types,
                       mol_offsets, (int32_t*)nullptr, pair_offsets, orders);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int64_t gcdm_num_nodes(const gcdm_handle* h) { return h ? h->N : -1; }
int64_t gcdm_num_edges(const gcdm_handle* h) { return h ? h->E : -1; }

int gcdm_debug_set_layer_limit(gcdm_handle* h, int32_t n) {
    if (!h) return -1;
    drop_step_graph(h);
    h->layer_limit = n;
    return 0;
}

// Tile size of the split-precision layer node kernel.  A 64-node tile streams each weight byte for twice the nodes, but takes ~1.55x the time of a
// 32-node tile (its load / vector / epilogue phases double, only its GEMM phases do not: 54-71 us against 37-42 us, tests/gpu_node_time.py), and
// the launch is a whole number of rounds over the CUs: 64-node tiles win where they save a round (GEOM 256 x 44 on one handle: 2 -> 1; a 512-molecule
// QM9 slice: 2 -> 1), and lose where they do not (QM9 1024 x 19 on one handle: 3 rounds of 32 against 2 of 64).  Same bits either way.
static int node_tile_for(const gcdm_handle* h, int N) {
    if (h->node_tile) return h->node_tile;
    const int cus = h->cus > 0 ? h->cus : 256;
    const int r32 = ((N + 31) / 32 + cus - 1) / cus, r64 = ((N + 63) / 64 + cus - 1) / cus;
    return 1.55f * r64 < (float)r32 ? 64 : 32;
}

int gcdm_forward(gcdm_handle* h, const float* xh, const float* t, const float* context, float* out, uint32_t* flags, void* stream_) {
    return gcdm_forward_sc(h, xh, nullptr, t, context, out, flags, stream_);
}

// The sampler's network evaluations (transition, gcdm_sample_final_sc) feed the time from the step instead of a t [N] tensor and fold the last stage
// (k_finish) into their k_sample launch: one value for the whole batch, from the device step table when the step is being captured
struct StepFeed { const StepRow* rows; int* cursor; float t_value; };
static int forward_impl(gcdm_handle* h, const float* xh, const float* xh_sc, const float* t, const StepFeed* feed, const float* context, float* out,
                        uint32_t* flags, void* stream_);

int gcdm_forward_sc(gcdm_handle* h, const float* xh, const float* xh_sc, const float* t, const float* context, float* out, uint32_t* flags,
                    void* stream_) {
    if (!h) return -1;
    if (!t) return fail(h, "gcdm_forward: null tensor");
    return forward_impl(h, xh, xh_sc, t, nullptr, context, out, flags, stream_);
}

static int forward_impl(gcdm_handle* h, const float* xh, const float* xh_sc, const float* t, const StepFeed* feed, const float* context, float* out,
                        uint32_t* flags, void* stream_) {
    if (!h) return -1;
    if (xh_sc && !h->sc) return fail(h, "gcdm_forward_sc: the handle was created without self_condition");
    if (!h->finalized) return fail(h, "gcdm_forward: weights not finalized");
    if (!h->N) return fail(h, "gcdm_forward: no batch plan");
    if (!xh || !out) return fail(h, "gcdm_forward: null tensor");
    if (h->C && !context) return fail(h, "gcdm_forward: context required");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = (hipStream_t)stream_;
    const int N = h->N, B = h->B;
    const int E = (int)h->E;
    h->fuse_active = 0;
    PrepArgs pa{xh, t, context, h->d_noff, N, h->F, h->C, h->FinG, h->X0, h->XC, h->FBAR, h->CHI0, (v4f*)h->HIN4, h->flat_prev, h->flat_next,
                h->sc, xh_sc, h->X0SC, h->d_mask};
    if (feed) { pa.t = nullptr; pa.t_rows = feed->rows; pa.t_cursor = feed->cursor; pa.t_value = feed->t_value; }
    pa.flags_dev = h->d_flags;            // cleared by the first kernel of the evaluation
    hipLaunchKernelGGL(k_prep, dim3(B), dim3(64), 3 * h->max_n * sizeof(float), st, pa);
    EdgeEmbedArgs ea{h->X0, h->XC, N, h->d_erow, h->d_ecol, E, h->ee_ws, h->ee_bs, h->ee_wd, h->ee_wdf, h->ee_kappa, h->ee_wg, h->ee_bg,
                     (v4f*)h->EP4, h->AL, h->U, h->FR,
                     h->sc, (h->sc ? 2 : 1) + h->Ve + 9, h->X0SC, h->ee_wd1, h->ee_wdf1, h->ee_kappa1, h->BL, h->USC};
    if (h->use_x3()) {
        EdgeEmbedX3Args ex{h->x3c(), ea, h->ee_xwH, h->ee_xwL, h->ee_xgH, h->ee_xgL};
        const int egrid = (E + 127) / 128;           // 4 waves x 32 edges
        if (h->Se == 64) hipLaunchKernelGGL((k_edge_embed_x3<64, 16>), dim3(egrid), dim3(256), 0, st, ex);
        else hipLaunchKernelGGL((k_edge_embed_x3<16, 8>), dim3(egrid), dim3(256), 0, st, ex);
    } else {
        const int egrid = (E + 255) / 256;
        if (h->Se == 64) hipLaunchKernelGGL((k_edge_embed<64, 16>), dim3(egrid), dim3(256), 0, st, ea);
        else hipLaunchKernelGGL((k_edge_embed<16, 8>), dim3(egrid), dim3(256), 0, st, ea);
    }

    const int L = (h->layer_limit >= 0 && h->layer_limit < h->L) ? h->layer_limit : h->L;
    const bool truncated = L < h->L;
    NodeArgs na{};
    na.N = N; na.F = h->F; na.C = h->C; na.FinG = h->FinG; na.Dout = h->D; na.pos_weight = h->cfg.node_positions_weight;
    na.HIN4 = (const v4f*)h->HIN4; na.CHI0 = h->CHI0; na.emb = h->emb;
    na.agg = AggSrc{h->AGG, h->PART, h->d_rowstart, h->d_ncnt, 0, h->ZROW}; na.H4 = (v4f*)h->H4; na.CHI = h->CHI; na.XC = h->XC; na.X0 = h->X0; na.FBAR = h->FBAR;
    na.PQ4 = (v4f*)h->PQ4; na.VDI = h->VDI; na.VDJ = h->VDJ; na.H0 = h->H0;          // (the embedding writes set 0: layer 0 gathers set 0)
    na.proj = h->proj; na.OUT = out; na.VEL = h->VEL; na.flags_dev = h->d_flags; na.mask = h->d_mask;
    auto set_next = [&](int l) {
        if (l < h->L) {
            const LayerDev& d = h->layers[l];
            na.has_next = 1; na.wpq = d.wpq; na.bpq = d.bpq; na.wddI = d.wddI; na.wddJ = d.wddJ;
        } else {
            na.has_next = 0;
        }
    };
    const int ngrid = (N + NT_ - 1) / NT_;
    NodeX3Args nx{};
    nx.x3c = h->x3c();
    bool node_kb_ok = true;
    auto prep_node_x3 = [&](int next_layer, const LayerDev* cur) {          // fills nx from na for the split-precision node kernels
        nx.base = na;
        nx.emb = h->embx; nx.proj = h->projx;
        if (cur) { nx.ff = cur->ffx; nx.pos = cur->posx; }
        if ((cur && (nx.ff.KB != 34 || nx.pos.KB != 18)) || nx.proj.KB != 19) {     // compile-time k-block counts of k_node_x3
            (void)fail(h, "internal: node k-block counts differ from the kernel's compile-time constants");
            node_kb_ok = false;
            return false;
        }
        nx.prof = (h->profile_node && cur && next_layer < h->L) ? h->PROF : nullptr;
        if (next_layer < h->L) { nx.wpqH = h->layers[next_layer].wpqH; nx.wpqL = h->layers[next_layer].wpqL; nx.vdH = h->layers[next_layer].vdH; nx.vdL = h->layers[next_layer].vdL; nx.bpqx = h->layers[next_layer].bpqx; }
        return true;
    };
    auto launch_node = [&](bool embed, int next_layer, const LayerDev* cur) {
        if (h->use_x3()) {
            if (!prep_node_x3(next_layer, cur)) return;
            if (embed && h->sc) hipLaunchKernelGGL((k_node_x3<true, 4>), dim3(ngrid), dim3(NX_THREADS), NK_LDS_BYTES, st, nx);
            else if (embed) hipLaunchKernelGGL(k_node_x3<true>, dim3(ngrid), dim3(NX_THREADS), NK_LDS_BYTES, st, nx);
            else if (node_tile_for(h, N) == 64) hipLaunchKernelGGL(k_node_x3w, dim3((N + NW_T - 1) / NW_T), dim3(512), NW_LDS_BYTES, st, nx);
            else hipLaunchKernelGGL(k_node_x3<false>, dim3(ngrid), dim3(NX_THREADS), NK_LDS_BYTES, st, nx);
        } else {
            if (embed && h->sc) hipLaunchKernelGGL((k_node<true, 4>), dim3(ngrid), dim3(256), NK_LDS_BYTES, st, na);
            else if (embed) hipLaunchKernelGGL(k_node<true>, dim3(ngrid), dim3(256), NK_LDS_BYTES, st, na);
            else hipLaunchKernelGGL(k_node<false>, dim3(ngrid), dim3(256), NK_LDS_BYTES, st, na);
        }
    };
    set_next(0);
    launch_node(true, 0, nullptr);
    if (!node_kb_ok) return -1;
    const int ET = h->tile();
    const int tiles = (E + ET - 1) / ET;
    na.agg.tile_shift = ET == 32 ? 5 : 6;
    for (int l = 0; l < L; ++l) {
        const LayerDev& d = h->layers[l];
        EdgeMsgArgs ma{};
        ma.EP4 = (const v4f*)h->EP4; ma.AL = h->AL; ma.U = h->U; ma.FR = h->FR; ma.EROW = h->d_erow; ma.ECOL = h->d_ecol; ma.NCNT = h->d_ncnt;
        ma.BL = h->BL; ma.USC = h->USC;
        // node-level msg0 halves: layer l gathers set l & 1, its node tiles write set (l + 1) & 1 for layer l + 1
        const bool odd = (l & 1) != 0;
        ma.E = E; ma.N = N; ma.PQ4 = (const v4f*)(odd ? h->PQ4b : h->PQ4); ma.VDI = odd ? h->VDIb : h->VDI; ma.VDJ = odd ? h->VDJb : h->VDJ; ma.AGG = h->AGG; ma.PART = h->PART;
        na.PQ4 = (v4f*)(odd ? h->PQ4 : h->PQ4b); na.VDI = odd ? h->VDI : h->VDIb; na.VDJ = odd ? h->VDJ : h->VDJb;
        ma.w0 = d.w0; ma.G0 = d.G0; ma.wddE = d.wddE; ma.wg0 = d.wg0; ma.bg0 = d.bg0; ma.wup0 = d.wup0;
        for (int k = 0; k < 3; ++k) ma.mk[k] = d.mk[k];
        ma.wa = d.wa; ma.ba = d.ba;
        ma.prof = h->profile_phases ? h->PROF : nullptr;
        if (h->profile) HIP_OK(h, hipEventRecord(h->ev[2 * l], st));
        if (h->use_x3()) {
            EdgeMsgX3Args xa{};
            xa.x3c = h->x3c();
            xa.base = ma;
            xa.w0H = d.w0H; xa.w0L = d.w0L; xa.KB0 = d.KB0; xa.wg0H = d.wg0H; xa.wg0L = d.wg0L; xa.KB = d.KB;
            xa.wbeH = d.wbeH; xa.wbeL = d.wbeL;
            for (int k = 0; k < 3; ++k) { xa.wH[k] = d.wH[k]; xa.wL[k] = d.wL[k]; xa.wgH[k] = d.wgH[k]; xa.wgL[k] = d.wgL[k]; }
            for (int k = 0; k < 3; ++k) { xa.vpH[k] = d.vpH[k]; xa.vpL[k] = d.vpL[k]; xa.vf1[k] = d.vf1[k]; xa.vf2[k] = d.vf2[k]; }
            xa.vf0H = d.vf0H; xa.vf0L = d.vf0L;
            xa.wax = d.wax;
            xa.wpool = h->wpool; xa.wpool_bytes = (uint32_t)h->wpool_bytes;
            xa.wspool = h->ws; xa.wspool_bytes = (uint32_t)(h->ws_floats * sizeof(float));
            xa.flags_dev = h->d_flags;
            if (d.KB != 18 || d.KB0 != x3_msg0_kb(h->Se, h->H0)) return fail(h, "internal: k-block counts differ from the kernel's compile-time constants");
            // persistent workgroups: as many as fit the chip at once (one per CU with 64-edge tiles, two with 32), a multiple of 8 so that every
            // XCD gets the same number; fewer tiles than that -> one tile per workgroup, as before
            // (32-edge tiles: 83 KB of LDS since round 5, i.e. ONE workgroup per CU -- a grid of 2 x CUs would run in two rounds of one per CU)
            constexpr int per_cu32 = EdgeGeo<32>::LDS_BYTES_X3 * 2 <= 160 * 1024 ? 2 : 1;
            int wgs = h->cus * (ET == 64 ? 1 : per_cu32) / 8 * 8;
            bool persistent = true;
            if (h->persistent == 0 || tiles <= wgs || wgs < 8) { wgs = tiles; xa.wg_stride = tiles; persistent = false; } else xa.wg_stride = wgs / 8;
            // one launch per layer: the node tiles as a tail role of the persistent workgroups (gcdm_layer_x3.hip.h) where the plan qualifies
            int ft = 0;
            if (ET == 64 && persistent && h->fuse_node && !h->profile_node && !h->d_mask && h->d_tail_ctr) {
                // nodes per node tile of the tail role: 32.  (64-node tiles -- node_tile_x3w -- were built and measured too: QM9 6.85 against 6.94 ms per step on one box,
                // 7.30 against 7.24 on another, GEOM 3.81 against 3.75; their node code keeps one SGPR spilled, which costs the edge role a VGPR: not instantiated)
                ft = h->d_tail_tab[0] ? 32 : 0;
            }
            h->fuse_active = ft != 0;
            if (ft) {
                na.ff = d.ff; na.pos = d.pos;
                set_next(l + 1);
                if (!prep_node_x3(l + 1, &d)) return -1;
                TailArgs ta{};
                ta.nx = nx; ta.ready = h->d_tail_ctr; ta.qcur = h->d_tail_ctr + h->tail_tiles32; ta.tab = h->d_tail_tab[ft == 64 ? 1 : 0];
                ta.num_wgs = wgs;
                {
                    NodeTailRole<32>::Args la{xa, ta};
                    if (h->Se == 64) hipLaunchKernelGGL((k_edge_msg_x3<64, 16, 64, NodeTailRole<32>>), dim3(wgs), dim3(512), NodeTailRole<32>::LDS_BYTES, st, la);
                    else hipLaunchKernelGGL((k_edge_msg_x3<16, 8, 64, NodeTailRole<32>>), dim3(wgs), dim3(512), NodeTailRole<32>::LDS_BYTES, st, la);
                }
                if (h->profile) {
                    HIP_OK(h, hipEventRecord(h->ev[2 * l + 1], st)); h->ev_used = l + 1;
                    HIP_OK(h, hipEventRecord(h->ev[2 * (size_t)h->L + l], st));
                }
                continue;
            }
            NoTailRole::Args la{xa};
            if (ET == 64) {
                if (h->Se == 64) hipLaunchKernelGGL((k_edge_msg_x3<64, 16, 64>), dim3(wgs), dim3(512), EdgeGeo<64>::LDS_BYTES_X3, st, la);
                else hipLaunchKernelGGL((k_edge_msg_x3<16, 8, 64>), dim3(wgs), dim3(512), EdgeGeo<64>::LDS_BYTES_X3, st, la);
            } else {
                if (h->Se == 64) hipLaunchKernelGGL((k_edge_msg_x3<64, 16, 32>), dim3(wgs), dim3(256), EdgeGeo<32>::LDS_BYTES_X3, st, la);
                else hipLaunchKernelGGL((k_edge_msg_x3<16, 8, 32>), dim3(wgs), dim3(256), EdgeGeo<32>::LDS_BYTES_X3, st, la);
            }
        } else if (ET == 64) {
            if (h->Se == 64) hipLaunchKernelGGL((k_edge_msg<64, 16, 64>), dim3(tiles), dim3(EdgeGeo<64>::THREADS), EdgeGeo<64>::LDS_BYTES, st, ma);
            else hipLaunchKernelGGL((k_edge_msg<16, 8, 64>), dim3(tiles), dim3(EdgeGeo<64>::THREADS), EdgeGeo<64>::LDS_BYTES, st, ma);
        } else {
            if (h->Se == 64) hipLaunchKernelGGL((k_edge_msg<64, 16, 32>), dim3(tiles), dim3(EdgeGeo<32>::THREADS), EdgeGeo<32>::LDS_BYTES, st, ma);
            else hipLaunchKernelGGL((k_edge_msg<16, 8, 32>), dim3(tiles), dim3(EdgeGeo<32>::THREADS), EdgeGeo<32>::LDS_BYTES, st, ma);
        }
        if (h->profile) { HIP_OK(h, hipEventRecord(h->ev[2 * l + 1], st)); h->ev_used = l + 1; }
        na.ff = d.ff; na.pos = d.pos;
        set_next(l + 1);   // next layer's msg0 halves, or the output projection after the last layer
        launch_node(false, l + 1, &d);
        if (h->profile) HIP_OK(h, hipEventRecord(h->ev[2 * (size_t)h->L + l], st));      // end of the layer's node kernel (gcdm_profile_node_kernel_ms)
    }
    if (!truncated && !feed) {              // (the sampler's evaluations: inside their k_sample launch)
        FinishArgs fa{h->VEL, h->d_noff, N, h->D, out, h->d_flags, flags, h->d_mask};
        hipLaunchKernelGGL(k_finish, dim3(B), dim3(64), 0, st, fa);
    }
    HIP_OK(h, hipGetLastError());
    return 0;
}

// ---- sampler ------------------------------------------------------------------------------------
static inline float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }               // torch F.softplus (threshold 20)
static inline float logsigmoidf(float x) { return x < 0.f ? x - log1pf(expf(x)) : -log1pf(expf(-x)); }
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// torch.round(t * T).long() (variational_diffusion.py:252-255): fp32 product, ties to EVEN -- rintf in the default rounding mode, not
// lroundf (ties away from zero): with e.g. 400 sampling steps on the T = 1000 table every odd s gives t * T = k + 0.5 exactly
int32_t gcdm_timestep_index(float t, int32_t num_timesteps) {
    long idx = (long)rintf(t * (float)num_timesteps);
    if (idx < 0) idx = 0;
    if (idx > num_timesteps) idx = num_timesteps;
    return (int32_t)idx;
}

static float gamma_lookup(const gcdm_handle* h, float t) { return h->gamma[gcdm_timestep_index(t, h->cfg.num_timesteps)]; }

static int launch_sample(gcdm_handle* h, StepArgs& sa, hipStream_t st) {
    if (h->d_mask) return fail(h, "the sampler entry points need an all-True node mask (plan with gcdm_plan_batch); masked plans serve gcdm_forward only");
    DeviceGuard guard(h->cfg.device);
    sa.noff = h->d_noff; sa.N = h->N; sa.D = h->D; sa.node_base = h->node_base;
    if (h->fix_noise) {                 // pre-pass: mean of this draw over all nodes (deterministic order)
        if (!h->d_gmean) HIP_OK(h, hipMalloc(&h->d_gmean, 4 * sizeof(float)));
        hipLaunchKernelGGL(k_noise_mean, dim3(1), dim3(1024), 0, st, sa.noise, sa.seed, sa.draw, h->node_base, h->N, h->D, h->d_gmean);
        sa.gmean = h->d_gmean;
    }
    hipLaunchKernelGGL(k_sample, dim3(h->B), dim3(64), (size_t)h->max_n * h->D * sizeof(float), st, sa);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int gcdm_sample_init(gcdm_handle* h, float* z, const float* noise, uint64_t seed, void* stream_) {
    if (!h || !z || !h->N) return fail(h, "gcdm_sample_init: bad argument / no plan");
    StepArgs sa{};
    sa.z = z; sa.noise = noise; sa.seed = seed; sa.draw = 0x7fffffffu; sa.mode = 1; sa.flags_dev = h->d_flags;
    return launch_sample(h, sa, (hipStream_t)stream_);
}

int gcdm_encode_samples(gcdm_handle* h, const float* xh, float* z, uint32_t* flags, void* stream_) {
    if (!h || !xh || !z || !h->N) return fail(h, "gcdm_encode_samples: bad argument / no plan");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = (hipStream_t)stream_;
    HIP_OK(h, hipMemsetAsync(h->d_flags + 1, 0, 2 * sizeof(uint32_t), st));
    EncodeArgs ea{};
    ea.xh = xh; ea.z = z; ea.noff = h->d_noff; ea.D = h->D; ea.num_atom_types = h->cfg.num_atom_types; ea.include_charges = h->cfg.include_charges;
    ea.nv0 = h->cfg.norm_values[0]; ea.nv1 = h->cfg.norm_values[1]; ea.nv2 = h->cfg.norm_values[2];
    ea.nb1 = h->cfg.norm_biases[1]; ea.nb2 = h->cfg.norm_biases[2];
    ea.stat = h->d_flags + 1;
    hipLaunchKernelGGL(k_encode, dim3(h->B), dim3(64), 0, st, ea);
    if (flags) hipLaunchKernelGGL(k_mean_flag, dim3(1), dim3(1), 0, st, h->d_flags + 1, flags);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int gcdm_sample_step(gcdm_handle* h, float* z, const float* context, int32_t s_index, int32_t num_steps, const float* noise, uint64_t seed,
                     uint32_t* flags, void* stream_) {
    return gcdm_sample_step_to(h, z, z, context, s_index, num_steps, noise, seed, flags, stream_);
}

// One ancestral transition z_t -> z_s (sample_p_zs_given_zt, variational_diffusion.py:1204-1278) for arbitrary normalised times s < t:
// network evaluation at t (with the self-conditioning input `sc`, or none), then the fused update into z_out.
// The three coefficients of z_s = z_t / alpha_coef - c_eps * eps + sigma * noise: sigma_and_alpha_t_given_s (:342-367), sigma (:318-325)
static StepRow step_row(const gcdm_handle* h, float s, float t, uint32_t draw) {
    const float gs = gamma_lookup(h, s), gt = gamma_lookup(h, t);
    const float s2ts = -expm1f(softplusf(gs) - softplusf(gt));
    const float alpha_ts = expf(0.5f * (logsigmoidf(-gt) - logsigmoidf(-gs)));
    const float sts = sqrtf(s2ts), sig_s = sqrtf(sigmoidf_(gs)), sig_t = sqrtf(sigmoidf_(gt));
    StepRow r{};
    r.t = t;
    r.alpha_coef = alpha_ts;
    r.c_eps = s2ts / alpha_ts / sig_t;
    r.sigma = sts * sig_s / sig_t;
    r.draw = draw;
    return r;
}

static int transition(gcdm_handle* h, const float* z_in, float* z_out, const float* sc, const float* context, float s, float t,
                      const float* noise, uint64_t seed, uint32_t draw, uint32_t* flags, void* stream_) {
    hipStream_t st = (hipStream_t)stream_;
    if (h->d_mask) return fail(h, "the sampler entry points need an all-True node mask (plan with gcdm_plan_batch); masked plans serve gcdm_forward only");
    // the step being captured into a graph: t and the coefficients come from the device table at run time
    const StepFeed feed{h->capturing ? h->d_rows : nullptr, h->capturing ? h->d_cursor : nullptr, t};
    if (forward_impl(h, z_in, sc, nullptr, &feed, context, h->EPS, flags, stream_)) return -1;
    const StepRow r = step_row(h, s, t, draw);
    StepArgs sa{};
    if (h->capturing) { sa.rows = h->d_rows; sa.cursor = h->d_cursor + 1; sa.cursor_rw = h->d_cursor; }
    sa.VEL = h->VEL;
    sa.z = const_cast<float*>(z_in); sa.z_out = z_out; sa.eps = h->EPS; sa.noise = noise; sa.seed = seed; sa.draw = draw; sa.mode = 0;
    sa.alpha_coef = r.alpha_coef;
    sa.c_eps = r.c_eps;
    sa.sigma = r.sigma;
    sa.user_flags = flags; sa.flags_dev = h->d_flags;
    return launch_sample(h, sa, st);
}

// One step through the handle's instantiated graph (see gcdm_handle::step_graph).  Returns 1 when the step was enqueued, 0 when the caller should
// launch directly (not eligible, or capture failed on this handle), -1 on error.
static int step_via_graph(gcdm_handle* h, float* z, const float* context, int32_t s_index, int32_t num_steps, uint64_t seed, uint32_t* flags,
                          hipStream_t st) {
    if (!h->step_graph || h->step_graph_failed || h->fix_noise || h->profile || h->profile_phases || h->profile_node || h->layer_limit >= 0 || h->sc ||
        h->d_mask || !h->N)
        return 0;
    DeviceGuard guard(h->cfg.device);
    auto give_up = [&](const char* what, hipError_t e) {
        h->step_graph_failed = true;
        h->capturing = false;
        h->err = std::string("step graph disabled on this handle (") + what + ": " + hipGetErrorString(e) + "); launching directly";
        (void)hipGetLastError();
        drop_step_graph(h);
        return 0;
    };
    hipError_t e;
    if (!h->cap_stream && (e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking)) != hipSuccess) return give_up("hipStreamCreate", e);
    if (!h->d_cursor && (e = hipMalloc(&h->d_cursor, 2 * sizeof(int))) != hipSuccess) return give_up("hipMalloc", e);      // two slots (k_cursor_set)
    if (h->rows_steps != num_steps) {              // the table of this step count (s = i / num_steps, t = (i + 1) / num_steps, as gcdm_sample_step_to)
        std::vector<StepRow> rows((size_t)num_steps);
        for (int i = 0; i < num_steps; ++i) rows[i] = step_row(h, (float)i / (float)num_steps, (float)(i + 1) / (float)num_steps, (uint32_t)i);
        if (h->d_rows) { wait_step_graph_idle(h); (void)hipFree(h->d_rows); h->d_rows = nullptr; }     // (a graph still reading the old table: whatever stream it was launched on)
        if ((e = hipMalloc(&h->d_rows, rows.size() * sizeof(StepRow))) != hipSuccess) return give_up("hipMalloc", e);
        if ((e = hipMemcpy(h->d_rows, rows.data(), rows.size() * sizeof(StepRow), hipMemcpyHostToDevice)) != hipSuccess) return give_up("hipMemcpy", e);
        if (h->sg.exec) { wait_step_graph_idle(h); (void)hipGraphExecDestroy(h->sg.exec); (void)hipGraphDestroy(h->sg.graph); h->sg = gcdm_handle::StepGraph{}; }
        h->rows_steps = num_steps;
    }
    if (!h->sg.exec || h->sg.z != z || h->sg.ctx != context || h->sg.flags != flags || h->sg.seed != seed || h->sg.num_steps != num_steps) {
        if (h->sg.exec) { wait_step_graph_idle(h); (void)hipGraphExecDestroy(h->sg.exec); (void)hipGraphDestroy(h->sg.graph); h->sg = gcdm_handle::StepGraph{}; }
        if ((e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal)) != hipSuccess) return give_up("hipStreamBeginCapture", e);
        h->capturing = true;
        const int rc = transition(h, z, z, nullptr, context, 0.f, 1.f / (float)num_steps, nullptr, seed, 0u, flags, (void*)h->cap_stream);
        h->capturing = false;
        hipGraph_t g = nullptr;
        e = hipStreamEndCapture(h->cap_stream, &g);
        if (rc != 0 || e != hipSuccess || !g) {
            if (g) (void)hipGraphDestroy(g);
            if (rc != 0 && e == hipSuccess) {
                // the step itself was refused (an argument / state check of transition(): e.g. a conditional model called without a context) while the
                // capture as such went through: that is the CALLER's error -- report transition()'s own diagnostic (h->err), exactly what the direct path
                // returns for the same call, and leave the graph enabled for the next valid call
                (void)hipGetLastError();
                return -1;
            }
            return give_up("capture of one step", e);
        }
        hipGraphExec_t ex = nullptr;
        if ((e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0)) != hipSuccess) { (void)hipGraphDestroy(g); return give_up("hipGraphInstantiate", e); }
        h->sg.graph = g; h->sg.exec = ex; h->sg.z = z; h->sg.ctx = context; h->sg.flags = flags; h->sg.seed = seed; h->sg.num_steps = num_steps;
        h->cursor_expected = -1;
    }
    // a caller that alternates streams: the new stream waits for the graph launch in flight on the old one (the step reads and writes the handle's workspace
    // anyway), so sg_done -- re-recorded below -- always covers everything that may still read the exec or the row table
    if (h->sg_inflight && h->sg_done && h->sg_last_stream != st) (void)hipStreamWaitEvent(st, h->sg_done, 0);
    h->sg_last_stream = st;
    if (h->cursor_expected != s_index) hipLaunchKernelGGL(k_cursor_set, dim3(1), dim3(1), 0, st, h->d_cursor, (int)s_index);
    if ((e = hipGraphLaunch(h->sg.exec, st)) != hipSuccess) return give_up("hipGraphLaunch", e);
    if (!h->sg_done && hipEventCreateWithFlags(&h->sg_done, hipEventDisableTiming) != hipSuccess) h->sg_done = nullptr;
    if (h->sg_done) h->sg_inflight = hipEventRecord(h->sg_done, st) == hipSuccess;
    if (!h->sg_inflight) (void)hipStreamSynchronize(st);          // (no event: the conservative form)
    h->cursor_expected = s_index - 1;
    h->graph_launches += 1;
    return 1;
}

int gcdm_sample_step_to(gcdm_handle* h, const float* z_in, float* z_out, const float* context, int32_t s_index, int32_t num_steps,
                        const float* noise, uint64_t seed, uint32_t* flags, void* stream_) {
    if (!h || !z_in || !z_out || num_steps <= 0 || s_index < 0 || s_index >= num_steps) return fail(h, "gcdm_sample_step: bad argument");
    if ((int64_t)h->gamma.size() != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_sample_step: gamma table not set");
    // s = s_index / num_steps, t = (s_index + 1) / num_steps (variational_diffusion.py:1335-1341); t [N] = t[batch_index] (:1239)
    if (z_in == z_out && !noise) {
        const int g = step_via_graph(h, z_out, context, s_index, num_steps, seed, flags, (hipStream_t)stream_);
        if (g != 0) return g < 0 ? -1 : 0;
    }
    return transition(h, z_in, z_out, nullptr, context, (float)s_index / (float)num_steps, (float)(s_index + 1) / (float)num_steps, noise, seed,
                      (uint32_t)s_index, flags, stream_);
}

int gcdm_sample_step_sc(gcdm_handle* h, float* z, float* self_cond, int32_t have_self_cond, const float* context, int32_t s_index,
                        int32_t num_steps, const float* noise, const float* noise_self_cond, uint64_t seed, uint32_t* flags, void* stream_) {
    if (!h || !z || !self_cond || num_steps <= 0 || s_index < 0 || s_index >= num_steps) return fail(h, "gcdm_sample_step_sc: bad argument");
    if (!h->sc) return fail(h, "gcdm_sample_step_sc: the handle was created without self_condition");
    if ((int64_t)h->gamma.size() != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_sample_step_sc: gamma table not set");
    const float s = (float)s_index / (float)num_steps, t = (float)(s_index + 1) / (float)num_steps;
    // z_s ~ p(z_s | z_t, previous estimate) (:1343-1352)
    if (transition(h, z, z, have_self_cond ? self_cond : nullptr, context, s, t, noise, seed, (uint32_t)s_index, flags, stream_)) return -1;
    // the next estimate: a jump from s to 0 WITHOUT a self-conditioning input (:1363-1375); own Philox draw index
    return transition(h, z, self_cond, nullptr, context, 0.0f, s, noise_self_cond, seed, 0x20000000u | (uint32_t)s_index, flags, stream_);
}

int gcdm_sample_final(gcdm_handle* h, const float* z0, const float* context, const float* noise, uint64_t seed, float* out, uint32_t* flags,
                      void* stream_) {
    return gcdm_sample_final_sc(h, z0, nullptr, context, noise, seed, out, flags, stream_);
}

int gcdm_sample_final_sc(gcdm_handle* h, const float* z0, const float* self_cond, const float* context, const float* noise, uint64_t seed,
                         float* out, uint32_t* flags, void* stream_) {
    if (!h || !z0 || !out) return fail(h, "gcdm_sample_final: bad argument");
    if ((int64_t)h->gamma.size() != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_sample_final: gamma table not set");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = (hipStream_t)stream_;
    if (h->d_mask) return fail(h, "the sampler entry points need an all-True node mask (plan with gcdm_plan_batch); masked plans serve gcdm_forward only");
    const StepFeed feed{nullptr, nullptr, 0.0f};
    if (forward_impl(h, z0, self_cond, nullptr, &feed, context, h->EPS, flags, stream_)) return -1;
    const float g0 = gamma_lookup(h, 0.0f);
    const float sigma_x = expf(0.5f * g0);                       // SNR(-0.5 * gamma_0)   (:855-859)
    const float sig0 = sqrtf(sigmoidf_(g0)), alp0 = sqrtf(sigmoidf_(-g0));
    StepArgs sa{};
    sa.z = const_cast<float*>(z0); sa.eps = h->EPS; sa.VEL = h->VEL; sa.noise = noise; sa.seed = seed; sa.draw = 0x7ffffffeu; sa.mode = 2;
    sa.alpha_coef = 1.0f / alp0; sa.c_eps = sig0; sa.sigma = sigma_x;
    sa.out = out; sa.num_atom_types = h->cfg.num_atom_types; sa.include_charges = h->cfg.include_charges;
    sa.nv0 = h->cfg.norm_values[0]; sa.nv1 = h->cfg.norm_values[1]; sa.nv2 = h->cfg.norm_values[2];
    sa.nb1 = h->cfg.norm_biases[1]; sa.nb2 = h->cfg.norm_biases[2];
    sa.user_flags = flags; sa.flags_dev = h->d_flags;
    if (launch_sample(h, sa, st)) return -1;
    // CoG drift re-projection is a whole-batch decision in the reference (:1389-1402); only "for examples without intermediate states"
    if (h->cog_fix) hipLaunchKernelGGL(k_cog_fix, dim3(h->B), dim3(64), 0, st, out, h->d_noff, h->D, h->d_flags, flags);
    HIP_OK(h, hipGetLastError());
    return 0;
}

// ---- RePaint inpainting (variational_diffusion.py:1582-1789) ----
int gcdm_inpaint_center(gcdm_handle* h, const float* xh, const uint8_t* fixed, float* xh0, void* stream_) {
    if (!h || !xh || !fixed || !xh0 || !h->N) return fail(h, "gcdm_inpaint_center: bad argument / no plan");
    DeviceGuard guard(h->cfg.device);
    hipLaunchKernelGGL(k_inpaint_center, dim3(h->B), dim3(64), 0, (hipStream_t)stream_, xh, fixed, h->d_noff, h->D, xh0);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int gcdm_inpaint_step(gcdm_handle* h, float* z, const float* xh0, const uint8_t* fixed, float* self_cond, int32_t have_self_cond,
                      const float* context, int32_t s_index, int32_t num_steps, const float* noise_known, const float* noise_unknown,
                      const float* noise_self_cond, uint64_t seed, uint32_t draw_base, uint32_t* flags, void* stream_) {
    if (!h || !z || !xh0 || !fixed || num_steps <= 0 || s_index < 0 || s_index >= num_steps || !h->N) return fail(h, "gcdm_inpaint_step: bad argument / no plan");
    if ((h->sc != 0) != (self_cond != nullptr)) return fail(h, "gcdm_inpaint_step: self_cond must be given exactly when the handle has self_condition");
    if ((int64_t)h->gamma.size() != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_inpaint_step: gamma table not set");
    DeviceGuard guard(h->cfg.device);
    hipStream_t st = (hipStream_t)stream_;
    const float s = (float)s_index / (float)num_steps, t = (float)(s_index + 1) / (float)num_steps;
    // known nodes: q(z_s | x, h) of the given molecule (compute_noised_representation, :910-931)
    const float gs = gamma_lookup(h, s);
    StepArgs sa{};
    sa.z = const_cast<float*>(xh0); sa.z_out = h->ZK; sa.noise = noise_known; sa.seed = seed; sa.draw = draw_base; sa.mode = 3;
    sa.alpha_coef = sqrtf(sigmoidf_(-gs)); sa.sigma = sqrtf(sigmoidf_(gs));
    sa.user_flags = flags; sa.flags_dev = h->d_flags;
    if (launch_sample(h, sa, st)) return -1;
    // everything: one ancestral step of the model (:1652-1662), then the next self-conditioning estimate from it (:1664-1676)
    if (transition(h, z, h->ZU, (self_cond && have_self_cond) ? self_cond : nullptr, context, s, t, noise_unknown, seed, draw_base + 1, flags, stream_)) return -1;
    if (self_cond && transition(h, h->ZU, self_cond, nullptr, context, 0.0f, s, noise_self_cond, seed, draw_base + 2, flags, stream_)) return -1;
    hipLaunchKernelGGL(k_inpaint_combine, dim3(h->B), dim3(64), 0, st, h->ZK, h->ZU, fixed, h->d_noff, h->D, z);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int gcdm_inpaint_jump(gcdm_handle* h, float* z, int32_t s_index, int32_t t_index, int32_t num_steps, const float* noise, uint64_t seed,
                      uint32_t draw, void* stream_) {
    if (!h || !z || num_steps <= 0 || s_index < 0 || t_index <= s_index || t_index > num_steps || !h->N) return fail(h, "gcdm_inpaint_jump: bad argument / no plan");
    if ((int64_t)h->gamma.size() != (int64_t)h->cfg.num_timesteps + 1) return fail(h, "gcdm_inpaint_jump: gamma table not set");
    // q(z_t | z_s) (sample_p_zt_given_zs, :1163-1201; sigma_and_alpha_t_given_s :342-367)
    const float gs = gamma_lookup(h, (float)s_index / (float)num_steps), gt = gamma_lookup(h, (float)t_index / (float)num_steps);
    StepArgs sa{};
    sa.z = z; sa.z_out = z; sa.noise = noise; sa.seed = seed; sa.draw = draw; sa.mode = 4;
    sa.alpha_coef = expf(0.5f * (logsigmoidf(-gt) - logsigmoidf(-gs)));
    sa.sigma = sqrtf(-expm1f(softplusf(gs) - softplusf(gt)));
    sa.flags_dev = h->d_flags;
    return launch_sample(h, sa, (hipStream_t)stream_);
}

int gcdm_unnormalize_z(gcdm_handle* h, const float* z, float* out, void* stream_) {
    if (!h || !z || !out || !h->N) return fail(h, "gcdm_unnormalize_z: bad argument / no plan");
    DeviceGuard guard(h->cfg.device);
    const int total = h->N * h->D;
    hipLaunchKernelGGL(k_unnormalize, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, z, out, h->N, h->D, h->cfg.num_atom_types,
                       h->cfg.norm_values[0], h->cfg.norm_values[1], h->cfg.norm_values[2], h->cfg.norm_biases[1], h->cfg.norm_biases[2]);
    HIP_OK(h, hipGetLastError());
    return 0;
}

int gcdm_set_option(gcdm_handle* h, const char* name, int32_t value) {
    if (!h || !name) return fail(h, "gcdm_set_option: bad argument");
    const std::string k(name);
    // every option below except these two is baked into a captured step (cog_fix acts in gcdm_sample_final only; the sampler sets it right behind its
    // last step); an option set to the value it already has changes nothing either
    if (k == "step_graph") { h->step_graph = value ? 1 : 0; h->step_graph_failed = false; if (!value) drop_step_graph(h); return 0; }
    if (k == "cog_fix") { h->cog_fix = value ? 1 : 0; return 0; }
    // (the getters of the options below return the stored field itself -- edge_tile: the effective tile, which is what a captured step bakes in; an option whose
    //  getter normalises its value must be added to the always-drop list beside mfma_mode)
    if (gcdm_get_option(h, name) != value || k == "mfma_mode" || k == "fuse_node") drop_step_graph(h);
    if (k == "mfma_mode") {                 // 0: fp32 MFMA, 1: split-precision f16 x3 (fp32-equivalent, 5.3x the matrix rate)
        if (value != 0 && value != 1) return fail(h, "gcdm_set_option(mfma_mode): 0 or 1");
        if (value == 1 && !h->x3_weights_ok) return fail(h, "gcdm_set_option(mfma_mode): a weight of this model is >= 2047 in magnitude (or not finite), outside the split-precision images at every exponent split; only mode 0 (fp32 MFMA) is available");
        h->mfma_x3 = value;
        return 0;
    }
    if (k == "fix_noise") { h->fix_noise = value ? 1 : 0; return 0; }
    if (k == "flat_prev") { h->flat_prev = value ? 1 : 0; return 0; }
    if (k == "flat_next") { h->flat_next = value ? 1 : 0; return 0; }
    if (k == "node_base") { if (value < 0) return fail(h, "gcdm_set_option(node_base): >= 0"); h->node_base = (uint32_t)value; return 0; }
    if (k == "persistent") { h->persistent = value ? 1 : 0; return 0; }
    if (k == "fuse_node") { h->fuse_node = value ? 1 : 0; return 0; }
    if (k == "fuse_tile") { if (value != 0 && value != 32) return fail(h, "gcdm_set_option(fuse_tile): 0 or 32"); h->fuse_tile = value; return 0; }
    if (k == "node_tile") {
        if (value != 0 && value != 32 && value != 64) return fail(h, "gcdm_set_option: node_tile must be 0 (automatic), 32 or 64");
        h->node_tile = value;
        return 0;
    }
    if (k == "edge_tile") {
        if (value != 0 && value != 32 && value != 64) return fail(h, "gcdm_set_option(edge_tile): 0 (automatic), 32 or 64");
        h->edge_tile = value;
        return 0;
    }
    return fail(h, "gcdm_set_option: unknown option " + k);
}

int gcdm_get_option(const gcdm_handle* h, const char* name) {
    if (!h || !name) return -1;
    const std::string k(name);
    if (k == "mfma_mode") return h->use_x3() ? 1 : 0;
    if (k == "edge_tile") return h->tile();
    if (k == "cog_fix") return h->cog_fix;
    if (k == "fix_noise") return h->fix_noise;
    if (k == "flat_prev") return h->flat_prev;
    if (k == "flat_next") return h->flat_next;
    if (k == "node_base") return (int)h->node_base;
    if (k == "x3_shift") return h->x3_shift;
    if (k == "persistent") return h->persistent;
    if (k == "fuse_node") return h->fuse_node;
    if (k == "fuse_active") return h->fuse_active;
    if (k == "fuse_tile") return h->fuse_tile;
    if (k == "node_tile") return h->node_tile;
    if (k == "step_graph") return (h->step_graph && !h->step_graph_failed) ? 1 : 0;
    if (k == "graph_launches") return (int)(h->graph_launches & 0x7fffffff);
    return -1;
}

int gcdm_profile_enable(gcdm_handle* h, int32_t enable) {
    if (!h) return -1;
    DeviceGuard guard(h->cfg.device);
    if (enable && h->ev.empty()) {
        h->ev.resize(3 * (size_t)h->L);
        for (auto& e : h->ev) HIP_OK(h, hipEventCreate(&e));
    }
    if (enable == 3 && !GCDM_HAVE_STAMPS)        // (enable == 2 on a build without the phase stamps: only entry 20, the end-of-tile stamp of the split-precision edge kernel, is written)
        return fail(h, "gcdm_profile_enable: in-kernel phase stamps need a library built with -DGCDM_STAMPS (tools/build_variants.sh stamps:-DGCDM_STAMPS)");
    drop_step_graph(h);
    h->profile = enable != 0;
    h->profile_phases = enable == 2;
    h->profile_node = enable == 3;
    h->ev_used = 0;
    return 0;
}

int gcdm_profile_edge_kernel_ms(gcdm_handle* h, double* total_ms, int32_t* launches) {
    if (!h || !total_ms || !launches) return fail(h, "gcdm_profile_edge_kernel_ms: bad argument");
    double tot = 0.0;
    for (int l = 0; l < h->ev_used; ++l) {
        HIP_OK(h, hipEventSynchronize(h->ev[2 * l + 1]));
        float ms = 0.f;
        HIP_OK(h, hipEventElapsedTime(&ms, h->ev[2 * l], h->ev[2 * l + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = h->ev_used;
    return 0;
}

int gcdm_profile_node_kernel_ms(gcdm_handle* h, double* total_ms, int32_t* launches) {
    if (!h || !total_ms || !launches) return fail(h, "gcdm_profile_node_kernel_ms: bad argument");
    double tot = 0.0;
    for (int l = 0; l < h->ev_used; ++l) {
        HIP_OK(h, hipEventSynchronize(h->ev[2 * (size_t)h->L + l]));
        float ms = 0.f;
        HIP_OK(h, hipEventElapsedTime(&ms, h->ev[2 * l + 1], h->ev[2 * (size_t)h->L + l]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = h->ev_used;
    return 0;
}

int64_t gcdm_debug_read(gcdm_handle* h, const char* name, float* host_out, int64_t capacity) {
    if (!h || !name || !h->N) return fail(h, "gcdm_debug_read: bad argument / no plan");
    const int64_t n = h->N, e = h->E;
    const std::string k(name);
    const float* p = nullptr;
    int64_t cnt = 0;
    if (k == "h") { p = h->H4; cnt = GCDM_S * n; }
    else if (k == "chi") { p = h->CHI; cnt = 96 * n; }
    else if (k == "x") { p = h->XC; cnt = 3 * n; }
    else if (k == "x0") { p = h->X0; cnt = 3 * n; }
    else if (k == "agg") {              // assembled on the host: whole rows from AGG, cut rows = their partial sums in tile order (AggSrc)
        cnt = GCDM_AGGW * n;
        if (!host_out) return cnt;
        if (capacity < cnt) return fail(h, "gcdm_debug_read: capacity too small");
        DeviceGuard guard(h->cfg.device);
        const int ET = h->tile(), sh = ET == 32 ? 5 : 6;
        const int64_t tiles = (e + ET - 1) / ET;
        std::vector<float> part((size_t)tiles * 2 * GCDM_AGGW);
        std::vector<int> rs(n), nc(n);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, h->AGG, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(part.data(), h->PART, part.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(rs.data(), h->d_rowstart, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(nc.data(), h->d_ncnt, n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(h, "gcdm_debug_read: copy failed");
        for (int64_t i = 0; i < n; ++i) {
            float* dst = host_out + i * GCDM_AGGW;
            if (nc[i] == 0) {           // masked node (no edges): its aggregate is zero, as the device-side agg_row has it
                for (int c = 0; c < GCDM_AGGW; ++c) dst[c] = 0.f;
                continue;
            }
            const int t0 = rs[i] >> sh, t1 = (rs[i] + nc[i] - 1) >> sh;
            if (t0 == t1) continue;
            const float* src = part.data() + ((size_t)t0 * 2 + ((rs[i] & (ET - 1)) ? 1 : 0)) * GCDM_AGGW;
            for (int c = 0; c < GCDM_AGGW; ++c) dst[c] = src[c];
            for (int t = t0 + 1; t <= t1; ++t)
                for (int c = 0; c < GCDM_AGGW; ++c) dst[c] += part[((size_t)t * 2) * GCDM_AGGW + c];
        }
        return cnt;
    }
    else if (k == "ep") { p = h->EP4; cnt = (int64_t)h->Se * e; }
    else if (k == "alpha") { p = h->AL; cnt = (int64_t)h->Ve * e; }
    else if (k == "u") { p = h->U; cnt = 3 * e; }
    else if (k == "frames") { p = h->FR; cnt = 9 * e; }
    else if (k == "pq") { p = h->PQ4; cnt = 512 * n; }
    else if (k == "vdi") { p = h->VDI; cnt = (int64_t)(h->H0 + 3) * 3 * n; }
    else if (k == "vdj") { p = h->VDJ; cnt = (int64_t)(h->H0 + 3) * 3 * n; }
    else if (k == "hin") { p = h->HIN4; cnt = 4 * (int64_t)h->FinG * n; }
    else if (k == "fbar") { p = h->FBAR; cnt = 9 * n; }
    else if (k == "chi0") { p = h->CHI0; cnt = (h->sc ? 12 : 6) * n; }
    else if (k == "vel") { p = h->VEL; cnt = 3 * n; }
    else if (k == "erow") { p = (const float*)h->d_erow; cnt = e; }        // int32 bit patterns (edge list of the plan: row / col node per flat edge)
    else if (k == "ecol") { p = (const float*)h->d_ecol; cnt = e; }
    else if (k == "phase") { p = h->PROF; cnt = ((e + h->tile() - 1) / h->tile()) * 192; }
    else if (k == "phase_node") { p = h->PROF; cnt = ((n + NT_ - 1) / NT_) * 192; }
    else return fail(h, "gcdm_debug_read: unknown buffer " + k);
    if (!host_out) return cnt;
    if (capacity < cnt) return fail(h, "gcdm_debug_read: capacity too small");
    DeviceGuard guard(h->cfg.device);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(host_out, p, cnt * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(h, "gcdm_debug_read: copy failed");
    return cnt;
}

double gcdm_forward_flops_executed(const gcdm_handle* h) {
    if (!h || !h->N) return 0.0;
    // multiply-adds actually issued per forward (x2; a split-precision product block counts once, padded MFMA rows / columns count), see DESIGN.md section 4
    const double N = h->N, E = (double)h->E, S = GCDM_S, V = GCDM_V, Se = h->Se, Ve = h->Ve, H0 = h->H0;
    const double G0 = h->layers.empty() ? 0 : h->layers[0].G0;
    const bool x3 = h->use_x3();
    // fp32 kernels: extended-K GEMM + gate + VALU vector products; split-precision kernels: 16-deep k-blocks, vector path on 16x16x32 MFMA tiles
    const double msg0 = x3 ? 16.0 * (h->layers.empty() ? 0 : h->layers[0].KB0) * S + 32.0 * S + (H0 + 3) * Ve + 3.0 * V * 32
                           : 8.0 * G0 * S + 32.0 * S + (H0 + 3) * (Ve + 2) * 1.0 + 3.0 * V * H0;
    const double msgk = x3 ? 288.0 * S + 32.0 * S + 16.0 * 32 * 3 + 32.0 * 32 * 3
                           : 280.0 * S + 32.0 * S + 11.0 * 3 * V + 3.0 * V * 8;
    const double edge = msg0 + 3 * msgk + S;
    const double node = 544.0 * S + S * S + 32.0 * S + 19.0 * 3 * 2 * V + 3.0 * V * 16      // ff
                        + 280.0 * S + 32.0 * S + 11.0 * 3 * V + 3.0 * 8                       // pos
                        + 512.0 * S + 2.0 * (H0 + 3) * 3 * V;                                 // next-layer halves
    const double emb_e = Se * (1 + Ve + 9) + Ve * Se, emb_n = 8.0 * h->emb.G * S + 32.0 * S + 35.0 * 3 * 2 + 3.0 * V * 32;
    const double proj = 8.0 * h->proj.G * 32 + 35.0 * 3 * V;
    return 2.0 * (h->L * (E * edge + N * node) + E * emb_e + N * (emb_n + proj));
}

}  // extern "C"
