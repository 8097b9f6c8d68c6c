// gcdm_layer_x3.hip.h -- one interaction layer as ONE launch (round 6): the node tiles of the layer (GCPInteractions.forward behind the aggregation,
// gcpnet.py:834-930) run as a TAIL ROLE of the persistent edge-message workgroups (k_edge_msg_x3<..., NodeTailRole<T>>, gcdm_edge_x3.hip.h).
//
// Why.  As two launches per layer the node kernel was a whole number of rounds of its own (QM9 1024 x 19: 608 tiles of 32 nodes on 256 CUs = 2.375 -> 3 rounds,
// 21 % of it quantisation), every workgroup issued its 210-270 KB load burst at the same instant, the edge kernel's own tail (22.56 -> 23 rounds) idled 44 % of
// the CUs for a tile's time, and there were 18 dependent launch boundaries per step.  Here a workgroup that has walked its edge tiles takes node tiles from
// its XCD's queue until the queue is empty: the two kinds of tiles pack onto the CUs back to back, node tiles start as their inputs complete instead of all
// together, and a layer is one boundary.
//
// Dependency.  Node tile t (T consecutive nodes) reads the AGG / PART rows of its nodes: the rows of the flat edges [ROWSTART[first node], ROWSTART[last] + n_last),
// i.e. of the edge tiles first_tile(t) .. last_tile(t) -- a plan-time-known, contiguous range.  `ready[t]` counts the edge tiles of that range whose rows are
// visible; the workgroup that has summed an edge tile adds 1 to the counter of every node tile its rows belong to (one or two).  A node tile is OWNED by the XCD
// whose contiguous range of edge tiles contains first_tile(t); the XCD's workgroups take its node tiles in ascending order (the order in which they become ready).
//
// No deadlock, whatever part of the grid is resident.  The edge schedule is static (workgroup b owns tiles (b >> 3) + j * stride of XCD b % 8's range), so a node
// tile may depend on a tile of a workgroup that has NOT STARTED -- when two launches from two streams share the chip (the sampler's sliced loop dispatches them
// interleaved: half of each grid resident), a workgroup that waited for such a tile would hold the very CU the missing workgroup needs.  Rule: a workgroup enters the
// node role only if EVERY workgroup of its XCD group has arrived (an arrival counter, bumped at kernel start); otherwise it leaves at once.  Workgroups that have
// arrived never wait inside their edge role, so every wait of the node role ends; and the group's LAST workgroup to arrive finds the count complete when it reaches
// its tail, so the queue is always drained by somebody.  The one node tile per XCD boundary whose rows end in the first tiles of the NEXT XCD's range additionally
// waits for those tiles' publication: they are the round-0 tiles of that XCD's workgroups, which start as CUs free up; XCD 7 waits for nobody, so the chain of
// such waits ends.  Every poll is bounded all the same: a counter that never completes raises GCDM_FLAG_TAIL instead of hanging the GPU.
//
// Visibility (MI355X guide, "Workgroup dispatch, XCD placement & inter-workgroup visibility").
//   * Same XCD: producer and consumer share the L2.  Producer: plain stores; every wave waits vmcnt(0) in front of the next workgroup barrier (the first barrier of
//     the NEXT edge tile -- the wait is free there: the loads in flight are consumed right behind that barrier; behind the persistent loop for the last tile), then
//     ONE thread adds to the counter with an agent-scope atomic.  Consumer: ONE thread polls with relaxed agent-scope loads, then a workgroup barrier, then plain
//     loads: the CU has never read these rows in this launch (the vector L1 is invalidated at the launch boundary), so no L1 line can be stale.
//   * Across XCDs (the boundary node tile): the producing workgroup of XCD x + 1 -- it knows: its tile's first row lies below rel_node_end[x + 1] -- does a full
//     agent-scope RELEASE (buffer_wbl2 sc1 + vmcnt(0)) before the add (a handful of workgroups per launch, once each, in their first round); the consumer does an
//     agent-scope ACQUIRE (buffer_inv sc1) behind the poll.
//   * Placement is an ASSUMPTION (workgroup b runs on XCD b % 8: observed, not promised).  It is checked in every launch: every workgroup ORs the bit of its
//     physical XCC id (s_getreg XCC_ID) into xcc_seen[b % 8]; a consumer that finds more than one bit in its group's word raises GCDM_FLAG_TAIL (the host then
//     re-runs with two launches per layer).
//   * The node role WRITES the next layer's gathered rows (PQ4 / VDI / VDJ) while other workgroups still gather the current layer's: the two layers use
//     different buffers (double-buffered by layer parity, gcdm_api.hip).
// Counters are self-resetting (the consumer of a node tile zeroes its counter; the last workgroup to leave zeroes the cursors and arrival words), so a captured
// step graph replays the same kernel arguments.
#pragma once
#include "gcdm_node_x3w.hip.h"

#define GCDM_FLAG_TAIL_BIT (16u | GCDM_FLAG_F16_RANGE_BIT)      // raised together with the range flag: every caller's existing re-run (fp32 mode, two launches per layer) repairs the result

struct TailArgs {
    NodeX3Args nx;                 // the layer's node kernel arguments (NodeX3Args::x3c repeats EdgeMsgX3Args::x3c)
    int* ready;                    // [node tiles] published edge tiles of the node tile's range; self-resetting
    int* qcur;                     // [0..7] next node tile of XCD x's queue; [8] workgroups that left; [9..16] xcc_seen; [17..24] workgroups of group x that have arrived
    const int* tab;                // [8 x + 0..3] XCD x: first owned node tile, number of owned node tiles, (unused), rel_node_end (an edge tile of x's range whose first
                                   //     row's node is below it belongs (also) to the boundary node tile owned by XCD x - 1 -> its rows are RELEASED; 0: none);
                                   // [64 .. 64 + node tiles) need[t] = edge tiles in the node tile's range | (1 << 16) when it reaches into the next XCD's (ACQUIRE)
    int num_wgs;
};
constexpr int TAIL_CTR_WORDS = 25;

template <int T>                   // nodes per node tile: 32 (node_tile_x3) or 64 (node_tile_x3w)
struct NodeTailRole {
    static constexpr bool ON = true;
    static constexpr int LOG_T = T == 64 ? 6 : 5;
    static constexpr int SPIN_LIMIT = 1 << 20;          // x ~0.5 us: far beyond any launch; then the flag, never a hang
    struct Args { EdgeMsgX3Args e; TailArgs t; };
    static constexpr int NODE_LDS = T == 64 ? NW_LDS_BYTES : NK_LDS_BYTES;
    static constexpr int WORD_OFF = EdgeGeo<64>::LDS_BYTES_X3 > NODE_LDS ? EdgeGeo<64>::LDS_BYTES_X3 : NODE_LDS;
    static constexpr int LDS_BYTES = WORD_OFF + 16;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

    static __device__ __forceinline__ int xcc_id() {
        int v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
        return v & 15;
    }
    static __device__ __forceinline__ int ld_relaxed(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

    // kernel start, one thread: this workgroup has arrived; its physical XCC into its group's word
    static __device__ __forceinline__ void arrived(const Args& lx, char* smem, int group) {
        ((int*)(smem + WORD_OFF))[1] = group;
        __hip_atomic_fetch_or(lx.t.qcur + 9 + group, 1 << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(lx.t.qcur + 17 + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ int rel_node_end(const Args& lx, int group) { return lx.t.tab[8 * group + 3]; }

    // one thread, behind a workgroup barrier that every wave entered with vmcnt(0): the rows of the edge tile whose first / last edge have the row nodes
    // `first_node` / `last_node` are in the L2 -> one more published tile for the node tile(s) of these rows
    static __device__ __forceinline__ void published(const Args& lx, int first_node, int last_node, int rel_end) {
        const int t0 = first_node >> LOG_T, t1 = last_node >> LOG_T;
        if (first_node < rel_end) {          // (also) rows of the previous XCD's boundary node tile: its consumer sits on another XCD -> write back this L2's dirty lines first
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (int t = t0; t <= t1; ++t) __hip_atomic_fetch_add(lx.t.ready + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // behind the persistent loop (all waves; LDS free; every wave has waited for its stores and passed a barrier)
    static __device__ __forceinline__ void tail(const Args& lx0, char* smem, bool have_prev, int prev_first, int prev_last, int rel_end) {
        int tid0 = threadIdx.x;
        asm volatile("" : "+v"(tid0));
        if (tid0 == 0 && have_prev) published(lx0, prev_first, prev_last, rel_end);
        // the XCD group from an LDS word (stashed at kernel start): carried across the edge role in an SGPR it gets spilled to a VGPR lane, and the VGPR reserved for that
        // is the one register the edge role's tile loop cannot spare (it then spills the thread index: two scratch reloads per tile)
        const int group = __builtin_amdgcn_readfirstlane(((const int*)(smem + WORD_OFF))[1]);       // (stashed by arrived(); the barrier in front of tail() has passed)
        // the kernel arguments and the thread index through OPAQUE copies (as the edge role does at the top of every tile): otherwise the compiler hoists the node
        // role's dozens of argument loads and lane-dependent offsets to the kernel entry and carries them -- spilled -- across the whole edge role
        typedef const Args __attribute__((address_space(4)))* karg_ptr;
        karg_ptr kp_ = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp_));
        const Args& lx = *(const Args*)kp_;
        const TailArgs& ta = lx.t;
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        int* word = (int*)(smem + WORD_OFF);
        const int q0 = ta.tab[8 * group], q1 = q0 + ta.tab[8 * group + 1];
        const int per_group = ta.num_wgs >> 3;
        for (;;) {
            __syncthreads();                                   // the previous node tile (or the edge role) is done with the LDS
            if (tid == 0) {
                int t = -1;
                // the node role may WAIT, so it is entered only when every workgroup of the group is running (see "No deadlock" above)
                if (ld_relaxed(ta.qcur + 17 + group) == per_group) {
                    const int k = __hip_atomic_fetch_add(ta.qcur + group, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (q0 + k < q1) {
                        t = q0 + k;
                        const int nd = ta.tab[64 + t], need = nd & 0xffff;
                        int spins = 0;
                        while (ld_relaxed(ta.ready + t) < need) {
                            __builtin_amdgcn_s_sleep(16);
                            if (++spins > SPIN_LIMIT) { atomicOr(ta.nx.base.flags_dev, GCDM_FLAG_TAIL_BIT); break; }
                        }
                        __hip_atomic_store(ta.ready + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (every producer of this counter has added: free for the next launch)
                        const int seen = ld_relaxed(ta.qcur + 9 + group);
                        if (seen & (seen - 1)) atomicOr(ta.nx.base.flags_dev, GCDM_FLAG_TAIL_BIT);             // the group's workgroups do not share ONE XCC: the same-L2 path is void
                        if (nd >> 16) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                       // rows of the next XCD's first tiles (released there)
                    }
                }
                *word = t;
            }
            __syncthreads();
            const int t = *word;
            if (t < 0) break;
            int tidn = threadIdx.x;
            asm volatile("" : "+v"(tidn));
            if constexpr (T == 64) node_tile_x3w(ta.nx, smem, t, tidn);
            else node_tile_x3<false, 2>(ta.nx, smem, t, tidn);
        }
        int tide = threadIdx.x;
        asm volatile("" : "+v"(tide));
        if (tide == 0) {
            karg_ptr ke_ = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();      // (afresh: an SGPR carried across the node tiles is spilled to a VGPR lane, see above)
            asm volatile("" : "+s"(ke_));
            const TailArgs& te = ((const Args*)ke_)->t;
            const int left = __hip_atomic_fetch_add(te.qcur + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (left == te.num_wgs - 1) {                      // the last workgroup out: every cursor has been advanced for the last time
                for (int i = 0; i < TAIL_CTR_WORDS; ++i) __hip_atomic_store(te.qcur + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
};
