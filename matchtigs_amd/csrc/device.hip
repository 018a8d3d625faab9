// device.hip -- MI355X (gfx950) device stage of the greedy-matchtigs engine.
//
// Replaces, for ALL sources at once, what the reference does one source at a time on the CPU:
//   * node classification                      /root/reference/src/implementation/greedytigs/mod.rs:222-255
//   * Dijkstra::shortest_path_lens (bounded)   call site greedytigs/mod.rs:324-335 (traitgraph-algo 8.1.2)
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   NodeBlock[V] 64-byte family block per node: <=4 inline (neighbour, clamped weight) pairs, degree, in-node flag,
//                its children's in-node flags and, while they fit, its children's out-edges. One aligned gather
//                fetches everything a path enumeration needs about a node AND its embedded children (the
//                cooperative levels read the first 32 bytes only). Nodes with more than 4 out-edges (never the
//                case in a de Bruijn graph) spill to a CSR side array. Built on the GPU from the edge arrays.
//   out_nodes[S] ascending source list (u32), mult[V] (i32) from classification.
//   pool[]       candidate keys (distance << 32 | node), per source contiguous and ascending.
//
// SSSP stage (integer, gather/latency bound, no MFMA) = a cascade of levels; a source a
// level cannot finish is appended to a device list by that kernel and re-run from scratch by the next:
//   level 0   sssp_enum_kernel: one LANE per source enumerates the bounded paths depth-first with a private LDS stack (a
//             (k-1)-ball of a unitig graph is almost a tree, so no visited table is needed); one 64-byte block gather per step,
//             self-refilling lanes, no barrier; results written straight to memory; fix_compact_kernel + sort_lists_kernel put
//             the lists that are not yet in Dijkstra order in order. Sources beyond its budgets go to level 1+.
//   level 1+  sssp_kernel: a workgroup takes a batch of BSRC sources and runs all their bounded searches
//             together as ONE label-correcting wavefront over a shared LDS open-addressing table keyed by
//             (local source, node) -> tentative distance (64-bit entries, ds_cmpst_b64 / ds_min_u64) with
//             one append-only frontier log. Work is balanced over frontier items, not over sources.
//             Distances are exact when the log drains (non-negative weights), so the result equals
//             Dijkstra's regardless of relaxation order. Emission ranks each source's targets by
//             (distance, node) in LDS and writes them contiguously. The last level keeps table and log
//             in a global workspace.
// Claim loop (greedytigs/mod.rs:301-523) on the device: replay_kernels.inc.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <future>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mtg_policy.h"
#include "device.hpp"
#include "hip_util.hpp"
#include "parallel.hpp"

namespace mtg {

#ifndef HIP_CHECK
#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MTG_DIE("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
    } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Node records
// ------------------------------------------------------------------------------------------------
enum : uint8_t { F_TARGET = 1, F_EXT = 2, F_SOURCE = 4, F_SELF_MIRROR = 8, F_REACH = 16 };  // (F_REACH: class bytes only -- a source that can reach an in-node within the bound)
constexpr uint32_t ODEG_REACH = 0x80000000u;  // bit 31 of odeg[n] (8:8 format): lb+(n) <= k - 1, set once with the device graph (build_lbx_kernel)

// One 64-byte "family" block per node: the node's own <= 4 out-edges (first 32 bytes: what the cooperative levels read) AND, for as
// many of its children as fit, the child's in-node flag and the child's out-edges (grandchildren of the node). A path enumeration
// therefore spends ONE 64-byte gather on a node and its embedded children instead of one gather each: 1.8 visited nodes per gather
// on the bench graph (35 fetched bytes per visited node instead of 64; a 32-byte record costs a 64-byte request anyway).
// Everything in it is a function of the graph alone, so it is built once with the device graph (build kernels below).
// The encoding is chosen so that the enumeration decodes a block without a single table lookup or slot -> parent search: an unused
// weight slot holds 0xFFFF (never within a bound < 0x8000), and a grandchild slot holds the weight of the whole two-edge path.
struct alignas(64) NodeBlock {
    uint32_t nbr[4];   // words 0-3: inline neighbours; if F_EXT: nbr[0]/nbr[1] = ext_begin lo/hi, nbr[2] = ext_count
    uint16_t w[4];     // words 4-5: weights clamped to min(w, k) (an edge with w >= k can never lie on a <= k-1 path); unused slots 0xFFFF.
                       //            k <= 255 ("8:8 format"): low byte = that weight, high byte = min(255, weight + lb+(child)), lb+(v) = the
                       //            distance from v to the nearest initial in-node BEYOND v: the goal-directed lower bound of what a
                       //            search needs v's own block for (whether v is an in-node itself is told by cmeta)
    uint8_t deg;       // word 6: inline degree 0..4 (0 if F_EXT)
    uint8_t flags;     //         F_TARGET (initial in-node, greedytigs/mod.rs:231-240) | F_EXT
    uint16_t cmeta;    //         bit j (0-3): child j is an in-node; bit 4+j: child j is NOT embedded below (it needs its own gather);
                       //         8:8 format: bit 8+t: the neighbour in gnbr[t] is an in-node
    uint32_t gnbr[6];  // words 7-12: out-neighbours of the embedded children, children in order, each child's edges in order
    uint16_t gw[6];    // words 13-15: weight of the path node -> child -> that neighbour, saturated at 0xFFFF; unused slots 0xFFFF
                       //            (8:8 format: low byte = path weight, high byte = path weight + lb+(that neighbour), both saturated at 255)
};
static_assert(sizeof(NodeBlock) == 64, "NodeBlock must be 64 bytes");
constexpr int GSLOTS = 6;

// compute_eulerian_superfluous_out_biedges (bigraph) and the classification rule of greedytigs/mod.rs:229-245
struct NodeClass { int32_t diff; uint8_t cls; };
__device__ __forceinline__ NodeClass classify_node(uint32_t out_deg, uint32_t out_deg_mirror, bool self_mirror) {
    NodeClass c;
    c.diff = self_mirror ? (int32_t)(out_deg & 1u) : (int32_t)out_deg - (int32_t)out_deg_mirror;  // in_degree(n) == out_degree(mirror(n))
    c.cls = self_mirror ? F_SELF_MIRROR : 0;
    if (self_mirror && c.diff != 0) c.cls |= F_TARGET | F_SOURCE;  // :231-236
    else if (c.diff > 0) c.cls |= F_TARGET;                        // :237-240
    else if (c.diff < 0) c.cls |= F_SOURCE;                        // :241-244
    return c;
}

// ------------------------------------------------------------------------------------------------
// Classification kernels (greedytigs/mod.rs:229-245): compact arrays only (out-degree, mirror -> multiplicity, class byte)
// ------------------------------------------------------------------------------------------------
constexpr int CLS_BLOCK = 256, CLS_PER = 8;  // nodes per workgroup = 2048 (the single-workgroup scan in between sees V / 2048 counts)
constexpr int CLS_NODES = CLS_BLOCK * CLS_PER;

__global__ __launch_bounds__(CLS_BLOCK) void classify_kernel(const uint32_t *odeg, const uint32_t *mirror, uint32_t n_nodes, int32_t *mult,
                                                             uint8_t *cls, uint32_t *block_counts, uint32_t *block_demand,
                                                             uint32_t *block_active) {
    __shared__ uint32_t wave_cnt[CLS_BLOCK / 64];
    __shared__ uint32_t wave_dem[CLS_BLOCK / 64];
    __shared__ uint32_t wave_act[CLS_BLOCK / 64];
    uint32_t cnt = 0, pos = 0, act = 0;
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {
        const uint64_t n64 = (uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x;
        if (n64 >= n_nodes) continue;
        const uint32_t n = (uint32_t)n64;
        const uint32_t m = mirror[n];
        const uint32_t on = odeg[n];
        const NodeClass c = classify_node(on & ~ODEG_REACH, m == n ? 0u : odeg[m] & ~ODEG_REACH, m == n);
        cnt += (c.cls & F_SOURCE) ? 1u : 0u;
        // (8:8 format) a source that can reach an in-node within the bound at all: the only ones the SSSP stage searches. The flag
        // is a function of the graph and rides in bit 31 of the out-degree word this kernel reads anyway.
        const bool reaches = (c.cls & F_SOURCE) && (on & ODEG_REACH);
        act += reaches ? 1u : 0u;
        mult[n] = c.diff;  // 0 for balanced nodes
        cls[n] = c.cls | (reaches ? F_REACH : 0);
        pos += c.diff > 0 ? (uint32_t)c.diff : 0u;
    }
    for (int dd = 32; dd >= 1; dd >>= 1) { pos += __shfl_down(pos, dd); cnt += __shfl_down(cnt, dd); act += __shfl_down(act, dd); }
    if ((threadIdx.x & 63) == 0) { wave_cnt[threadIdx.x >> 6] = cnt; wave_dem[threadIdx.x >> 6] = pos; wave_act[threadIdx.x >> 6] = act; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0, dm = 0, ac = 0;
        for (int i = 0; i < CLS_BLOCK / 64; i++) { s += wave_cnt[i]; dm += wave_dem[i]; ac += wave_act[i]; }
        block_counts[blockIdx.x] = s;
        block_demand[blockIdx.x] = dm;
        block_active[blockIdx.x] = ac;
    }
}

// single-workgroup exclusive scan of block_counts -> block_offsets (in place), total in *total_out; sum of block_demand in *demand_out.
// Launched with two workgroups, the second one scans a second array the same way (counts2 -> *total2_out).
__global__ __launch_bounds__(1024) void scan_blocks_kernel(uint32_t *counts, uint32_t n, unsigned long long *total_out,
                                                           const uint32_t *block_demand, unsigned long long *demand_out,
                                                           uint32_t *counts2 = nullptr, unsigned long long *total2_out = nullptr) {
    if (blockIdx.x == 1) { counts = counts2; total_out = total2_out; block_demand = nullptr; demand_out = nullptr; }
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    __shared__ unsigned long long dem_sum;
    if (threadIdx.x == 0) { carry = 0; dem_sum = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long dem = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? counts[i] : 0;
        dem += (block_demand && i < n) ? block_demand[i] : 0u;
        uint32_t incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        uint32_t wave_off = 0;
        for (int j = 0; j < wv; j++) wave_off += wave_tot[j];
        const uint32_t c = carry;
        if (i < n) counts[i] = c + wave_off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + wave_off + incl;
        __syncthreads();
    }
    for (int dd = 32; dd >= 1; dd >>= 1) dem += __shfl_down(dem, dd);
    if (lane == 0 && dem) atomicAdd(&dem_sum, dem);
    __syncthreads();
    if (threadIdx.x == 0) { *total_out = carry; if (demand_out) *demand_out = dem_sum; }
}

// out_nodes = the sources, ascending; and the sources that can reach an in-node within the bound at all (F_REACH in their class
// byte; 8:8 format), in order: act_index = their positions in out_nodes, act_node = their nodes -- the only sources the SSSP stage
// searches (the others have an empty candidate list by construction). No gather and no pass of its own: the flag arrives with the
// class byte this pass reads anyway.
__global__ __launch_bounds__(CLS_BLOCK) void compact_sources_kernel(const uint8_t *cls, uint32_t n_nodes,
                                                                    const uint32_t *block_offsets, uint32_t *out_nodes,
                                                                    const uint32_t *block_act_offsets, uint32_t *act_index, uint32_t *act_node) {
    __shared__ uint32_t wave_cnt[CLS_PER][CLS_BLOCK / 64], wave_act[CLS_PER][CLS_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long bal[CLS_PER], abal[CLS_PER];
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {
        const uint64_t n = (uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x;
        const uint8_t c = n < n_nodes ? cls[n] : (uint8_t)0;
        bal[p] = __ballot((c & F_SOURCE) != 0);
        abal[p] = __ballot((c & F_REACH) != 0);
        if (lane == 0) { wave_cnt[p][wv] = (uint32_t)__popcll(bal[p]); wave_act[p][wv] = (uint32_t)__popcll(abal[p]); }
    }
    __syncthreads();
    uint32_t off = block_offsets[blockIdx.x], aoff = block_act_offsets[blockIdx.x];
#pragma unroll
    for (int p = 0; p < CLS_PER; p++) {  // ascending: lane order within a wave, wave order within a pass, pass order, block order
        for (int j = 0; j < CLS_BLOCK / 64; j++) {
            if (j == wv && ((bal[p] >> lane) & 1ull)) {
                const uint32_t node = (uint32_t)((uint64_t)blockIdx.x * CLS_NODES + (uint64_t)p * CLS_BLOCK + threadIdx.x);
                const uint32_t idx = off + (uint32_t)__popcll(bal[p] & ((1ull << lane) - 1ull));
                out_nodes[idx] = node;
                if ((abal[p] >> lane) & 1ull) {
                    const uint32_t pos = aoff + (uint32_t)__popcll(abal[p] & ((1ull << lane) - 1ull));
                    act_index[pos] = idx;
                    act_node[pos] = node;
                }
            }
            off += wave_cnt[p][j];
            aoff += wave_act[p][j];
        }
    }
}

__global__ void export_live_kernel(const uint8_t *cls, uint32_t n_nodes, uint8_t *live) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) live[n] = (cls[n] & F_TARGET) ? 1 : 0;
}

// The searched sources of a launch over the sources [src_begin, src_end): the part of the classification's list of sources that can
// reach an in-node (act_index, ascending) inside that range -- first entry and number, by a wave-wide 64-ary search (one wave; each
// round probes 64 evenly spaced entries: four rounds for ten million). *act_begin = first entry, *act_count = their number.
__global__ __launch_bounds__(64) void active_range_kernel(const uint32_t *act_index, const unsigned long long *n_act_total, uint64_t src_begin,
                                                          uint64_t src_end, unsigned long long *act_begin, unsigned long long *act_count) {
    const int lane = threadIdx.x;
    const unsigned long long n = *n_act_total;
    auto lower_bound = [&](uint64_t key) -> unsigned long long {  // first i in [0, n] with act_index[i] >= key
        unsigned long long lo = 0, hi = n;
        while (lo < hi) {
            const unsigned long long chunk = (hi - lo + 63) / 64;
            const unsigned long long idx = lo + (unsigned long long)lane * chunk;
            const bool ge = idx >= hi || (uint64_t)act_index[idx] >= key;  // (monotone over the lanes)
            const unsigned long long m = __ballot(ge);
            const int first = m ? __builtin_ctzll(m) : 64;
            if (first == 0) break;  // act_index[lo] >= key
            const unsigned long long new_lo = lo + (unsigned long long)(first - 1) * chunk + 1;
            const unsigned long long at_first = lo + (unsigned long long)first * chunk;
            hi = (first < 64 && at_first < hi) ? at_first : hi;
            lo = new_lo;
        }
        return lo;
    };
    const unsigned long long b = lower_bound(src_begin), e = lower_bound(src_end);
    if (lane == 0) { *act_begin = b; *act_count = e - b; }
}

// ------------------------------------------------------------------------------------------------
// Device graph build: original edge arrays -> family blocks (+ CSR spill for nodes with more than 4 out-edges)
// Out-edges are kept in edge-id order per node (slots are claimed in any order, then sorted by edge id), so the content is
// independent of thread timing.
// ------------------------------------------------------------------------------------------------
// (the count and the edge's slot at its from-node come from the same atomic: `rank` is read back, coalesced, by build_fill_kernel --
// one random atomic per edge instead of two: 18.5 -> see DESIGN 3.1)
__global__ void build_count_kernel(const uint32_t *e_from, uint64_t n_edges, uint32_t *odeg, uint32_t *rank) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) rank[e] = atomicAdd(&odeg[e_from[e]], 1u);
}
__global__ void build_ext_need_kernel(const uint32_t *odeg, uint64_t n_nodes, uint32_t *ext_need) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < n_nodes) ext_need[n] = odeg[n] > 4 ? odeg[n] : 0u;
}
// every edge leaves its EDGE ID in the slot of its from-node the counting pass gave it (inline slot or spill position)
__global__ void build_fill_kernel(const uint32_t *e_from, uint64_t n_edges, const uint32_t *odeg, const unsigned long long *ext_off,
                                  const uint32_t *rank, NodeBlock *blocks, uint32_t *ext_col) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const uint32_t f = e_from[e];
    const uint32_t slot = rank[e];
    if (odeg[f] > 4) ext_col[ext_off[f] + slot] = (uint32_t)e;
    else blocks[f].nbr[slot] = (uint32_t)e;
}
// per node: edge ids ascending -> (neighbour, clamped weight); degree, own class flag. The head of edge e is mirror(from(e ^ 1)) (the
// mirror edge runs mirror(to) -> mirror(from), clib.rs:244-248) and both edges of unitig u = e >> 1 weigh w_unitig[u], so neither a
// head array nor per-edge weights are uploaded. adj0 (optional): the node's out-edges in ascending id at row0[n] -- the buckets of
// the original darts the finishing stages keep with the graph (finish_device.hip), a by-product of the sort here.
__global__ void build_nodes_kernel(uint64_t n_nodes, const uint32_t *odeg, const uint32_t *mirror, const unsigned long long *ext_off,
                                   const uint32_t *e_from, const uint16_t *w_unitig, NodeBlock *blocks, uint32_t *ext_col, uint16_t *ext_w,
                                   const uint32_t *row0, uint32_t *adj0) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t dg = odeg[n];
    const uint32_t m = mirror[n];
    const NodeClass c = classify_node(dg, m == n ? 0u : odeg[m], m == n);
    NodeBlock b;
    uint32_t *bw = reinterpret_cast<uint32_t *>(&b);
#pragma unroll
    for (int i = 0; i < 16; i++) bw[i] = 0;
    b.flags = c.cls & F_TARGET;
    b.cmeta = 0x00F0;  // no child embedded (build_children_kernel fills this in)
    for (int j = 0; j < 4; j++) b.w[j] = 0xFFFFu;
    for (int t = 0; t < GSLOTS; t++) b.gw[t] = 0xFFFFu;
    if (dg <= 4) {
        uint32_t ids[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (uint32_t j = 0; j < dg; j++) ids[j] = blocks[n].nbr[j];
        auto cswap = [&](int x, int y) { if (ids[x] > ids[y]) { const uint32_t t = ids[x]; ids[x] = ids[y]; ids[y] = t; } };
        cswap(0, 1); cswap(2, 3); cswap(0, 2); cswap(1, 3); cswap(1, 2);
        for (uint32_t j = 0; j < dg; j++) { b.nbr[j] = mirror[e_from[ids[j] ^ 1u]]; b.w[j] = w_unitig[ids[j] >> 1]; }
        if (adj0)
            for (uint32_t j = 0; j < dg; j++) adj0[row0[n] + j] = ids[j];
        b.deg = (uint8_t)dg;
    } else {
        const unsigned long long off = ext_off[n];
        for (uint32_t i = 1; i < dg; i++) {  // insertion sort of the spill segment by edge id
            const uint32_t key = ext_col[off + i];
            uint32_t q = i;
            while (q > 0 && ext_col[off + q - 1] > key) { ext_col[off + q] = ext_col[off + q - 1]; q--; }
            ext_col[off + q] = key;
        }
        for (uint32_t i = 0; i < dg; i++) {
            const uint32_t e = ext_col[off + i];
            if (adj0) adj0[row0[n] + i] = e;
            ext_col[off + i] = mirror[e_from[e ^ 1u]];
            ext_w[off + i] = w_unitig[e >> 1];
        }
        b.flags |= F_EXT;
        b.nbr[0] = (uint32_t)(off & 0xFFFFFFFFull);
        b.nbr[1] = (uint32_t)(off >> 32);
        b.nbr[2] = dg;
    }
    blocks[n] = b;
}
// ------------------------------------------------------------------------------------------------
// Goal-directed lower bounds (k <= 255): lb(v) = distance from v to the nearest initial in-node, 0 for an in-node, "infinite" beyond
// k - 1. A function of the graph alone, like the in-node flags. A search at node u with distance d only ever needs the successor v
// over an edge of weight w if d + w + lb(v) <= k - 1: every in-node behind v is at least that far from the source, so dropping v
// can never drop a candidate -- the lists stay exactly Dijkstra's (greedytigs/mod.rs:324-335; the reference truncates its search
// too, by target_amount). On the bench graph 7 of 10 sources have no in-node within the bound at all and the remaining searches
// visit a third of their balls.
// dist(v -> t) in G = dist(mirror t -> mirror v) in G (every edge has its mirror edge with the same weight), so ONE bounded
// multi-source search from the mirrors of the in-nodes gives D(x) = lb(mirror x). Dial's buckets without queues: weights are >= 1,
// so the nodes with D == r are final in round r; round r relaxes their out-edges with atomicMin. k - 1 rounds of a streaming pass
// over D plus one 32-byte gather per reached node, once per device graph.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LB_INF = 0xFFFFFFFFu;
__global__ void lb_init_kernel(const uint32_t *odeg, const uint32_t *mirror, uint64_t n_nodes, uint32_t *D) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t m = mirror[n];
    const NodeClass c = classify_node(odeg[m], m == n ? 0u : odeg[n], m == n);  // class of mirror(n)
    D[n] = (c.cls & F_TARGET) ? 0u : LB_INF;
}
// (first halves still hold plain 16-bit weights here: build_lb_kernel rewrites them afterwards)
__global__ void lb_round_kernel(const NodeBlock *blocks, const uint32_t *ext_col, const uint16_t *ext_w, uint64_t n_nodes, uint32_t r, uint32_t K1,
                                uint32_t *D) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes || D[n] != r) return;
    const uint4 *rp = reinterpret_cast<const uint4 *>(blocks + n);
    const uint4 lo = rp[0], hi = rp[1];
    const uint32_t nb[4] = {lo.x, lo.y, lo.z, lo.w};
    const uint32_t wt[4] = {hi.x & 0xFFFFu, hi.x >> 16, hi.y & 0xFFFFu, hi.y >> 16};
    if ((hi.z >> 8) & F_EXT) {
        const uint64_t b = (uint64_t)lo.x | ((uint64_t)lo.y << 32);
        for (uint32_t e = 0; e < lo.z; e++) {
            const uint32_t nd = r + ext_w[b + e];
            if (nd <= K1) atomicMin(&D[ext_col[b + e]], nd);
        }
    } else {
        const uint32_t dg = hi.z & 0xFFu;
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t nd = r + wt[j];
            if (j < dg && nd <= K1) atomicMin(&D[nb[j]], nd);
        }
    }
}
__global__ void lb_mirror_kernel(const uint32_t *mirror, const uint32_t *D, uint64_t n_nodes, uint8_t *lb8) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t v = D[mirror[n]];
    lb8[n] = (uint8_t)(v < 255u ? v : 255u);
}
// Pass 1: lbx[n] = lb(n) | lb+(n) << 8, where lb+(n) = min over the out-edges n -> c of weight + lb(c) is the distance from n to the
// nearest in-node BEYOND n (for a node that is not an in-node itself lb+ = lb; for an in-node lb = 0 and lb+ says what a search that
// has recorded n still needs n's block for). bit 31 of odeg[n] (ODEG_REACH) is set iff lb+(n) <= k - 1: a source without that has an empty candidate list and
// is never searched. (First halves still hold plain 16-bit weights here.)
__global__ void build_lbx_kernel(uint64_t n_nodes, const NodeBlock *blocks, const uint32_t *ext_col, const uint16_t *ext_w, const uint8_t *lb8, uint32_t K1,
                                 uint16_t *lbx, uint32_t *odeg) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t *me = reinterpret_cast<const uint32_t *>(blocks + n);
    const uint32_t meta = me[6];
    uint32_t best = 255u;
    if ((meta >> 8) & F_EXT) {
        const uint64_t b = (uint64_t)me[0] | ((uint64_t)me[1] << 32);
        for (uint32_t e = 0; e < me[2]; e++) best = min(best, (uint32_t)ext_w[b + e] + lb8[ext_col[b + e]]);
    } else {
        const uint32_t dg = meta & 0xFFu;
        for (uint32_t j = 0; j < dg; j++) {
            const uint32_t w = (me[4 + (j >> 1)] >> ((j & 1u) * 16)) & 0xFFFFu;  // <= k <= 255
            best = min(best, w + lb8[me[j]]);
        }
    }
    lbx[n] = (uint16_t)(lb8[n] | (best << 8));
    if (best <= K1) odeg[n] |= ODEG_REACH;  // (read by the classification with the degree; every other reader of odeg runs before this kernel or masks)
}
// second half of every block: the children's in-node flags and, while they fit, the children's out-edges.
// W8 = false: plain 16-bit weights (k > 255, or a device graph built without lower bounds).
// W8 = true: the 8:8 format, written in the same pass (round 5: one kernel where build_lb_kernel rewrote the first halves and this
// kernel then read the children's rewritten halves). Own first half: low byte the weight, high byte weight + lb+(child) -- what a
// search needs the CHILD'S BLOCK for; whether the child is an in-node itself (lb = 0) travels in cmeta bit j, so the parent's step
// records that candidate and the child's gather only happens when something lies beyond it ("leaf" in-nodes, a quarter of all
// visits on the bench graph, cost no gather). Second half: path weight | path weight + lb+(grandchild), both saturated at 255 (> any
// bound of this format), and the grandchild's in-node flag in cmeta bit 8 + t. lb and lb+ of children and grandchildren come from
// lbx[] (build_lbx_kernel), never from another node's block: while this kernel runs a node's first half is read by its parents'
// threads and rewritten by its own, and either version decodes to the same neighbours, degree, flags and LOW bytes (a plain
// 16-bit weight is <= k <= 255, an unused slot is 0xFFFF in both formats; the three words are written whole). A spilled adjacency
// keeps plain weights (no pruning behind such a node).
template <bool W8>
__global__ void build_children_kernel(uint64_t n_nodes, NodeBlock *blocks, const uint16_t *lbx) {
    const uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    const uint32_t *me = reinterpret_cast<const uint32_t *>(blocks + n);
    const uint32_t meta = me[6];
    if ((meta >> 8) & F_EXT) return;
    const uint32_t dg = meta & 0xFFu;
    const uint32_t nb[4] = {me[0], me[1], me[2], me[3]};
    const uint32_t w45[2] = {me[4], me[5]};
    uint32_t cmeta = 0, used = 0;
    uint32_t gn[GSLOTS];
    uint32_t gwt[GSLOTS];
    uint32_t slot[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
#pragma unroll
    for (int i = 0; i < GSLOTS; i++) { gn[i] = 0; gwt[i] = 0xFFFFu; }
    bool open = true;  // children are embedded in order while they fit
    for (uint32_t j = 0; j < 4; j++) {
        if (j >= dg) continue;
        const uint32_t *ch = reinterpret_cast<const uint32_t *>(blocks + nb[j]);  // first half only
        const uint32_t cmt = ch[6];
        const uint32_t cflags = (cmt >> 8) & 0xFFu, cdeg = cmt & 0xFFu;
        const uint32_t sj = (w45[j >> 1] >> ((j & 1u) * 16)) & 0xFFFFu;
        const uint32_t wj = W8 ? (sj & 0xFFu) : sj;
        if (cflags & F_TARGET) cmeta |= 1u << j;
        if constexpr (W8) slot[j] = (min(255u, wj + ((uint32_t)lbx[nb[j]] >> 8)) << 8) | wj;
        if (open && !(cflags & F_EXT) && used + cdeg <= (uint32_t)GSLOTS) {
            for (uint32_t t = 0; t < cdeg; t++) {
                const uint32_t gc = ch[t];
                gn[used + t] = gc;
                const uint32_t st = (ch[4 + (t >> 1)] >> ((t & 1u) * 16)) & 0xFFFFu;
                if constexpr (W8) {
                    const uint32_t x = lbx[gc], wt = st & 0xFFu;
                    gwt[used + t] = min(255u, wj + wt) | (min(255u, wj + wt + (x >> 8)) << 8);
                    if ((x & 0xFFu) == 0) cmeta |= 1u << (8 + used + t);  // the grandchild is an in-node
                } else {
                    const uint32_t sum = wj + st;
                    gwt[used + t] = sum < 0xFFFFu ? sum : 0xFFFFu;
                }
            }
            used += cdeg;
        } else {
            open = false;
            cmeta |= 16u << j;
        }
    }
    // The WHOLE block is stored, the unchanged neighbour words included (four 16-byte stores per thread): a line that leaves the L2
    // partly written is a masked write, which the ECC-protected HBM turns into a read-modify-write (round 5; the same finding as the
    // claim replay's records, replay_kernels.inc). Readers of this node's first half see old or new words, which decode alike (above).
    uint32_t o[16];
    o[0] = nb[0]; o[1] = nb[1]; o[2] = nb[2]; o[3] = nb[3];
    o[4] = W8 ? (slot[0] | (slot[1] << 16)) : w45[0];
    o[5] = W8 ? (slot[2] | (slot[3] << 16)) : w45[1];
    o[6] = (meta & 0xFFFFu) | (cmeta << 16);
#pragma unroll
    for (int i = 0; i < GSLOTS; i++) o[7 + i] = gn[i];
#pragma unroll
    for (int i = 0; i < GSLOTS / 2; i++) o[13 + i] = gwt[2 * i] | (gwt[2 * i + 1] << 16);
    uint4 *out = reinterpret_cast<uint4 *>(blocks + n);
#pragma unroll
    for (int q = 0; q < 4; q++) out[q] = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// ------------------------------------------------------------------------------------------------
// SSSP kernel
// ------------------------------------------------------------------------------------------------
constexpr uint32_t CAND_OVERFLOW = 0xFFFFFFFFu;
constexpr unsigned long long TBL_EMPTY = 0xFFFFFFFFFFFFFFFFull;
// table entry: [63:54] local source (10 bits) | [53:22] node (32) | [21:1] distance (21) | [0] 1 = not (yet) known to be a target
constexpr int ENT_SRC_SHIFT = 54, ENT_NODE_SHIFT = 22;
constexpr unsigned long long ENT_DIST_MASK = 0x1FFFFFull;

enum Counter : int {
    C_BATCH = 0,     // (unused since kernel v2: batches are strided statically)
    C_POOL = 1,      // pool cursor (keys)
    C_OVERFLOW = 2,  // number of overflowed sources
    C_SETTLED = 3,
    C_RELAXED = 4,
    C_EMITTED = 5,
    C_ATTEMPTS = 6,
    C_OVF_LIST = 7,  // cursor of the overflow source list
    C_FIX = 8,       // enumeration level: number of candidate lists its post-pass has to put in order
    C_PUSHES = 9,    // COUNT: frontier-log items of all finished batches
    C_MAX_LOG = 10,  // COUNT: longest frontier log of one batch
    C_MAX_ENT = 11,  // COUNT: most table entries of one batch
    C_DEMAND = 12,   // classification: sum of the positive multiplicities (bounds the number of pairs)
    C_FIX_CLASS0 = 13,  // enumeration level's post-pass: number of work-list entries per length class (13, 14, 15)
    C_FIX_CURSOR0 = 16, // ... and the cursors of its compaction (16, 17, 18)
    C_ACTIVE = 19,      // sources of the launch's range that can reach an in-node within the bound (the only ones searched)
    C_ACT_BEGIN = 20,   // ... and where they start in the classification's list of such sources (active_range_kernel)
    C_COUNT = 24
};

struct SsspArgs {
    const NodeBlock *recs;  // family blocks (the cooperative levels read the first 32 bytes of each)
    const uint32_t *ext_col;
    const uint16_t *ext_w;
    const uint32_t *sources;     // out_nodes (ascending)
    const uint32_t *src_index;   // optional list of absolute source indices to process (re-runs); null = contiguous range
    uint64_t n_items;            // number of sources in this launch
    uint64_t src_begin;          // first absolute source index (outputs are indexed by abs - src_begin)
    uint32_t K1;                 // bound k-1 (inclusive)
    unsigned long long *pool;
    uint64_t pool_cap;
    unsigned long long *cand_start;
    uint32_t *cand_count;
    unsigned long long *counters;
    unsigned long long *ws;      // global workspace (GLOBAL_WS levels)
    uint64_t ws_stride;          // 64-bit words per block
    uint32_t *ovf_list;          // out: absolute indices of the sources this launch could not finish (cursor: C_OVERFLOW)
    uint32_t *fix_list;          // out (enumeration level): post-pass work list in chunks of ENUM_FIX_CHUNK slots (cursor: C_FIX): slot 0 = the
                                 // chunk's length class, then indices (relative to src_begin) of lists of that class, FIX_NONE = unused
    uint32_t wmask;              // inline weight slots: 0xFF in the 8:8 format (k <= 255: weight | weight + lower bound << 8), else 0xFFFF
    uint32_t prune;              // cooperative levels: 1 = skip a successor whose lower bound puts every in-node behind it beyond the bound
    const uint32_t *act_index;   // enumeration level with pruning: the classification's list of sources that can reach an in-node (absolute
    const uint32_t *act_node;    // indices, ascending) and their nodes; the launch's part of it is counters[C_ACT_BEGIN] .. + counters[C_ACTIVE]
                                 // (neither number travels to the host before the launch)
};

template <bool GLOBAL_WS>
struct Mem {
    // LDS: workgroup scope suffices. Global workspace: agent scope so that loads bypass the CU's L1
    // (atomics execute in L2 and do not refresh an L1-resident line).
    static constexpr int SCOPE = GLOBAL_WS ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    template <typename T> static __device__ __forceinline__ T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
    template <typename T> static __device__ __forceinline__ void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
    static __device__ __forceinline__ unsigned long long cas(unsigned long long *p, unsigned long long expect, unsigned long long v) {
        __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE);
        return expect;  // previous value
    }
    static __device__ __forceinline__ unsigned long long fmin(unsigned long long *p, unsigned long long v) {
        return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, SCOPE);
    }
    static __device__ __forceinline__ void fand(unsigned long long *p, unsigned long long v) {
        __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, SCOPE);
    }
};

template <int LOGH>
__device__ __forceinline__ uint32_t tbl_hash(uint32_t src, uint32_t node) {
    const uint32_t x = node * 0x9E3779B1u + src * 0x85EBCA77u;
    return (x ^ (x >> 15)) * 0x2C1B3C6Du >> (32 - LOGH);
}

// returns > 0 improved (2 = new entry, 1 = existing entry; slot set), 0 not improved, -1 table full
template <int LOGH, bool GLOBAL_WS>
__device__ __forceinline__ int tbl_relax(unsigned long long *table, uint32_t src, uint32_t node, uint32_t dist, uint32_t &slot) {
    constexpr uint32_t H = 1u << LOGH;
    constexpr int MAX_PROBE = H < 256 ? (int)H : 256;
    const unsigned long long key = ((unsigned long long)src << ENT_SRC_SHIFT) | ((unsigned long long)node << ENT_NODE_SHIFT);
    const unsigned long long val = key | ((unsigned long long)dist << 1) | 1ull;
    uint32_t h = tbl_hash<LOGH>(src, node);
    for (int probe = 0; probe < MAX_PROBE; probe++) {
        unsigned long long cur = Mem<GLOBAL_WS>::ld(&table[h]);
        if (cur == TBL_EMPTY) {
            cur = Mem<GLOBAL_WS>::cas(&table[h], TBL_EMPTY, val);
            if (cur == TBL_EMPTY) { slot = h; return 2; }
        }
        if ((cur >> ENT_NODE_SHIFT) == (key >> ENT_NODE_SHIFT)) {
            if (((cur >> 1) & ENT_DIST_MASK) <= dist) return 0;
            const unsigned long long old = Mem<GLOBAL_WS>::fmin(&table[h], val);
            slot = h;
            return (((old >> 1) & ENT_DIST_MASK) > dist) ? 1 : 0;
        }
        h = (h + 1) & (H - 1);
    }
    return -1;
}

// ---- SSSP kernel, version 2 ------------------------------------------------------------------
// Changes against the first correct kernel (profiles/r01_baseline_*): no global atomic per batch any more
// (batches are strided statically over the persistent grid; pool space is taken in block-local chunks;
// counters are flushed once per block), the frontier is ONE append-only log so that emission and the
// table clean-up walk only the touched slots instead of scanning/clearing the whole table per batch.
constexpr unsigned long long POOL_CHUNK = 2048;  // keys per block-local pool chunk

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool COUNT, bool GLOBAL_WS>
struct SsspLds {
    static constexpr uint32_t H = 1u << LOGH;
    unsigned long long table[GLOBAL_WS ? 1 : H];
    unsigned long long stage[GLOBAL_WS ? 1 : SCAP];  // emission staging: keys per source segment
    uint32_t log[GLOBAL_WS ? 1 : QCAP];              // every push of the batch, in push order (rounds are ranges)
    uint16_t stage_src[GLOBAL_WS ? 1 : SCAP];
    uint32_t srcnode[BSRC];
    uint32_t cnt[BSRC];
    uint32_t off[BSRC];
    uint32_t fill[BSRC];
    uint32_t tail;   // next free log position (may run past QCAP -> overflow)
    uint32_t end;    // snapshot of tail taken between two barriers
    uint32_t ovf;
    uint32_t total;  // keys emitted by this batch
    unsigned long long base;                    // pool position of this batch's keys
    unsigned long long chunk_next, chunk_end;   // block-local pool chunk
    unsigned long long bt_settled, bt_relaxed, bt_attempts;              // per batch (COUNT)
    unsigned long long st_settled, st_relaxed, st_attempts, st_emitted;  // per block (COUNT)
    unsigned long long st_pushes, st_max_log, st_max_ent;
};

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool COUNT, bool GLOBAL_WS>
__global__ __launch_bounds__(BLOCK) void sssp_kernel(SsspArgs a) {
    static_assert(BSRC <= 512 && BSRC <= BLOCK && BLOCK >= 64, "BSRC/BLOCK limits");
    static_assert(LOGH <= 22, "slot must fit the log item");
    constexpr uint32_t H = 1u << LOGH;
    constexpr int HINT_BITS = 32 - LOGH;  // distance (hint) bits of a log item
    constexpr uint32_t HINT_MASK = (1u << HINT_BITS) - 1u;
    // With >= 16 hint bits the hint IS the distance (k-1 <= 65535): the push that carries an entry's final
    // distance is unique, so the log doubles as the list of live table slots.
    constexpr bool LOG_EMIT = HINT_BITS >= 16;
    using M = Mem<GLOBAL_WS>;
    __shared__ SsspLds<BLOCK, LOGH, QCAP, SCAP, BSRC, COUNT, GLOBAL_WS> s;

    unsigned long long *table, *stage;
    uint32_t *log;
    uint16_t *stage_src;
    if constexpr (GLOBAL_WS) {
        unsigned long long *base = a.ws + (uint64_t)blockIdx.x * a.ws_stride;
        table = base;
        stage = base + H;
        log = reinterpret_cast<uint32_t *>(base + H + SCAP);
        stage_src = reinterpret_cast<uint16_t *>(base + H + SCAP + (QCAP + 1) / 2);
    } else {
        table = s.table;
        stage = s.stage;
        log = s.log;
        stage_src = s.stage_src;
    }

    const int tid = threadIdx.x;
    const uint64_t n_batches = (a.n_items + BSRC - 1) / BSRC;

    for (uint32_t i = tid; i < H; i += BLOCK) M::st(&table[i], TBL_EMPTY);
    if (tid == 0) {
        s.chunk_next = 0; s.chunk_end = 0;
        s.st_settled = 0; s.st_relaxed = 0; s.st_attempts = 0; s.st_emitted = 0;
        s.st_pushes = 0; s.st_max_log = 0; s.st_max_ent = 0;
    }
    __syncthreads();

    for (uint64_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
        const uint64_t item0 = batch * BSRC;
        const int nsrc = (int)min((uint64_t)BSRC, a.n_items - item0);
        // ---- init (the table is clean here) ----
        if (tid < BSRC) { s.cnt[tid] = 0; s.fill[tid] = 0; }
        if (tid == 0) { s.tail = (uint32_t)nsrc; s.ovf = 0; s.bt_settled = 0; s.bt_relaxed = 0; s.bt_attempts = 0; }
        __syncthreads();
        uint32_t begin = 0;
        if (tid < nsrc) {
            const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
            const uint32_t node = a.sources[abs_idx];
            s.srcnode[tid] = node;
            uint32_t slot = 0;
            const int r = tbl_relax<LOGH, GLOBAL_WS>(table, (uint32_t)tid, node, 0u, slot);
            if (r < 0) s.ovf = 1;
            M::st(&log[tid], (slot << HINT_BITS) | 0u);
        }
        __syncthreads();
        if (tid == 0) s.end = s.ovf ? 0u : (uint32_t)nsrc;  // loop bounds are only ever published between two barriers
        __syncthreads();

        // ---- label-correcting rounds: round r processes log[begin, end), pushes append at tail ----
        uint32_t end = s.end;
        for (int round = 0; begin < end; round++) {  // uniform: `end` is a snapshot published by thread 0
            // one lane per (frontier item, inline edge j): the four lanes of an item read the same 32-byte record
            // (one memory request) and each relaxes one edge, instead of one lane running four divergent table updates
            for (uint32_t w = begin * 4 + tid; w < end * 4; w += BLOCK) {
                const uint32_t i = w >> 2, j = w & 3u;
                const uint32_t item = M::ld(&log[i]);
                const uint32_t slot = item >> HINT_BITS;
                const unsigned long long e = M::ld(&table[slot]);
                const uint32_t d = (uint32_t)((e >> 1) & ENT_DIST_MASK);
                if ((d & HINT_MASK) != (item & HINT_MASK)) continue;  // superseded by a shorter distance
                const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                const uint32_t *rw = reinterpret_cast<const uint32_t *>(a.recs + node);
                const uint32_t meta = rw[6];  // deg | flags << 8
                const uint32_t nb_j = rw[j];  // issued together with meta: one memory latency per round, not two
                const uint32_t w_pair = rw[4 + (j >> 1)];
                const uint32_t flags = (meta >> 8) & 0xFFu;
                if (j == 0 && (flags & F_TARGET)) M::fand(&table[slot], ~1ull);  // node property: confirmed in-node
                uint32_t pushed_ovf = 0;
                auto relax = [&](uint32_t nb, uint32_t wt, uint32_t wl) {  // wl = weight + lower bound of nb (= wt without one)
                    const uint32_t nd = d + wt;
                    if (nd > a.K1) return;
                    if (a.prune && d + wl > a.K1) return;  // no in-node behind nb within the bound (build_lb_kernel)
                    uint32_t nslot = 0;
                    const int r = tbl_relax<LOGH, GLOBAL_WS>(table, src, nb, nd, nslot);
                    if (r > 0) {
                        const uint32_t pos = atomicAdd(&s.tail, 1u);
                        if (pos < (uint32_t)QCAP) M::st(&log[pos], (nslot << HINT_BITS) | (nd & HINT_MASK));
                        else pushed_ovf = 1;
                    } else if (r < 0) pushed_ovf = 1;
                };
                if (!(flags & F_EXT)) {
                    const uint32_t deg = meta & 0xFFu;
                    const uint32_t wslot = (w_pair >> ((j & 1u) * 16)) & 0xFFFFu;
                    // (8:8 format: the high byte is weight + lb+(child), what the child's block is needed for beyond the child itself -- an
                    // in-node child, cmeta bit j, is needed for itself: this kernel emits a node from its own entry)
                    if (j < deg) relax(nb_j, wslot & a.wmask, a.wmask == 0xFFu ? (((meta >> (16 + j)) & 1u) ? (wslot & 0xFFu) : wslot >> 8) : wslot);
                    if constexpr (COUNT) { if (j == 0) atomicAdd(&s.bt_attempts, (unsigned long long)deg); }
                } else if (j == 0) {  // spilled adjacency (more than 4 out-edges): one lane walks the list
                    const uint64_t eb = ((uint64_t)rw[1] << 32) | nb_j;  // j == 0: nb_j is word 0
                    const uint32_t deg = rw[2];
                    for (uint32_t q = 0; q < deg; q++) relax(a.ext_col[eb + q], a.ext_w[eb + q], a.ext_w[eb + q]);
                    if constexpr (COUNT) atomicAdd(&s.bt_attempts, (unsigned long long)deg);
                }
                if (pushed_ovf) s.ovf = 1;
            }
            __syncthreads();
            if (tid == 0) {
                if (round > (1 << 20)) s.ovf = 1;
                s.end = s.ovf ? end : min(s.tail, (uint32_t)QCAP);  // on overflow: no progress -> the loop ends
            }
            __syncthreads();
            begin = end;
            end = s.end;
        }
        __syncthreads();
        const uint32_t n_log = min(s.tail, (uint32_t)QCAP);

        // visits every live table entry exactly once: f(entry)
        auto for_each_entry = [&](auto &&f) {
            if constexpr (LOG_EMIT) {
                for (uint32_t i = tid; i < n_log; i += BLOCK) {
                    const uint32_t item = M::ld(&log[i]);
                    const unsigned long long e = M::ld(&table[item >> HINT_BITS]);
                    if (((uint32_t)((e >> 1) & ENT_DIST_MASK) & HINT_MASK) != (item & HINT_MASK)) continue;  // not the final push of this slot
                    f(e);
                }
            } else {
                for (uint32_t i = tid; i < H; i += BLOCK) {
                    const unsigned long long e = M::ld(&table[i]);
                    if (e != TBL_EMPTY) f(e);
                }
            }
        };

        // ---- emission ----
        if (!s.ovf) {
            for_each_entry([&](unsigned long long e) {  // pass 1: per-source counts
                const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                if constexpr (COUNT) {
                    const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + node);
                    const uint4 lo = rp[0];
                    const uint4 hi = rp[1];
                    const uint32_t flags = (hi.z >> 8) & 0xFFu;
                    // the units of the pruned search: every entry is a settled node (its distance is final and used), but an in-node
                    // with nothing beyond it within the bound is not EXPANDED -- the enumeration level records it from its parent's
                    // block and never gathers its own, so its out-edges are not relaxed (here it has an entry because this kernel
                    // emits a node from its own entry). lb+(node) = the smallest "what the child is needed for" of its own slots.
                    bool gathered = true;
                    if (a.prune && !(flags & F_EXT) && node != s.srcnode[src]) {
                        const uint32_t ws[4] = {hi.x & 0xFFFFu, hi.x >> 16, hi.y & 0xFFFFu, hi.y >> 16};
                        uint32_t beyond = 255u;
                        for (uint32_t j = 0; j < (hi.z & 0xFFu); j++) beyond = min(beyond, ((hi.z >> (16 + j)) & 1u) ? (ws[j] & 0xFFu) : (ws[j] >> 8));
                        gathered = (uint32_t)((e >> 1) & ENT_DIST_MASK) + beyond <= a.K1;
                    }
                    atomicAdd(&s.bt_settled, 1ull);
                    if (gathered) atomicAdd(&s.bt_relaxed, (unsigned long long)((flags & F_EXT) ? lo.z : (hi.z & 0xFFu)));
                }
                if (!(e & 1ull) && node != s.srcnode[src]) atomicAdd(&s.cnt[src], 1u);
            });
            __syncthreads();
            if (tid < 64) {  // exclusive scan of cnt[0..nsrc) by the first wave, then pool space for the batch
                uint32_t running = 0;
                for (int base = 0; base < BSRC; base += 64) {
                    const int i = base + tid;
                    const uint32_t v = i < nsrc ? s.cnt[i] : 0;
                    uint32_t incl = v;
                    for (int dd = 1; dd < 64; dd <<= 1) {
                        const uint32_t t = __shfl_up(incl, dd);
                        if (tid >= dd) incl += t;
                    }
                    if (i < BSRC) s.off[i] = running + incl - v;
                    running += __shfl(incl, 63);
                }
                if (tid == 0) {
                    s.total = running;
                    // staging too small -> larger level. The ranking below is quadratic in a source's candidates: fine up to the
                    // few thousand the LDS levels can stage, not for the millions the global-workspace level could -- beyond
                    // 32768 (10^9 comparisons, a few ms) that level hands the source to the dense level, which sorts (n log n).
                    if (running > (uint32_t)SCAP || (GLOBAL_WS && running > 32768u)) s.ovf = 1;
                    else if constexpr (!COUNT) {
                        if (s.chunk_next + running > s.chunk_end) {
                            const unsigned long long grab = running > POOL_CHUNK ? (unsigned long long)running : POOL_CHUNK;
                            s.chunk_next = atomicAdd(&a.counters[C_POOL], grab);
                            s.chunk_end = s.chunk_next + grab;
                        }
                        s.base = s.chunk_next;
                        s.chunk_next += running;
                    } else s.base = 0;
                }
            }
            __syncthreads();
        }
        if (!s.ovf) {
            if constexpr (!COUNT) {
                for_each_entry([&](unsigned long long e) {  // pass 2: stage keys per source segment (unsorted)
                    if (e & 1ull) return;
                    const uint32_t node = (uint32_t)(e >> ENT_NODE_SHIFT);
                    const uint32_t src = (uint32_t)(e >> ENT_SRC_SHIFT);
                    if (node == s.srcnode[src]) return;
                    const uint32_t pos = s.off[src] + atomicAdd(&s.fill[src], 1u);
                    M::st(&stage[pos], (((e >> 1) & ENT_DIST_MASK) << 32) | (unsigned long long)node);
                    M::st(&stage_src[pos], (uint16_t)src);
                });
                __syncthreads();
                // pass 3: rank within the source's segment (= sort by (distance, node)), write contiguous
                const uint32_t total = s.total;
                const unsigned long long base = s.base;
                for (uint32_t i = tid; i < total; i += BLOCK) {
                    const uint32_t src = M::ld(&stage_src[i]);
                    const unsigned long long key = M::ld(&stage[i]);
                    const uint32_t lo_ = s.off[src], hi_ = lo_ + s.cnt[src];
                    uint32_t rank = 0;
                    for (uint32_t j = lo_; j < hi_; j++) rank += mtg_policy_pops_before(M::ld(&stage[j]), key) ? 1u : 0u;  // policy P1 (mtg_policy.h)
                    const unsigned long long dst = base + lo_ + rank;
                    if (dst < a.pool_cap) a.pool[dst] = key;
                }
            }
            if (tid < nsrc) {
                const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
                const uint64_t o = abs_idx - a.src_begin;
                a.cand_start[o] = s.base + s.off[tid];
                a.cand_count[o] = s.cnt[tid];
            }
            if constexpr (COUNT) {
                if (tid == 0) {
                    s.st_settled += s.bt_settled; s.st_relaxed += s.bt_relaxed; s.st_attempts += s.bt_attempts;
                    s.st_emitted += s.total;
                    s.st_pushes += n_log;
                    if (n_log > s.st_max_log) s.st_max_log = n_log;
                    if (s.bt_settled > s.st_max_ent) s.st_max_ent = s.bt_settled;
                }
            }
        } else {
            if (tid == 0) s.base = atomicAdd(&a.counters[C_OVERFLOW], (unsigned long long)nsrc);  // slot in the overflow list
            __syncthreads();
            if (tid < nsrc) {
                const uint64_t abs_idx = a.src_index ? a.src_index[item0 + tid] : a.src_begin + item0 + tid;
                a.cand_count[abs_idx - a.src_begin] = CAND_OVERFLOW;
                a.ovf_list[s.base + tid] = (uint32_t)abs_idx;
            }
        }
        __syncthreads();

        // ---- clean the table for the next batch ----
        if (LOG_EMIT && !s.ovf) {
            for (uint32_t i = tid; i < n_log; i += BLOCK) M::st(&table[M::ld(&log[i]) >> HINT_BITS], TBL_EMPTY);
        } else {  // overflowed batches may have table entries that never reached the log
            for (uint32_t i = tid; i < H; i += BLOCK) M::st(&table[i], TBL_EMPTY);
        }
        __syncthreads();
    }

    if constexpr (COUNT) {
        if (tid == 0) {
            atomicAdd(&a.counters[C_SETTLED], s.st_settled);
            atomicAdd(&a.counters[C_RELAXED], s.st_relaxed);
            atomicAdd(&a.counters[C_ATTEMPTS], s.st_attempts);
            atomicAdd(&a.counters[C_EMITTED], s.st_emitted);
            atomicAdd(&a.counters[C_PUSHES], s.st_pushes);
            atomicMax(&a.counters[C_MAX_LOG], s.st_max_log);
            atomicMax(&a.counters[C_MAX_ENT], s.st_max_ent);
        }
    }
}

// Wave-level helpers shared by the lane-per-source level

// Overflowed sources of a wave are buffered in LDS and appended to the global overflow list 64 at a time, so the
// list cursor sees one atomic per 64 sources instead of one per overflow event.
struct WaveOvfBuf {
    uint32_t buf[128];
};
__device__ __forceinline__ void wave_ovf_push(WaveOvfBuf &w, uint32_t &nbuf, bool ovf, uint32_t abs_idx, const SsspArgs &a, int lane) {
    const unsigned long long m = __ballot(ovf);
    if (!m) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (ovf) w.buf[nbuf + rank] = abs_idx;
    nbuf += (uint32_t)__popcll(m);
    if (nbuf >= 64) {
        unsigned long long p0 = 0;
        if (lane == 0) p0 = atomicAdd(&a.counters[C_OVERFLOW], 64ull);
        p0 = __shfl(p0, 0);
        a.ovf_list[p0 + lane] = w.buf[lane];
        const uint32_t rest = w.buf[64 + lane];
        w.buf[lane] = rest;
        nbuf -= 64;
    }
}
__device__ __forceinline__ void wave_ovf_flush(WaveOvfBuf &w, uint32_t nbuf, const SsspArgs &a, int lane) {
    if (!nbuf) return;
    unsigned long long p0 = 0;
    if (lane == 0) p0 = atomicAdd(&a.counters[C_OVERFLOW], (unsigned long long)nbuf);
    p0 = __shfl(p0, 0);
    if ((uint32_t)lane < nbuf) a.ovf_list[p0 + lane] = w.buf[lane];
}

// ------------------------------------------------------------------------------------------------
// Lane-per-source kernel WITHOUT a table (level 0 of the default plan)
//
// A (k-1)-ball of a unitig graph is almost a tree (bench graph: 1.0004 path enumerations per settled node), so the search needs
// no visited set at all: every lane enumerates the bounded PATHS from its source depth-first with a small private stack in LDS --
// gather the node's 64-byte family block (the node, its children's in-node flags and the embedded children's out-edges), remember
// the in-nodes among the node and its children, push the grandchildren (and not-embedded children) whose distance stays <= k-1,
// pop the next one. No select-min, no find, no insert. A node reached along two paths is expanded twice (bounded: every edge
// weighs >= 1 and the path length is capped at k-1), its target hits are de-duplicated (minimum distance) and sorted by
// (distance, node) -- lists of up to four in registers when the source finishes, longer ones by the post-pass below --, which
// makes the output identical to the Dijkstra order. A source whose enumeration exceeds the step budget or its LDS space is handed
// to the cooperative cascade, which is exact for any ball.
//
// What bounds it: the version at the end of round 2's first session was bound by instruction issue (460 VALU +
// 366 SALU per wave step, 188 VGPRs, 8 waves per CU). Written branch-free -- every potential push / hit is an UNCONDITIONAL LDS store to the
// lane's next free slot (a store that does not count leaves the counter where it was), the next node always comes off the stack,
// sources are handed out by two cross-lane permutes from chunks held in registers -- a step is half the instructions and the
// kernel becomes bound by the latency of the gathers, i.e. by the number of waves per CU, i.e. by LDS. Hence:
//  * per lane only S1 stack and H1 hit slots; whatever a search needs beyond them lives in an EXTENSION BLOCK of 16 entries
//    (stack from the bottom, hits from the top) taken from a per-wave pool. The pool's free mask is one wave-uniform 64-bit
//    value: allocation and release are a few scalar instructions, no atomics;
//  * nothing is staged: a finished source takes its pool space with one LDS atomic and writes its keys, (start, count) and, if
//    needed, its post-pass work-list entry straight to memory.
// 10 KB of LDS per wave -> 16 waves per CU, where the SIMDs are busy again (4 waves x 27 % active each): the level now sits between
// instruction issue and its gather ceiling (78 % of it at 2^27).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t ENUM_POP_BUDGET = 256;
constexpr uint32_t ENUM_POOL_CHUNK = 2048;  // keys per wave-local pool chunk (one global atomic per chunk)
constexpr uint32_t ENUM_FIX_CHUNK = 512;    // post-pass work-list slots a wave takes per global atomic (unused ones hold FIX_NONE)
constexpr uint32_t FIX_CLASS_TAG = 0xFFFFFF00u;  // slot 0 of a work-list chunk: FIX_CLASS_TAG | class (0: <= 8 keys, 1: <= 16, 2: <= 32)
constexpr uint32_t FIX_NONE = 0xFFFFFFFFu;
#ifndef MTG_ENUM_S1
#define MTG_ENUM_S1 4
#define MTG_ENUM_H1 4
#define MTG_ENUM_NB 40
#define MTG_ENUM_BE 16
#endif
constexpr int ENUM_BE = MTG_ENUM_BE;        // entries per extension block
// LDS words between the starts of two extension blocks: one more than a block holds. With a stride of 16 eight-byte entries
// (128 bytes = all 32 banks once) the same row of every block falls on the same bank pair, and lanes that work in different
// blocks -- the common case -- collide on every access (3.6 conflict cycles per LDS instruction in the round-2 PMC pass); an odd
// stride walks the rows of consecutive blocks through the banks.
#ifndef MTG_ENUM_BS
#define MTG_ENUM_BS (MTG_ENUM_BE + 1)
#endif
constexpr int ENUM_BS = MTG_ENUM_BS;

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {  // value of the first lane, known uniform to the compiler
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

#ifdef MTG_ENUM_STATS
__device__ unsigned long long g_enum_prof[16384][3];  // development build: per wave [start, end (100-MHz ticks), steps]
#endif
#ifndef MTG_ENUM_WAVES_PER_SIMD
#define MTG_ENUM_WAVES_PER_SIMD 1  // (experiments: 5 makes the compiler keep the registers within a fifth wave per SIMD)
#endif
// W8: the blocks are in the 8:8 format (k <= 255). PRUNE (needs W8): successors are tested against distance + weight + lower bound,
// and the sources come from the launch's list of sources that can reach an in-node at all (act_index / act_node, counters[C_ACTIVE]).
template <int WPB, int S1, int H1, int NB, bool QUAD, bool W8, bool PRUNE>
__global__ __launch_bounds__(WPB * 64, MTG_ENUM_WAVES_PER_SIMD) void sssp_enum_kernel(SsspArgs a) {
    static_assert(W8 || !PRUNE, "the lower bounds live in the 8:8 format");
    static_assert(NB >= 1 && NB <= 64 && H1 >= 2, "pool free mask is one 64-bit word; lists of two are sorted from the first tier");
    constexpr int BE = ENUM_BE, BS = ENUM_BS;
    static_assert(BS >= BE, "block stride below the block size");
    constexpr uint32_t T1 = (uint32_t)(S1 + H1) * 64u;     // words of the per-lane tiers
    constexpr uint32_t SCRATCH = T1 + (uint32_t)NB * BS;   // block that absorbs the stores of lanes without a block of their own
    constexpr uint32_t IDLE_DIST = 0xFFFF0000u;            // distance of a lane without a source: nothing is within the bound from there
    // stack entry: node | (distance | own-flag-done << 16) << 32; hit entry = candidate key: node | distance << 32
    __shared__ unsigned long long s_mem[WPB][T1 + (NB + 1) * BS];
    __shared__ uint32_t s_cnt[WPB];
    __shared__ WaveOvfBuf s_ovf[WPB];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t K1 = a.K1;
    unsigned long long *const mem = s_mem[wv];

    unsigned long long free_mask = NB == 64 ? ~0ull : ((1ull << (NB & 63)) - 1ull);  // wave-uniform: free extension blocks
    unsigned long long pool_base = 0;                                                // wave-uniform
    unsigned long long fix_next[3] = {0, 0, 0}, fix_end[3] = {0, 0, 0};             // wave-uniform: the wave's open work-list chunk per length class
    uint32_t fix_total[3] = {0, 0, 0};                                              // wave-uniform: entries appended per class
    uint32_t n_overflow = 0;                                                         // wave-uniform
    if (lane == 0) s_cnt[wv] = ENUM_POOL_CHUNK;  // position inside the wave's pool chunk (no chunk yet)
#ifdef MTG_ENUM_STATS
    uint32_t st_starved = 0, st_full = 0, st_budget = 0, st_steps = 0, st_lanes = 0;
    const unsigned long long st_t0 = wall_clock64();
#endif

    // ---- sources: chunks of 64 in a STATIC stride (chunk c belongs to wave c mod n_waves), held in registers: `ids` is the chunk
    // being handed out, `ahead` the wave's next one (loaded a whole chunk before it is needed). A lane that needs a source gets
    // the next unused one by a cross-lane permute: no atomic, no memory round trip, no LDS. ----
    const unsigned long long n_items = PRUNE ? a.counters[C_ACTIVE] : a.n_items;
    const unsigned long long act_first = PRUNE ? a.counters[C_ACT_BEGIN] : 0ull;
    const unsigned long long n_waves = (unsigned long long)gridDim.x * WPB;
    unsigned long long next_chunk = (unsigned long long)blockIdx.x * WPB + wv;  // chunk that `ahead` will hold (wave-uniform)
    uint32_t cur_base = 0, cur_len = 0, cur_pos = 0, nxt_base = 0, nxt_len = 0;  // wave-uniform
    uint32_t ids = 0, ahead = 0;
    uint32_t ids_item = 0, ahead_item = 0;  // PRUNE: the sources' indices relative to src_begin (without a list they are cur_base + position)
    // (measured and dropped: the last quarter of the chunks handed out by a global counter, requested two chunks ahead so that the
    // atomic is never waited for -- 7 % slower at 2^27 and no better lane utilisation at 2^24)
    auto prefetch_chunk = [&]() {
        const unsigned long long lo = next_chunk * 64;
        nxt_base = (uint32_t)lo;
        nxt_len = lo >= n_items ? 0u : (n_items - lo < 64 ? (uint32_t)(n_items - lo) : 64u);
        if constexpr (PRUNE) {
            ahead = (uint32_t)lane < nxt_len ? a.act_node[act_first + lo + lane] : 0u;
            ahead_item = (uint32_t)lane < nxt_len ? (uint32_t)(a.act_index[act_first + lo + lane] - a.src_begin) : 0u;
        } else {
            ahead = (uint32_t)lane < nxt_len ? a.sources[a.src_begin + lo + lane] : 0u;
        }
        next_chunk += n_waves;
    };
    prefetch_chunk();
    ids = ahead; ids_item = ahead_item; cur_base = nxt_base; cur_len = nxt_len;
    prefetch_chunk();
    bool exhausted = cur_len == 0;
    auto take_source = [&](bool want, uint32_t &new_item, uint32_t &new_src) -> bool {
        if (exhausted) return false;
        const unsigned long long need = __ballot(want);
        const uint32_t idx = cur_pos + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        const bool second = idx >= 64u;  // (only a full chunk has a successor: cur_len < 64 means nxt_len == 0)
        const bool got = want && (second ? idx - 64u < nxt_len : idx < cur_len);
        const int sel = (int)((idx & 63u) << 2);
        const uint32_t from_cur = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ids), from_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ahead);
        new_src = second ? from_nxt : from_cur;
        if constexpr (PRUNE) {
            const uint32_t item_cur = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ids_item), item_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute(sel, (int)ahead_item);
            new_item = second ? item_nxt : item_cur;
        } else {
            new_item = second ? nxt_base + idx - 64u : cur_base + idx;
        }
        cur_pos += (uint32_t)__popcll(need);
        if (cur_pos >= 64u) {
            cur_pos -= 64u;
            ids = ahead; ids_item = ahead_item; cur_base = nxt_base; cur_len = nxt_len;
            prefetch_chunk();  // in flight while the chunk that just became current is handed out
        }
        exhausted = cur_pos >= cur_len;
        return got;
    };

    // ---- per-lane search state ----
    bool active = false;
    uint32_t sp = 0, nhit = 0, pops = 0, src_node = 0, item = 0;
    uint32_t blk = SCRATCH;                       // word index of the lane's extension block
    uint32_t cur_node = 0, cur_dist = IDLE_DIST;  // the node whose block is in b0..b3
    bool cur_chk = false;                         // its own in-node flag was already evaluated from its parent's block
    // The gather. QUAD = false: every lane loads the four quarters of its own block (four requests per lane to the same line).
    // QUAD = true: the four lanes of a quad load one block together -- in load l lane q fetches quarter q of the block of quad
    // lane l -- and transpose the quad's 4 x 4 quarters in registers (DPP) when the data is needed. An instruction then touches
    // 16 lines instead of 64. Measured (tools/gather_bench_tlb.hip): with a 5.7-GB table (the 2^27 graph) dependent random
    // 64-byte gathers run at 18.8 G/s with four requests per lane and at 44 G/s either way of making it one request per lane
    // and line -- beyond ~4 GB every lane-request pays an address translation; below 3 GB both forms reach 51-55 G/s.
    uint4 g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0}, g2 = {0, 0, 0, 0}, g3 = {0, 0, 0, 0};  // as loaded
    uint4 b0 = g0, b1 = g0, b2 = g0, b3 = g0;                                          // block of cur_node
    constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
    auto quad_bcast = [](uint32_t v, auto sel) -> uint32_t {  // value of lane `sel` of the quad
        constexpr int L = decltype(sel)::value;
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, L * 0x55, 0xF, 0xF, false);
    };
    auto load_block = [&](bool need, uint32_t node) {
        if constexpr (!QUAD) {
            if (need) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(a.recs + node);
                g0 = rp[0]; g1 = rp[1]; g2 = rp[2]; g3 = rp[3];
            }
        } else {
            const uint32_t want = need ? node : NO_NODE;
            const uint32_t q = (uint32_t)lane & 3u;
            const uint32_t n0 = quad_bcast(want, std::integral_constant<int, 0>{}), n1 = quad_bcast(want, std::integral_constant<int, 1>{});
            const uint32_t n2 = quad_bcast(want, std::integral_constant<int, 2>{}), n3 = quad_bcast(want, std::integral_constant<int, 3>{});
            if (n0 != NO_NODE) g0 = reinterpret_cast<const uint4 *>(a.recs + n0)[q];
            if (n1 != NO_NODE) g1 = reinterpret_cast<const uint4 *>(a.recs + n1)[q];
            if (n2 != NO_NODE) g2 = reinterpret_cast<const uint4 *>(a.recs + n2)[q];
            if (n3 != NO_NODE) g3 = reinterpret_cast<const uint4 *>(a.recs + n3)[q];
        }
    };
    auto arrive_block = [&]() {  // g -> b (QUAD: transpose of the quad's quarters, two exchange stages)
        if constexpr (!QUAD) {
            b0 = g0; b1 = g1; b2 = g2; b3 = g3;
        } else {
            // two exchange stages (lane ^ 1, lane ^ 2); in a stage even lanes take hi <- partner's lo, odd lanes lo <- partner's hi.
            // (v_cndmask_b32_dpp would do select and exchange in one instruction, but the compiler emits mov_dpp + cndmask for the
            // builtin, and two inline-asm forms -- one block per word pair, one per stage and register quadruple, the latter with the
            // block ADDRESS handed round the quad instead of the node id -- measured 3 % slower: fewer instructions, more stalls.)
            b0 = g0; b1 = g1; b2 = g2; b3 = g3;
            const bool odd1 = (lane & 1) != 0, odd2 = (lane & 2) != 0;
            auto xchg = [](uint32_t &lo, uint32_t &hi, bool odd, auto ctrl) {
                constexpr int C = decltype(ctrl)::value;
                const uint32_t from_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, C, 0xF, 0xF, false);
                const uint32_t from_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, C, 0xF, 0xF, false);
                lo = odd ? from_hi : lo;
                hi = odd ? hi : from_lo;
            };
            auto xchg4 = [&](uint4 &lo, uint4 &hi, bool odd, auto ctrl) {
                xchg(lo.x, hi.x, odd, ctrl); xchg(lo.y, hi.y, odd, ctrl); xchg(lo.z, hi.z, odd, ctrl); xchg(lo.w, hi.w, odd, ctrl);
            };
            xchg4(b0, b1, odd1, std::integral_constant<int, 0xB1>{});  // quad_perm [1,0,3,2]
            xchg4(b2, b3, odd1, std::integral_constant<int, 0xB1>{});
            xchg4(b0, b2, odd2, std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
            xchg4(b1, b3, odd2, std::integral_constant<int, 0x4E>{});
        }
    };
    // LDS word of stack / hit row `row` of this lane: first tier (slot-major / lane-minor: conflict free), then the extension block
    // at `base` (stack from its bottom, hits from its top). Rows are in range by construction: a lane whose step would not fit
    // is redirected to the scratch block BEFORE it stores anything.
    auto stack_word = [&](uint32_t base, uint32_t row) -> uint32_t {
        return row < (uint32_t)S1 ? row * 64u + (uint32_t)lane : base - (uint32_t)S1 + row;
    };
    auto hit_word = [&](uint32_t base, uint32_t row) -> uint32_t {
        return row < (uint32_t)H1 ? (uint32_t)(S1 * 64) + row * 64u + (uint32_t)lane : base + (uint32_t)(BE - 1 + H1) - row;
    };
    {
        uint32_t ni = 0, ns = 0;
        if (take_source(true, ni, ns)) {
            item = ni; src_node = ns; cur_node = ns; cur_dist = 0;
            active = true;
        }
        load_block(active, cur_node);
    }
    while (__any(active)) {
        // (the quad form runs the first half of a step -- transpose, decode, stores, pop, until its gather has left -- at a raised wave
        // priority: 2.5 % faster at 2^27; the per-lane form measured 10 % slower with it at 2^24)
        if constexpr (QUAD) __builtin_amdgcn_s_setprio(2);
        arrive_block();
        // ---- the block that arrived: which hits and successors count ----
        const uint32_t u = cur_node, d = cur_dist;
        const uint32_t meta = b1.z;
        const uint32_t nb[4] = {b0.x, b0.y, b0.z, b0.w};
        const uint32_t gn[GSLOTS] = {b1.w, b2.x, b2.y, b2.z, b2.w, b3.x};
        // distance of a child / grandchild, and (PRUNE) the smallest distance of any in-node through it
        uint32_t dc[4], dg[GSLOTS], lc[4], lg[GSLOTS];
        {
            const uint32_t cs[4] = {b1.x & 0xFFFFu, b1.x >> 16, b1.y & 0xFFFFu, b1.y >> 16};
            const uint32_t gs[GSLOTS] = {b3.y & 0xFFFFu, b3.y >> 16, b3.z & 0xFFFFu, b3.z >> 16, b3.w & 0xFFFFu, b3.w >> 16};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                dc[j] = d + (W8 ? (cs[j] & 0xFFu) : cs[j]);
                lc[j] = PRUNE ? d + (cs[j] >> 8) : dc[j];
            }
#pragma unroll
            for (int t = 0; t < GSLOTS; t++) {
                dg[t] = d + (W8 ? (gs[t] & 0xFFu) : gs[t]);
                lg[t] = PRUNE ? d + (gs[t] >> 8) : dg[t];
            }
        }
        const bool is_ext = active && (meta & ((uint32_t)F_EXT << 8));
        // PRUNE: a grandchild that is an in-node (cmeta bit 8 + t) is recorded from THIS block, like the children, and pushed -- with its
        // own flag done -- only if something lies beyond it (the high bytes carry weight + lb+): a quarter of the node visits of the
        // bench graph are in-nodes with nothing behind them within the bound, and their blocks are never gathered.
        bool hv[5], hg[GSLOTS], pv[4 + GSLOTS];
        hv[0] = active && !cur_chk && (meta & ((uint32_t)F_TARGET << 8)) && u != src_node;  // forbid_source_target, greedytigs/mod.rs:329
        uint32_t nh_f = nhit + (hv[0] ? 1u : 0u), sp_f = sp;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            hv[1 + j] = dc[j] <= K1 && (meta & (0x10000u << j)) && nb[j] != src_node;
            pv[j] = lc[j] <= K1 && (meta & (0x100000u << j));
            nh_f += hv[1 + j] ? 1u : 0u;
            sp_f += pv[j] ? 1u : 0u;
        }
#pragma unroll
        for (int t = 0; t < GSLOTS; t++) {
            pv[4 + t] = lg[t] <= K1;
            sp_f += pv[4 + t] ? 1u : 0u;
            hg[t] = PRUNE && dg[t] <= K1 && (meta & (0x1000000u << t)) && gn[t] != src_node;
            nh_f += hg[t] ? 1u : 0u;
        }
        if (__any(is_ext)) {  // spilled adjacency (more than 4 out-edges; never in a de Bruijn graph): count first
            if (is_ext) {
                const uint64_t ext_begin = ((uint64_t)b0.y << 32) | b0.x;
                for (uint32_t j = 0; j < b0.z; j++) sp_f += d + a.ext_w[ext_begin + j] <= K1 ? 1u : 0u;
            }
        }
        // ---- extension blocks for the lanes that outgrow their first tier in this step ----
        const bool need_blk = blk == SCRATCH && (sp_f > (uint32_t)S1 || nh_f > (uint32_t)H1);
        unsigned long long nm = __ballot(need_blk);
        while (nm && free_mask) {
            const int l = __builtin_ctzll(nm);
            const int bi = __builtin_ctzll(free_mask);
            nm &= nm - 1;
            free_mask &= free_mask - 1;
            blk = lane == l ? T1 + (uint32_t)bi * BS : blk;
        }
        const uint32_t es = sp_f > (uint32_t)S1 ? sp_f - (uint32_t)S1 : 0u, eh = nh_f > (uint32_t)H1 ? nh_f - (uint32_t)H1 : 0u;
        pops += active ? 1u : 0u;
        // (a lane the pool had no block for, or whose block is full, hands its source to the cascade)
        // (one word stays free: the store after the last one that counts must not land on the other side's newest entry)
        const bool ovf = active && ((need_blk && blk == SCRATCH) || es + eh >= (uint32_t)BE || pops > ENUM_POP_BUDGET);
#ifdef MTG_ENUM_STATS  // development build: why sources leave this level, how many steps run, how full the lanes are
        st_starved += (uint32_t)__popcll(__ballot(active && need_blk && blk == SCRATCH));
        st_full += (uint32_t)__popcll(__ballot(active && !(need_blk && blk == SCRATCH) && es + eh >= (uint32_t)BE));
        st_budget += (uint32_t)__popcll(__ballot(active && pops > ENUM_POP_BUDGET));
        st_steps += 1;
        st_lanes += (uint32_t)__popcll(__ballot(active));
#endif

        // ---- hits and successors, straight into LDS (an overflowing lane scribbles into the scratch block instead) ----
        const uint32_t base = ovf ? SCRATCH : blk;
        uint32_t wsp = ovf ? 0u : sp, wnh = ovf ? 0u : nhit;
        if (!PRUNE || __any(hv[0])) {  // (PRUNE: only the successors of a spilled adjacency arrive with their own flag still open)
            mem[hit_word(base, wnh)] = ((unsigned long long)d << 32) | u;
            wnh += hv[0] ? 1u : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j < 2 || __any(hv[1 + j])) {
                mem[hit_word(base, wnh)] = ((unsigned long long)dc[j] << 32) | nb[j];
                wnh += hv[1 + j] ? 1u : 0u;
            }
        }
        if constexpr (PRUNE) {
#pragma unroll
            for (int t = 0; t < GSLOTS; t++) {
                if (__any(hg[t])) {
                    mem[hit_word(base, wnh)] = ((unsigned long long)dg[t] << 32) | gn[t];
                    wnh += hg[t] ? 1u : 0u;
                }
            }
        }
        if (__any(is_ext)) {
            if (is_ext && !ovf) {
                const uint64_t ext_begin = ((uint64_t)b0.y << 32) | b0.x;
                for (uint32_t j = 0; j < b0.z; j++) {
                    const uint32_t nd = d + a.ext_w[ext_begin + j];
                    if (nd <= K1) { mem[stack_word(base, wsp)] = ((unsigned long long)nd << 32) | a.ext_col[ext_begin + j]; wsp++; }
                }
            }
        }
#pragma unroll
        for (int t = GSLOTS - 1; t >= 0; t--) {
            if (t < 4 || __any(pv[4 + t])) {
                mem[stack_word(base, wsp)] = ((unsigned long long)(dg[t] | (PRUNE ? 0x10000u : 0u)) << 32) | gn[t];  // (PRUNE: flag done above)
                wsp += pv[4 + t] ? 1u : 0u;
            }
        }
#pragma unroll
        for (int j = 3; j >= 0; j--) {  // (children come off the stack before grandchildren: nearer nodes first)
            if (j < 2 || __any(pv[j])) {
                mem[stack_word(base, wsp)] = ((unsigned long long)(dc[j] | 0x10000u) << 32) | nb[j];  // its in-node flag is done
                wsp += pv[j] ? 1u : 0u;
            }
        }
        sp = wsp; nhit = wnh;
        const bool go_on = active && !ovf && sp > 0;  // the next node comes off the stack
        const unsigned long long top = mem[stack_word(base, sp > 0 ? sp - 1u : 0u)];
        sp -= go_on ? 1u : 0u;
        const bool fin = active && !ovf && !go_on;  // the source is finished

        // ---- lanes without a next node take a new source; every lane's next gather leaves now ----
        uint32_t new_item = 0, new_src = 0;
        const bool got_new = take_source(!go_on, new_item, new_src);
        const uint32_t nx_node = go_on ? (uint32_t)top : new_src;
        load_block(go_on || got_new, nx_node);
        if constexpr (QUAD) __builtin_amdgcn_s_setprio(0);

        // ---- finished / overflowed sources write their result ----
        const unsigned long long donemask = __ballot(fin || ovf);
        if (donemask) {
            uint32_t c = fin ? nhit : 0u;
            // lists of up to four keys are put in Dijkstra order in registers; longer ones (and repeated nodes) go to the post-pass
            unsigned long long k0 = ~0ull, k1 = ~0ull, k2 = ~0ull, k3 = ~0ull;
            if (c > 0) k0 = mem[hit_word(blk, 0)];
            if (c > 1) k1 = mem[hit_word(blk, 1)];
            if (c > 2) k2 = mem[hit_word(blk, 2)];
            if (c > 3) k3 = mem[hit_word(blk, 3)];
            bool fix = c > 4;
            if (__any(c >= 2 && c <= 4)) {  // (a longer list is sorted as a whole by the post-pass: no need to run the network for it alone)
                auto cswap = [](unsigned long long &x, unsigned long long &y) {  // pop order: policy P1 (mtg_policy.h)
                    const bool xf = mtg_policy_pops_before(x, y);
                    const unsigned long long lo = xf ? x : y, hi = xf ? y : x;
                    x = lo; y = hi;
                };
                cswap(k0, k1); cswap(k2, k3); cswap(k0, k2); cswap(k1, k3); cswap(k1, k2);  // (longer lists are sorted again anyway)
                const uint32_t n0 = (uint32_t)k0, n1 = (uint32_t)k1, n2 = (uint32_t)k2, n3 = (uint32_t)k3;
                const bool dup = (c >= 2 && c <= 4) && (n0 == n1 || (c > 2 && (n0 == n2 || n1 == n2)) || (c > 3 && (n0 == n3 || n1 == n3 || n2 == n3)));
                if (dup && c == 2) c = 1;  // the same node along two paths: the shorter distance stays
                else fix |= dup;
            }
            uint32_t off = c ? atomicAdd(&s_cnt[wv], c) : 0u;  // any order: (start, count) index the content
            if (__builtin_amdgcn_readfirstlane(s_cnt[wv]) > ENUM_POOL_CHUNK) {  // chunk used up: this step's lists go to a new one
                unsigned long long p0 = 0;
                if (lane == 0) {
                    p0 = atomicAdd(&a.counters[C_POOL], (unsigned long long)ENUM_POOL_CHUNK);
                    s_cnt[wv] = 0;
                }
                pool_base = uniform_u64(p0);
                off = c ? atomicAdd(&s_cnt[wv], c) : 0u;
            }
            const unsigned long long pos = pool_base + off;
            const bool room = pos + c <= a.pool_cap;  // (pool too small: the host retries with a larger one)
            if (c > 0 && room) a.pool[pos] = k0;
            if (c > 1 && room) a.pool[pos + 1] = k1;
            if (c > 2 && room) a.pool[pos + 2] = k2;
            if (c > 3 && room) a.pool[pos + 3] = k3;
            for (uint32_t r = 4; __any(r < c); r++)
                if (r < c && room) a.pool[pos + r] = mem[hit_word(blk, r)];
            // (7 of 10 sources have no candidate: their counts are zeroed by one streaming pass before the launch, their starts are
            // never read -- two scattered partial-line stores less per such source: 1.85 -> 1.40 -> ... GB written per launch at 2^27)
            if (fin && c) {
                a.cand_start[item] = pos;
                a.cand_count[item] = c;
            } else if (ovf) a.cand_count[item] = CAND_OVERFLOW;
            // post-pass work list: a wave fills one chunk per length class at a time (a wave of the post-pass then sorts lists of
            // similar length with a network of that size)
            auto append_fix = [&](bool f, int cls) {
                const unsigned long long fm = __ballot(f);
                if (!fm) return;
                const uint32_t nf = (uint32_t)__popcll(fm);
                if (fix_next[cls] + nf > fix_end[cls]) {  // the rest of the old chunk is marked unused
                    for (unsigned long long t = fix_next[cls] + lane; t < fix_end[cls]; t += 64) a.fix_list[t] = FIX_NONE;
                    unsigned long long f0 = 0;
                    if (lane == 0) {
                        f0 = atomicAdd(&a.counters[C_FIX], (unsigned long long)ENUM_FIX_CHUNK);
                        a.fix_list[f0] = FIX_CLASS_TAG | (uint32_t)cls;
                    }
                    fix_next[cls] = uniform_u64(f0) + 1;
                    fix_end[cls] = fix_next[cls] - 1 + ENUM_FIX_CHUNK;
                }
                if (f) a.fix_list[fix_next[cls] + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = item;
                fix_next[cls] += nf;
                fix_total[cls] += nf;
            };
            append_fix(fix && c <= 8, 0);
            append_fix(fix && c > 8 && c <= 16, 1);
            append_fix(fix && c > 16, 2);
            unsigned long long rel = __ballot((fin || ovf) && blk != SCRATCH);  // extension blocks go back to the pool
            while (rel) {
                const int l = __builtin_ctzll(rel);
                rel &= rel - 1;
                free_mask |= 1ull << ((((uint32_t)__builtin_amdgcn_readlane((int)blk, l) - T1) / (uint32_t)BS) & 63u);
            }
            wave_ovf_push(s_ovf[wv], n_overflow, ovf, (uint32_t)(a.src_begin + item), a, lane);
        }

        // ---- commit: every lane moves to its next block ----
        if (go_on) {
            cur_node = nx_node;
            cur_dist = (uint32_t)(top >> 32) & 0xFFFFu;
            cur_chk = ((uint32_t)(top >> 32) & 0x10000u) != 0u;
        } else {
            sp = 0; nhit = 0; pops = 0;
            blk = SCRATCH;
            cur_chk = false;
            active = got_new;
            item = new_item; src_node = new_src; cur_node = new_src;
            cur_dist = got_new ? 0u : IDLE_DIST;
        }
    }
    for (int cls = 0; cls < 3; cls++) {
        for (unsigned long long t = fix_next[cls] + lane; t < fix_end[cls]; t += 64) a.fix_list[t] = FIX_NONE;
        if (lane == 0 && fix_total[cls]) atomicAdd(&a.counters[C_FIX_CLASS0 + cls], (unsigned long long)fix_total[cls]);
    }
    wave_ovf_flush(s_ovf[wv], n_overflow, a, lane);
#ifdef MTG_ENUM_STATS
    if (lane == 0) {
        const uint32_t wid = blockIdx.x * WPB + wv;
        if (wid < 16384) { g_enum_prof[wid][0] = st_t0; g_enum_prof[wid][1] = wall_clock64(); g_enum_prof[wid][2] = st_steps; }
        atomicAdd(&a.counters[C_SETTLED], (unsigned long long)st_steps);
        atomicAdd(&a.counters[C_RELAXED], (unsigned long long)st_lanes);
        atomicAdd(&a.counters[C_EMITTED], (unsigned long long)st_starved);
        atomicAdd(&a.counters[C_ATTEMPTS], (unsigned long long)st_full);
        atomicAdd(&a.counters[C_PUSHES], (unsigned long long)st_budget);
    }
#endif
}

// Post-pass of the enumeration level: a source's hits arrive in discovery order and may name a node more than once (one
// hit per path). Keep the smallest distance per node and order by (distance, node) -- what Dijkstra's pop order gives
// (SURVEY App. A.1). The level itself puts lists of up to four keys in order; what it leaves here are the longer lists (5.8 % of
// the bench graph's sources have 5-8 candidates, 2.8 % 9-16, 0.4 % more: together 58 % of all keys) and the rare short list with a
// repeated node. What was measured on the way (2^27 bench graph): one thread per list with an insertion sort in LDS
// (round 2) 0.61 ms -- dependent LDS round trips and divergent loop control; one thread per list with the keys in registers and a
// sorting network 0.15 + 0.23 ms for the lists of <= 8 / <= 16 keys, bound by the number of memory requests (every lane reads and
// writes its own list), and a 32-slot network is 40 KB of straight-line code whose first pass runs at the latency of
// instruction-cache misses (0.26 ms for a handful of lists); the form below 0.13 + 0.12 + 0.03 ms.
// The level's work list is chunked (slot 0 of a chunk names its length class, unused slots hold FIX_NONE); this pass makes it dense:
// class 0 first, then class 1, then class 2 (the level counted the entries per class), so that the sorting kernels below run over
// plain ranges.
__global__ __launch_bounds__(256) void fix_compact_kernel(const uint32_t *fix_list, unsigned long long *counters, uint32_t *dense) {
    __shared__ uint32_t s_cnt[3], s_off[3];
    __shared__ unsigned long long s_base[3];
    const unsigned long long n_slots = counters[C_FIX];
    const unsigned long long off[3] = {0, counters[C_FIX_CLASS0], counters[C_FIX_CLASS0] + counters[C_FIX_CLASS0 + 1]};
    // a workgroup takes a contiguous, chunk-aligned slice of the slots: it counts its entries per class, reserves its dense ranges with
    // ONE global atomic per class (same-address atomics cost ~7 ns each: one per wave made this pass 0.64 ms at 2^27), then copies
    const unsigned long long n_chunks = (n_slots + ENUM_FIX_CHUNK - 1) / ENUM_FIX_CHUNK;
    const unsigned long long per_block = (n_chunks + gridDim.x - 1) / gridDim.x * ENUM_FIX_CHUNK;
    const unsigned long long lo = (unsigned long long)blockIdx.x * per_block, hi = lo + per_block < n_slots ? lo + per_block : n_slots;
    if (threadIdx.x < 3) { s_cnt[threadIdx.x] = 0; s_off[threadIdx.x] = 0; }
    __syncthreads();
    for (unsigned long long s0 = lo; s0 < hi; s0 += 256) {  // (a wave stays inside one chunk: one class)
        const unsigned long long sl = s0 + threadIdx.x;
        const uint32_t e = sl < hi && (sl % ENUM_FIX_CHUNK) != 0 ? fix_list[sl] : FIX_NONE;
        const unsigned long long m = __ballot(e != FIX_NONE);
        if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt[fix_list[sl / ENUM_FIX_CHUNK * ENUM_FIX_CHUNK] & 3u], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_cnt[threadIdx.x]) s_base[threadIdx.x] = off[threadIdx.x] + atomicAdd(&counters[C_FIX_CURSOR0 + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    for (unsigned long long s0 = lo; s0 < hi; s0 += 256) {
        const unsigned long long sl = s0 + threadIdx.x;
        const uint32_t e = sl < hi && (sl % ENUM_FIX_CHUNK) != 0 ? fix_list[sl] : FIX_NONE;
        const unsigned long long m = __ballot(e != FIX_NONE);
        if (!m) continue;
        const uint32_t cls = fix_list[sl / ENUM_FIX_CHUNK * ENUM_FIX_CHUNK] & 3u;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&s_off[cls], (uint32_t)__popcll(m));
        base = __shfl(base, 0);
        if (e != FIX_NONE) dense[s_base[cls] + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = e;
    }
}

// Lane-parallel form: G lanes per list, one key per lane -- coalesced reads and writes (a list is one or two memory requests instead
// of one per key or key pair), bitonic sort across the lanes, then each key looks at the keys before it for its node.
template <int G, int CLS, int BLOCK>
__device__ __forceinline__ void sort_lists_class(unsigned long long *pool, uint64_t pool_cap, const unsigned long long *cand_start,
                                                 uint32_t *cand_count, const uint32_t *dense, const unsigned long long *counters, uint32_t block,
                                                 uint32_t n_blocks) {
    constexpr uint32_t LPB = BLOCK / G;  // lists per workgroup pass
    unsigned long long begin = 0;
    for (int k = 0; k < CLS; k++) begin += counters[C_FIX_CLASS0 + k];
    const unsigned long long n = counters[C_FIX_CLASS0 + CLS];
    const uint32_t g = threadIdx.x % G;
    for (unsigned long long l0 = (unsigned long long)block * LPB; l0 < n; l0 += (unsigned long long)n_blocks * LPB) {
        const unsigned long long l = l0 + threadIdx.x / G;
        const uint32_t i = l < n ? dense[begin + l] : FIX_NONE;
        uint32_t c = i != FIX_NONE ? cand_count[i] : 0u;
        const unsigned long long st = i != FIX_NONE ? cand_start[i] : 0ull;
        if (c < 2 || c > (uint32_t)G || st + c > pool_cap) c = 0;  // (pool too small: the host retries with a larger one)
        unsigned long long key = g < c ? pool[st + g] : ~0ull;
#pragma unroll
        for (int k = 2; k <= G; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j);
                const bool take_min = ((g & (uint32_t)k) == 0) == ((g & (uint32_t)j) == 0);
                const bool kf = mtg_policy_pops_before(key, other);  // pop order: policy P1 (mtg_policy.h)
                const unsigned long long lo = kf ? key : other, hi = kf ? other : key;
                key = take_min ? lo : hi;
            }
        }
        const uint32_t node = (uint32_t)key;
        bool dup = false;  // a node named twice: only its first (= smallest distance) occurrence stays
#pragma unroll
        for (int dlt = 1; dlt < G; dlt++) {
            const uint32_t before = __shfl_up(node, dlt, G);
            dup |= g >= (uint32_t)dlt && g < c && before == node;
        }
        const unsigned long long dm = __ballot(dup);
        if (dm) {  // (rare)
            const uint32_t lane = threadIdx.x & 63u;
            const unsigned long long group = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << (lane - g);
            const uint32_t before = (uint32_t)__popcll(dm & group & ((1ull << lane) - 1ull));
            if (g < c && !dup) pool[st + g - before] = key;
            if (g == 0 && (dm & group)) cand_count[i] = c - (uint32_t)__popcll(dm & group);
        } else if (g < c) pool[st + g] = key;
    }
}
// the three length classes in ONE launch (a third of the grid each: their lists are disjoint, and three launches in a row spent more
// on their gaps and tails than on sorting)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void sort_lists_kernel(unsigned long long *pool, uint64_t pool_cap, const unsigned long long *cand_start,
                                                           uint32_t *cand_count, const uint32_t *dense, const unsigned long long *counters) {
    const uint32_t per = gridDim.x / 3u, cls = blockIdx.x / per, block = blockIdx.x % per;
    if (cls == 0) sort_lists_class<8, 0, BLOCK>(pool, pool_cap, cand_start, cand_count, dense, counters, block, per);
    else if (cls == 1) sort_lists_class<16, 1, BLOCK>(pool, pool_cap, cand_start, cand_count, dense, counters, block, per);
    else if (cls == 2) sort_lists_class<32, 2, BLOCK>(pool, pool_cap, cand_start, cand_count, dense, counters, block, per);
}

#include "replay_kernels.inc"

// ------------------------------------------------------------------------------------------------
// Host side of the device stage
// ------------------------------------------------------------------------------------------------
// Work arrays of the GPU claim replay (replay_kernels.inc); owned by the Device they were allocated on.

struct ReplayWork {
    uint64_t cap_v = 0, cap_s = 0, cap_spill = 0, cap_blocks = 0, cap_out = 0;
    unsigned long long *state = nullptr;      // [V] working node states {mirror, multiplicity, live}
    unsigned long long *resv[2] = {nullptr, nullptr};
    uint32_t tag_base = 0xFFFFFFFFu;          // reservation tags used so far (the arrays are never cleared between calls)
    Touch *touch = nullptr;
    uint32_t *src_mirror = nullptr;
    Dense *dense = nullptr;                   // the sources that have candidates (replay_kernels.inc)
    uint64_t cap_dense = 0;
    unsigned long long *claims = nullptr;
    uint32_t *pending[2] = {nullptr, nullptr}, *spill = nullptr;
    unsigned long long *final_off = nullptr, *block_sums = nullptr;
    unsigned long long *ctl = nullptr, *h_ctl = nullptr;  // control block (device / pinned host copy)
    mtg_pair *out = nullptr;
    mtg_pair *h_out = nullptr;                // pinned staging of the pair download (pageable D2H runs at a few GB/s)
    uint64_t cap_h_out = 0;
    unsigned grid = 0, grid_small = 0;        // co-resident workgroups of the cooperative launch (workgroups of 1024 / of 256)
};

struct Device {
    int dev = 0;
    uint64_t k = 0;
    uint32_t K1 = 0;
    uint64_t V = 0;
    uint64_t E0 = 0;  // original edges of the graph the device copy was built from
    double lower_bounds_ms = 0.0;  // GPU time of the lower-bound precompute (0 without)
    bool single_use = false;  // the caller searches once (mtg_compute_tigs_cfg): what only the search needs goes back before the claim replay takes its arrays
    NodeBlock *d_recs = nullptr;              // [V] family blocks
    uint32_t *d_odeg = nullptr;               // [V] out-degree (classification)
    uint8_t *d_cls = nullptr;                 // [V] class byte of the last classification (F_TARGET | F_SOURCE | F_SELF_MIRROR)
    uint32_t *d_ext_col = nullptr;
    uint16_t *d_ext_w = nullptr;
    uint64_t ext_n = 0;
    int32_t *d_mult = nullptr;
    uint32_t *d_out_nodes = nullptr;
    uint32_t *d_block_counts = nullptr;
    uint64_t n_cls_blocks = 0;
    uint64_t n_sources = 0;
    uint64_t total_demand = 0;  // sum of the positive multiplicities (classification)
    bool classified = false;
    unsigned long long *d_counters = nullptr;
    unsigned long long *h_counters = nullptr;  // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_r[4] = {nullptr, nullptr, nullptr, nullptr};  // claim replay: start, before / after the rounds kernel, end of the GPU work
    double last_replay_kernel_ms = 0.0, last_replay_gpu_ms = 0.0;
    double last_wall_s[3] = {0, 0, 0};  // host wall clock of the last device_pairs[_multi]: SSSP stage (+ gather), claim replay, pair download
    double last_kernel_ms = 0.0;  // sum of the SSSP level kernels' HIP-event durations of the last call
    uint32_t *d_mirror = nullptr;             // [V] mirror node (claim replay)
    uint32_t *d_ovf[2] = {nullptr, nullptr};  // ping-pong overflow source lists
    uint32_t *d_fix = nullptr, *d_fix_dense = nullptr;  // enumeration level: work list of its post-pass (chunked, as written / dense, by length class)
    uint64_t ovf_cap = 0;
    int last_n_levels = 0;           // per-level record of the last call: kernel ms and sources handed to the level
    double last_level_ms[8] = {0};
    uint64_t last_level_sources[8] = {0};
    std::string last_level_name[8];
    int plan = 0;  // 0 = table-free path enumeration per lane, then the cooperative cascade for the heaviest sources (the form of its gathers
                   // chosen by the size of the graph); 1 = cascade only; 2 / 3 = plan 0 with quad-cooperative / per-lane gathers regardless of size;
                   // + 4 = without the goal-directed pruning (full balls, every source searched: A/B runs and tests)
    bool w8 = false;                 // blocks in the 8:8 format with lower bounds (k <= 255)
    // the sources that can reach an in-node (8:8 format), in order, written by the classification (cap: act_cap sources); per-block
    // counts of them (d_act_blocks, n_cls_blocks words) and their number (d_act_total, on the device)
    uint32_t *d_act_index = nullptr, *d_act_node = nullptr, *d_act_blocks = nullptr;
    unsigned long long *d_act_total = nullptr;
    uint64_t act_cap = 0;
    uint64_t last_active_sources = 0;
    int n_cu = 256;
    uint64_t graph_bytes = 0;
    ReplayWork replay;
    // claim replay tuning (0 = the engine's choice; never changes a result: tests run the rounds under several settings)
    uint64_t tune_windows = 0;
    int tune_block = 0, tune_grid = 0, tune_role_mod = 0;
    bool tune_plain_barrier = false;  // every workgroup releases at the grid barrier (no per-XCD stage)
    int last_replay_rounds = 0;
    uint64_t last_n_pairs = 0;        // pairs of the last claim replay; they stay in replay.out until the next one (or device_take_pairs)
    uint64_t last_replay_visits = 0;  // sum over the rounds of the pending-list lengths (first RC_TRACE_ROUNDS rounds)
};

typedef void (*sssp_fn)(SsspArgs);
static void scan_u32(Device *d, hipStream_t st, ReplayWork &w, const uint32_t *in, uint64_t n, unsigned long long *out,
                     unsigned long long *d_total);

struct LevelCfg {
    sssp_fn fn;
    sssp_fn fn_count;
    int block;
    int bsrc;
    int logh;
    int qcap;
    int scap;
    bool global_ws;
    std::string name() const {
        char b[96];
        std::snprintf(b, sizeof b, "sssp_kernel<%d,%d,%d,%d,%d%s>", block, logh, qcap, scap, bsrc, global_ws ? ",global" : "");
        return b;
    }
};

template <int BLOCK, int LOGH, int QCAP, int SCAP, int BSRC, bool GLOBAL_WS>
static LevelCfg make_cfg() {
    return LevelCfg{sssp_kernel<BLOCK, LOGH, QCAP, SCAP, BSRC, false, GLOBAL_WS>,
                    sssp_kernel<BLOCK, LOGH, QCAP, SCAP, BSRC, true, GLOBAL_WS>, BLOCK, BSRC, LOGH, QCAP, SCAP, GLOBAL_WS};
}

constexpr int ENUM_WPB = 4, ENUM_S1 = MTG_ENUM_S1, ENUM_H1 = MTG_ENUM_H1, ENUM_NB = MTG_ENUM_NB;  // 40 KB of LDS per workgroup: 4 workgroups = 16 waves per CU
constexpr int ENUM_MAX_HITS = ENUM_H1 + ENUM_BE - 1;  // longest list the level can emit
static std::string enum_level_name(bool quad, bool prune) {
    char b[160];
    std::snprintf(b, sizeof b, "%ssssp_enum_kernel<%d,%d,%d,%d,%s%s> + fix_compact_kernel + sort_lists_kernel", prune ? "active_range_kernel + " : "",
                  ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, quad ? "quad" : "lane", prune ? ",pruned" : "");
    return b;
}

// Cascade of cooperative levels: 32 sources per workgroup, then 8, then 1 (128 KB LDS table), then a 32 MB
// global-memory table. A level re-runs the sources whose batch overflowed the previous level's tables.
static const int N_COOP_LEVELS = 5;
static LevelCfg coop_level(int i, bool after_enum) {
    switch (i) {  //                    BLOCK LOGH  QCAP  SCAP BSRC
        // 25 KB of LDS per workgroup -> 6 workgroups per CU: the level is bound by rounds x gather latency per batch
        // (4 us per round, ~10 rounds), so concurrency per CU is what counts; its heavy batches overflow the 1024-item
        // log early and are re-run by the next level
        case 0: return make_cfg<256, 11, 1024, 512, 32, false>();
        case 1:
            // what the enumeration level hands on are its ~1 % heaviest sources: 8 per workgroup in 29 KB of LDS (5 workgroups
            // per CU) measured 0.18 ms against 0.27 ms for 16 per workgroup in 58 KB
            if (after_enum) return make_cfg<256, 11, 2048, 512, 8, false>();
            return make_cfg<256, 12, 4096, 1024, 16, false>();
        case 2: return make_cfg<256, 13, 8192, 2048, 8, false>();
        case 3: return make_cfg<256, 14, 4096, 1024, 1, false>();
        default: return make_cfg<256, 22, 1 << 22, 1 << 21, 1, true>();
    }
}

static float elapsed_ms(Device *d) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, d->ev0, d->ev1));
    return ms;
}

static bool enum_uses_quad_gathers(const Device *d) {
    // beyond ~3 GB of family blocks every lane-request pays an address translation: quad-cooperative gathers (see the kernel)
    return (d->plan & 3) == 2 || ((d->plan & 3) == 0 && d->V * sizeof(NodeBlock) > (3ull << 30));
}
static bool enum_prunes(const Device *d) { return d->w8 && !(d->plan & 4) && (d->plan & 3) != 1; }  // (plan 1 = the plain cascade: full balls)

// the searched sources of [src_begin, src_begin + n): the part of the classification's list (d_act_index / d_act_node) inside the range --
// first entry and number in counters[C_ACT_BEGIN] / counters[C_ACTIVE]; zeroes cand_count (a search only stores the count of a non-empty list)
static void launch_active_list(Device *d, hipStream_t st, const SsspArgs &args, uint64_t n) {
    hipLaunchKernelGGL(active_range_kernel, dim3(1), dim3(64), 0, st, d->d_act_index, d->d_act_total, args.src_begin, args.src_begin + n,
                       &args.counters[C_ACT_BEGIN], &args.counters[C_ACTIVE]);
    HIP_CHECK(hipMemsetAsync(args.cand_count, 0, n * sizeof(uint32_t), st));
    HIP_CHECK(hipGetLastError());
}

static void launch_enum(Device *d, hipStream_t st, SsspArgs args) {
    if (args.n_items == 0) return;
    const bool quad = enum_uses_quad_gathers(d), prune = enum_prunes(d);
    sssp_fn fn;
    if (prune) fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, true, true, true> : sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, false, true, true>;
    else if (d->w8) fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, true, true, false> : sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, false, true, false>;
    else fn = quad ? sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, true, false, false> : sssp_enum_kernel<ENUM_WPB, ENUM_S1, ENUM_H1, ENUM_NB, false, false, false>;
    int occ = 1;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, ENUM_WPB * 64, 0));
    if (occ < 1) occ = 1;
    static const bool debug = std::getenv("MTG_DEBUG") != nullptr;
    if (debug) std::fprintf(stderr, "[mtg] enumeration level: %d workgroups of %d waves per CU\n", occ, ENUM_WPB);
    // (with pruning the number of searched sources stays on the GPU: the grid is sized for all of them, waves without a chunk leave at once)
    const uint64_t waves_needed = (args.n_items + 63) / 64;
    // (always the full grid: 8 waves per CU with twice the chunks per wave measured 20 % slower at 2^24, 65 % slower at 2^22)
    uint64_t grid = std::min<uint64_t>((uint64_t)d->n_cu * (uint64_t)occ, (waves_needed + ENUM_WPB - 1) / ENUM_WPB);
    grid = std::max<uint64_t>(grid, 1);
    HIP_CHECK(hipEventRecord(d->ev0, st));
    if (prune) {
        launch_active_list(d, st, args, args.n_items);
        args.act_index = d->d_act_index;
        args.act_node = d->d_act_node;
    } else {
        HIP_CHECK(hipMemsetAsync(args.cand_count, 0, args.n_items * sizeof(uint32_t), st));  // (part of the level: see the kernel's result stores)
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(ENUM_WPB * 64), 0, st, args);
    HIP_CHECK(hipGetLastError());
    const unsigned post_grid = (unsigned)std::min<uint64_t>((args.n_items + 255) / 256 + 1, (uint64_t)d->n_cu * 8);
    static_assert(ENUM_MAX_HITS <= 32, "the post-pass sorts up to 32 keys");
    hipLaunchKernelGGL(fix_compact_kernel, dim3(d->n_cu * 4), dim3(256), 0, st, args.fix_list, args.counters, d->d_fix_dense);
    hipLaunchKernelGGL((sort_lists_kernel<256>), dim3(3 * post_grid), dim3(256), 0, st, args.pool, args.pool_cap, args.cand_start,
                       args.cand_count, d->d_fix_dense, args.counters);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
}

static void launch_level(Device *d, hipStream_t st, const LevelCfg &cfg, bool count, SsspArgs args) {
    if (args.n_items == 0) return;
    const uint64_t n_batches = (args.n_items + cfg.bsrc - 1) / cfg.bsrc;
    sssp_fn fn = count ? cfg.fn_count : cfg.fn;
    int occ = 1;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, cfg.block, 0));
    if (occ < 1) occ = 1;
    uint64_t grid = (uint64_t)d->n_cu * (uint64_t)occ;
    unsigned long long *ws = nullptr;
    uint64_t ws_stride = 0;
    if (cfg.global_ws) {
        grid = std::min<uint64_t>(grid, 64);
        const uint64_t H = 1ull << cfg.logh;
        // table[H] u64 | stage[SCAP] u64 | log[QCAP] u32 | stage_src[SCAP] u16
        ws_stride = H + (uint64_t)cfg.scap + ((uint64_t)cfg.qcap + 1) / 2 + ((uint64_t)cfg.scap + 3) / 4;
        grid = std::min<uint64_t>(grid, n_batches);
        hu::device_malloc(&ws, grid * ws_stride * 8);
    }
    grid = std::max<uint64_t>(1, std::min<uint64_t>(grid, n_batches));
    args.ws = ws;
    args.ws_stride = ws_stride;
    HIP_CHECK(hipEventRecord(d->ev0, st));
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(cfg.block), 0, st, args);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
    if (ws) {
        HIP_CHECK(hipStreamSynchronize(st));
        hu::device_free(ws);
    }
}

// ------------------------------------------------------------------------------------------------
// Last level: no limit on the ball. One source at a time over DENSE arrays in HBM (a 32-bit tentative distance and a round
// stamp per node): frontier rounds of label-correcting relaxations (atomicMin; a node joins the next frontier once per
// round), until the frontier is empty; then one pass over the nodes collects the in-nodes within the bound. O(V) per source
// on top of the ball: only sources that overflowed the 2^22-entry global-workspace level land here (none on any genome graph;
// the reference's search has no such limit either, greedytigs/mod.rs:548-551), and the price buys "no abort on legal input".
// ------------------------------------------------------------------------------------------------
__global__ void dense_relax_kernel(const NodeBlock *recs, const uint32_t *ext_col, const uint16_t *ext_w, uint32_t K1, uint32_t wmask, const uint32_t *frontier,
                                   uint64_t n_front, uint32_t round, uint32_t *dist, uint32_t *stamp, uint32_t *next, unsigned long long *n_next) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    const uint32_t u = frontier[i];
    const uint32_t du = __hip_atomic_load(&dist[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const NodeBlock &r = recs[u];
    auto relax = [&](uint32_t v, uint32_t w) {
        const uint64_t nd = (uint64_t)du + w;
        if (nd > K1) return;
        const uint32_t old = atomicMin(&dist[v], (uint32_t)nd);
        if ((uint32_t)nd < old && atomicExch(&stamp[v], round) != round) next[atomicAdd(n_next, 1ull)] = v;
    };
    if (r.flags & F_EXT) {
        const uint64_t b = (uint64_t)r.nbr[0] | ((uint64_t)r.nbr[1] << 32);
        for (uint32_t e = 0; e < r.nbr[2]; e++) relax(ext_col[b + e], ext_w[b + e]);
    } else {
        for (uint32_t e = 0; e < r.deg; e++) relax(r.nbr[e], r.w[e] & wmask);
    }
}
__global__ void dense_collect_kernel(const NodeBlock *recs, uint64_t n_nodes, uint32_t source, uint32_t K1, const uint32_t *dist,
                                     unsigned long long *keys, unsigned long long *n_keys) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes || v == source) return;  // forbid_source_target
    const uint32_t d = dist[v];
    if (d <= K1 && (recs[v].flags & F_TARGET)) keys[atomicAdd(n_keys, 1ull)] = ((unsigned long long)d << 32) | v;
}
__global__ void dense_set_kernel(uint32_t *dist, uint32_t *frontier, uint32_t source) {
    dist[source] = 0;
    frontier[0] = source;
}
__global__ void dense_result_kernel(unsigned long long *cand_start, uint32_t *cand_count, uint64_t slot, unsigned long long start, uint32_t count) {
    cand_start[slot] = start;
    cand_count[slot] = count;
}

// Sources [0, n_src) of `list` (absolute source indices). Keys go to the pool at the cursor C_POOL like every other level's;
// their order (distance, node) -- all distinct -- is established by a sort of the collected keys on the host.
static void run_dense_level(Device *d, hipStream_t st, const SsspArgs &a, const uint32_t *d_list, uint64_t n_src) {
    const uint64_t V = d->V;
    std::vector<uint32_t> list(n_src);
    HIP_CHECK(hipMemcpyAsync(list.data(), d_list, n_src * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    std::sort(list.begin(), list.end());
    uint32_t *d_dist = nullptr, *d_stamp = nullptr, *d_front[2] = {nullptr, nullptr};
    unsigned long long *d_keys = nullptr, *d_n = nullptr;
    hu::device_malloc(&d_dist, V * 4);
    hu::device_malloc(&d_stamp, V * 4);
    hu::device_malloc(&d_front[0], V * 4);
    hu::device_malloc(&d_front[1], V * 4);
    hu::device_malloc(&d_keys, V * 8);
    hu::device_malloc(&d_n, 16);
    std::vector<unsigned long long> keys;
    std::vector<uint32_t> sources(1);
    for (const uint32_t idx : list) {
        HIP_CHECK(hipMemcpyAsync(sources.data(), d->d_out_nodes + idx, 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t s = sources[0];
        HIP_CHECK(hipMemsetAsync(d_dist, 0xFF, V * 4, st));
        HIP_CHECK(hipMemsetAsync(d_stamp, 0xFF, V * 4, st));
        hipLaunchKernelGGL(dense_set_kernel, dim3(1), dim3(1), 0, st, d_dist, d_front[0], s);
        uint64_t n_front = 1;
        int cur = 0;
        for (uint32_t round = 0; n_front; round++) {
            if (round == 0xFFFFFFFEu) MTG_DIE("dense search level: round counter exhausted");
            HIP_CHECK(hipMemsetAsync(d_n, 0, 8, st));
            hipLaunchKernelGGL(dense_relax_kernel, dim3((unsigned)((n_front + 255) / 256)), dim3(256), 0, st, a.recs, a.ext_col, a.ext_w, a.K1, a.wmask,
                               d_front[cur], n_front, round, d_dist, d_stamp, d_front[cur ^ 1], d_n);
            HIP_CHECK(hipGetLastError());
            unsigned long long h = 0;
            HIP_CHECK(hipMemcpyAsync(&h, d_n, 8, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            n_front = h;
            cur ^= 1;
        }
        HIP_CHECK(hipMemsetAsync(d_n, 0, 8, st));
        hipLaunchKernelGGL(dense_collect_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, a.recs, V, s, a.K1, d_dist, d_keys, d_n);
        HIP_CHECK(hipGetLastError());
        unsigned long long n_keys = 0;
        HIP_CHECK(hipMemcpyAsync(&n_keys, d_n, 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (n_keys > 0xFFFFFFFFull) MTG_DIE("dense search level: a candidate list beyond 2^32 entries");
        keys.resize(n_keys);
        if (n_keys) HIP_CHECK(hipMemcpyAsync(keys.data(), d_keys, n_keys * 8, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::sort(keys.begin(), keys.end(), [](unsigned long long x, unsigned long long y) { return mtg_policy_pops_before(x, y) != 0; });  // policy P1
        const unsigned long long start = d->h_counters[C_POOL];  // host mirror of the pool cursor (read_counters ran after the last level)
        d->h_counters[C_POOL] += n_keys;
        if (start + n_keys <= a.pool_cap && n_keys) HIP_CHECK(hipMemcpyAsync(a.pool + start, keys.data(), n_keys * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(dense_result_kernel, dim3(1), dim3(1), 0, st, a.cand_start, a.cand_count, (uint64_t)idx - a.src_begin, start, (uint32_t)n_keys);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    HIP_CHECK(hipMemcpyAsync(&d->d_counters[C_POOL], &d->h_counters[C_POOL], 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (void *p : {(void *)d_dist, (void *)d_stamp, (void *)d_front[0], (void *)d_front[1], (void *)d_keys, (void *)d_n}) hu::device_free(p);
}

static void read_counters(Device *d, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(d->h_counters, d->d_counters, C_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
}

// runs level 0 over [src_begin, src_end) and larger levels over whatever overflowed
static int run_levels(Device *d, hipStream_t st, int count_mode, uint64_t src_begin, uint64_t src_end, unsigned long long *d_pool,
                      uint64_t pool_cap, unsigned long long *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed,
                      mtg_sssp_stats *stats) {
    if (!d->d_recs) MTG_DIE("this device copy was built for one search (mtg_compute_tigs_cfg) and has given its search arrays back");
    if (!d->classified) MTG_DIE("mtg_sssp_candidates: call mtg_classify first");
    if (src_end > d->n_sources || src_begin > src_end) MTG_DIE("mtg_sssp_candidates: source range out of bounds");
    const uint64_t n = src_end - src_begin;
    const bool count = count_mode != 0;   // 1 = unit counters (cooperative plan), 2 = per-query performance data (one source per workgroup)
    const int first_coop = count_mode == 2 ? 3 : 0;
    HIP_CHECK(hipMemsetAsync(d->d_counters, 0, C_COUNT * sizeof(unsigned long long), st));
    SsspArgs a{};
    a.recs = d->d_recs; a.ext_col = d->d_ext_col; a.ext_w = d->d_ext_w;
    a.sources = d->d_out_nodes; a.src_index = nullptr; a.n_items = n; a.src_begin = src_begin;
    a.K1 = d->K1; a.pool = d_pool; a.pool_cap = pool_cap; a.cand_start = d_cand_start; a.cand_count = d_cand_count;
    a.counters = d->d_counters;
    a.wmask = d->w8 ? 0xFFu : 0xFFFFu;
    // count_mode 3 = the units of the pruned search (what the default plan really visits): the counting kernels over the searched sources only
    const bool prune_count = count_mode == 3;
    a.prune = (prune_count || !count) && enum_prunes(d) ? 1u : 0u;
    if (prune_count && !enum_prunes(d)) MTG_DIE("mtg_sssp_count_visited: this device graph / plan does not prune (k > 255 or plan + 4)");
    if (d->ovf_cap < n) {  // two overflow lists of up to n source indices each + the post-pass work list
        for (int i = 0; i < 2; i++) {
            if (d->d_ovf[i]) hu::device_free(d->d_ovf[i]);
            hu::device_malloc(&d->d_ovf[i], std::max<uint64_t>(n, 1) * sizeof(uint32_t));
        }
        if (d->d_fix) hu::device_free(d->d_fix);

        // (a wave abandons a work-list chunk with fewer than 64 free slots: < 1/7 of every chunk incl. its tag) + three open chunks per wave
        const uint64_t fix_slots = std::max<uint64_t>(n, 1) * 5 / 4 + (uint64_t)d->n_cu * 32 * 3 * ENUM_FIX_CHUNK;
        hu::device_malloc(&d->d_fix, fix_slots * sizeof(uint32_t));
        if (d->d_fix_dense) hu::device_free(d->d_fix_dense);
        hu::device_malloc(&d->d_fix_dense, std::max<uint64_t>(n, 1) * sizeof(uint32_t));
        d->ovf_cap = n;
    }
    a.ovf_list = d->d_ovf[0];
    a.fix_list = d->d_fix;
    double total_ms = 0.0;
    // the counting instantiations (untimed instrumentation) exist for the cooperative kernel only: it counts DISTINCT
    // settled nodes, an enumeration counts path steps
    const bool use_enum = (d->plan & 3) != 1 && !count && d->K1 < 0x8000u;  // (the enumeration level keeps 15-bit distances on its stack)
    const bool dense_only = d->K1 >= (1u << 21);  // (the cooperative levels keep 21-bit distances in their table entries)
    if (dense_only) {  // every source straight to the dense level: correct for any bound, O(V) per source
        if (count) MTG_DIE("the counting kernels need k - 1 < 2^21");
        std::vector<uint32_t> all(n);
        for (uint64_t i = 0; i < n; i++) all[i] = (uint32_t)(src_begin + i);
        if (n) HIP_CHECK(hipMemcpyAsync(d->d_ovf[0], all.data(), n * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        d->h_counters[C_POOL] = 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (n) run_dense_level(d, st, a, d->d_ovf[0], n);
        d->last_kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        d->last_n_levels = 1;
        d->last_level_ms[0] = d->last_kernel_ms;
        d->last_level_sources[0] = n;
        d->last_level_name[0] = "dense_relax_kernel rounds + dense_collect_kernel (one source at a time)";
        const bool small = d->h_counters[C_POOL] > pool_cap;
        if (pool_needed) *pool_needed = d->h_counters[C_POOL];
        return small ? 1 : 0;
    }
    uint64_t n_first = n;  // sources the first level really launches over
    if (use_enum) launch_enum(d, st, a);
    else if (prune_count) {
        if (n) {
            launch_active_list(d, st, a, n);
            read_counters(d, st);
            n_first = d->h_counters[C_ACTIVE];
            SsspArgs b = a;
            b.src_index = d->d_act_index + d->h_counters[C_ACT_BEGIN];
            b.n_items = n_first;
            launch_level(d, st, coop_level(first_coop, false), count, b);
        }
    } else launch_level(d, st, coop_level(first_coop, false), count, a);
    read_counters(d, st);
    if (use_enum && enum_prunes(d)) d->last_active_sources = d->h_counters[C_ACTIVE];
    d->last_n_levels = 0;
    if (n && n_first) {
        total_ms += elapsed_ms(d);
        d->last_level_ms[0] = elapsed_ms(d); d->last_level_sources[0] = n; d->last_n_levels = 1;
        d->last_level_name[0] = use_enum ? enum_level_name(enum_uses_quad_gathers(d), enum_prunes(d)) : coop_level(first_coop, false).name();
    }
    static const bool debug = std::getenv("MTG_DEBUG") != nullptr;
    if (debug && n && n_first) std::fprintf(stderr, "[mtg] level0 (%s): %llu sources (%llu searched), %.3f ms, %llu overflowed, cum settled %llu\n", use_enum ? "enum" : "coop level 0",
                                 (unsigned long long)n, (unsigned long long)(use_enum && enum_prunes(d) ? d->h_counters[C_ACTIVE] : n_first), elapsed_ms(d),
                                 (unsigned long long)d->h_counters[C_OVERFLOW], (unsigned long long)d->h_counters[C_SETTLED]);
#ifdef MTG_ENUM_STATS
    if (use_enum && n) std::fprintf(stderr, "[mtg] enum stats: wave steps %llu, lane steps %llu (%.1f of 64), starved %llu, block full %llu, step budget %llu, fix cursor %llu, pool cursor %llu\n",
                                    d->h_counters[C_SETTLED], d->h_counters[C_RELAXED], (double)d->h_counters[C_RELAXED] / (double)std::max<unsigned long long>(d->h_counters[C_SETTLED], 1),
                                    d->h_counters[C_EMITTED], d->h_counters[C_ATTEMPTS], d->h_counters[C_PUSHES], d->h_counters[C_FIX], d->h_counters[C_POOL]);
    if (use_enum && n) {  // when the waves end, and how many steps they ran
        static std::vector<unsigned long long> hp(16384 * 3);
        HIP_CHECK(hipMemcpyFromSymbol(hp.data(), HIP_SYMBOL(g_enum_prof), hp.size() * 8));
        std::vector<double> end_us, life_us, steps, start_us;
        unsigned long long t_min = ~0ull;
        for (int w = 0; w < 16384; w++) if (hp[3 * w + 1]) t_min = std::min(t_min, hp[3 * w]);
        for (int w = 0; w < 16384; w++)
            if (hp[3 * w + 1]) { start_us.push_back((hp[3 * w] - t_min) * 0.01); end_us.push_back((hp[3 * w + 1] - t_min) * 0.01); life_us.push_back((hp[3 * w + 1] - hp[3 * w]) * 0.01); steps.push_back((double)hp[3 * w + 2]); }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[(size_t)(q * (v.size() - 1))]; };
        std::fprintf(stderr, "[mtg] enum waves: %zu; end of wave (us after the first start) min %.0f p10 %.0f median %.0f p90 %.0f max %.0f; steps per wave min %.0f median %.0f max %.0f; us per step median %.3f\n",
                     end_us.size(), pct(end_us, 0), pct(end_us, 0.1), pct(end_us, 0.5), pct(end_us, 0.9), pct(end_us, 1), pct(steps, 0), pct(steps, 0.5), pct(steps, 1),
                     pct(life_us, 0.5) / std::max(1.0, pct(steps, 0.5)));
        std::fprintf(stderr, "[mtg] enum waves: start (us after the first) median %.0f p90 %.0f p99 %.0f max %.0f; life median %.0f p90 %.0f max %.0f us\n",
                     pct(start_us, 0.5), pct(start_us, 0.9), pct(start_us, 0.99), pct(start_us, 1), pct(life_us, 0.5), pct(life_us, 0.9), pct(life_us, 1));
        std::vector<unsigned long long> zero(16384 * 3, 0);
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_enum_prof), zero.data(), zero.size() * 8));
    }
#endif
    const uint64_t total_overflow = d->h_counters[C_OVERFLOW];
    // remaining levels over whatever overflowed the previous one; each launch appends the sources it could not finish
    // to the other of two ping-pong lists
    int cur_list = 0;
    // (what the enumeration level hands on are the heavy balls: batches of 32 of them do not fit the first cooperative level)
    for (int li = first_coop + 1; li < N_COOP_LEVELS && d->h_counters[C_OVERFLOW] > 0; li++) {
        const LevelCfg next = coop_level(li, use_enum);
        const uint64_t n_ovf = d->h_counters[C_OVERFLOW];
        HIP_CHECK(hipMemsetAsync(&d->d_counters[C_OVERFLOW], 0, sizeof(unsigned long long), st));
        SsspArgs b = a;
        b.src_index = d->d_ovf[cur_list];
        b.ovf_list = d->d_ovf[cur_list ^ 1];
        cur_list ^= 1;
        b.n_items = n_ovf;
        launch_level(d, st, next, count, b);
        read_counters(d, st);
        total_ms += elapsed_ms(d);
        if (d->last_n_levels < 8) {
            d->last_level_ms[d->last_n_levels] = elapsed_ms(d);
            d->last_level_sources[d->last_n_levels] = n_ovf;
            d->last_level_name[d->last_n_levels] = next.name();
            d->last_n_levels++;
        }
        if (debug) std::fprintf(stderr, "[mtg] level %d (coop, %d src/block): %llu sources, %.3f ms, %llu overflowed, cum settled %llu\n", li,
                                next.bsrc, (unsigned long long)n_ovf, elapsed_ms(d), (unsigned long long)d->h_counters[C_OVERFLOW], (unsigned long long)d->h_counters[C_SETTLED]);
    }
    if (d->h_counters[C_OVERFLOW] > 0) {  // balls beyond the 2^22 entries of the global-workspace level: the dense level has no limit
        const uint64_t n_ovf = d->h_counters[C_OVERFLOW];
        if (count) MTG_DIE("the counting kernels cannot follow %llu source(s) beyond the last cooperative level", (unsigned long long)n_ovf);
        const auto t0 = std::chrono::steady_clock::now();
        run_dense_level(d, st, a, d->d_ovf[cur_list], n_ovf);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        total_ms += ms;
        if (d->last_n_levels < 8) {
            d->last_level_ms[d->last_n_levels] = ms;
            d->last_level_sources[d->last_n_levels] = n_ovf;
            d->last_level_name[d->last_n_levels] = "dense_relax_kernel rounds + dense_collect_kernel (one source at a time)";
            d->last_n_levels++;
        }
        if (debug) std::fprintf(stderr, "[mtg] dense level: %llu sources, %.3f ms\n", (unsigned long long)n_ovf, ms);
        d->h_counters[C_OVERFLOW] = 0;
    }
    if (!count) d->last_kernel_ms = total_ms;
    if (stats) {
        stats->sources = n;
        stats->settled_nodes = d->h_counters[C_SETTLED];
        stats->relaxed_edges = d->h_counters[C_RELAXED];
        stats->emitted = d->h_counters[C_EMITTED];
        stats->relax_attempts = d->h_counters[C_ATTEMPTS];
        stats->overflow_sources = total_overflow;
    }
    // C_POOL is the pool cursor: keys plus the unused tails of block-local chunks (positions are launch-dependent,
    // (start,count) index the content). A retry may chunk differently, hence the slack.
    const bool too_small = !count && d->h_counters[C_POOL] > pool_cap;
    if (pool_needed) *pool_needed = d->h_counters[C_POOL] + (too_small ? (uint64_t)d->n_cu * 16 * POOL_CHUNK : 0);
    return too_small ? 1 : 0;
}

// ---- public (device.hpp) ----
int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Bytes of device memory a whole call on a graph of V nodes and E original edges has in use at its peak (device graph with the build's
// scratch, or the device graph beside the search and replay arrays, or the finish's dart arrays): the size of the arena chunk a
// first call reserves. Calibrated on G-csr graphs from 2^20 to 2^30 edges (MTG_DEBUG prints the arena's peak at the end of a call); an
// estimate that is too small costs a second chunk, one that is too large memory the driver has to map for nothing.
size_t device_call_bytes_estimate(uint64_t V, uint64_t E, uint64_t k) {
    (void)k;
    return (size_t)(V * 106 + E * 14) + (64u << 20);
}

// Kernel code objects load lazily, at the first launch of a kernel of their translation unit (3-10 ms each on a cold process); asking
// for a kernel's attributes loads them without launching anything.
// The first COOPERATIVE launch of a process costs 6 ms more than any later one (measured on the claim replay's rounds kernel, 10.7
// against 4.8 ms): an empty one is issued here, on the helper thread, on the stream the replay will use.
__global__ void coop_warm_kernel(unsigned) {}
void device_warm_device_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(classify_kernel));
    unsigned zero = 0;
    void *args[] = {&zero};
    (void)hipLaunchCooperativeKernel(reinterpret_cast<const void *>(coop_warm_kernel), dim3(1), dim3(64), args, 0, nullptr);
}
void device_warm_finish_kernels();  // finish_device.hip
void device_warm_euler_kernels();   // euler_device.hip

void device_arena_stats(int device_id, uint64_t out[4]) {
    hu::DeviceArena &a = hu::device_arena(device_id);
    std::lock_guard<std::mutex> lock(a.m);
    out[0] = a.chunk_bytes; out[1] = a.live_bytes; out[2] = a.peak_bytes; out[3] = a.n_chunk_allocs;
}
void device_arena_reset_peak(int device_id) {
    hu::DeviceArena &a = hu::device_arena(device_id);
    std::lock_guard<std::mutex> lock(a.m);
    a.peak_bytes = a.live_bytes;
}
static std::atomic<int> g_default_device{0};
void device_set_default(int device_id) { g_default_device.store(device_id); }
int device_get_default() { return g_default_device.load(); }

// Called when a host graph of V nodes and E edges comes into being (graph_build.cpp): on a helper thread, beside the host's own work,
// the HIP runtime starts, the code objects load, and the arena of the default device gets its chunk for the call that will follow --
// ONE hipMalloc, whose cost (nothing to 60 ms per GB depending on the box) no stage then waits for. Small graphs reserve nothing.
// A host-only constructor does not know which GPU the computation will name, so what it reserves is provisional: the device is
// remembered, and a call that computes elsewhere gives the chunk back when it ends (device_drop_foreign_reservation);
// mtg_set_reserve_ahead(0) turns the whole thing off for callers that want a constructor without GPU side effects. The helper threads
// are joinable: each new reservation joins the ones that are through, and the library's teardown joins the rest.
static std::atomic<int> g_reserve_ahead{1};
static std::atomic<int> g_reserved_device{-1};  // device of the last provisional reservation no call has confirmed yet
void device_set_reserve_ahead(int on) { g_reserve_ahead.store(on ? 1 : 0); }
namespace {
struct ReserveThreads {
    std::mutex m;
    std::vector<std::pair<std::thread, std::shared_future<void>>> th;
    void reap(bool all) {
        std::vector<std::thread> done;
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t i = 0; i < th.size();) {
                if (all || th[i].second.wait_for(std::chrono::seconds(0)) == std::future_status::ready) {
                    done.push_back(std::move(th[i].first));
                    th.erase(th.begin() + (long)i);
                } else i++;
            }
        }
        for (std::thread &t : done) if (t.joinable()) t.join();
    }
    ~ReserveThreads() { reap(true); }  // library teardown: no helper thread is left inside HIP when the process goes on to exit
};
ReserveThreads &reserve_threads() {
    static ReserveThreads r;
    return r;
}
}  // namespace
void device_reserve_async(uint64_t V, uint64_t E, int device_id) {
    const size_t bytes = device_call_bytes_estimate(V, E, 31);
    if (bytes < (256u << 20)) return;
    const bool provisional = device_id < 0;
    if (provisional && !g_reserve_ahead.load()) return;
    const int dev = device_id >= 0 ? device_id : device_get_default();
    std::promise<void> done;
    std::shared_future<void> fut = done.get_future().share();
    {
        hu::DeviceArena &arena = hu::device_arena(dev);
        std::lock_guard<std::mutex> lock(arena.m);
        // (one at a time; a reservation nobody had to wait for is over by now)
        if (arena.pending.valid() && arena.pending.wait_for(std::chrono::seconds(0)) != std::future_status::ready) return;
        arena.pending = fut;
    }
    if (provisional) g_reserved_device.store(dev);
    ReserveThreads &rt = reserve_threads();
    rt.reap(false);
    std::thread t([dev, bytes](std::promise<void> p) {
        if (device_count() > dev && hipSetDevice(dev) == hipSuccess) {
            hu::device_arena(dev).reserve(bytes);
            (void)hu::finish_stream(dev);
            {   // the pinned ring of the sliced transfers (4 x 16 MB of page-locked memory: 10 ms the first upload would pay)
                hu::TransferRing &r = hu::transfer_ring(dev);
                std::lock_guard<std::mutex> lock(r.m);
                r.ready();
            }
            device_warm_device_kernels();
            device_warm_finish_kernels();
            device_warm_euler_kernels();
        }
        (void)hipGetLastError();
        p.set_value();
    }, std::move(done));
    std::lock_guard<std::mutex> lock(rt.m);
    rt.th.emplace_back(std::move(t), fut);
}
// A call that computed on `used` (n of them): a provisional reservation that sits on another device goes back to the driver.
void device_drop_foreign_reservation(const int *used, int n) {
    const int dev = g_reserved_device.exchange(-1);
    if (dev < 0) return;
    for (int i = 0; i < n; i++) if (used[i] == dev) return;
    hu::DeviceArena &arena = hu::device_arena(dev);
    {
        std::unique_lock<std::mutex> lock(arena.m);
        if (arena.pending.valid()) {
            std::shared_future<void> f = arena.pending;
            arena.pending = std::shared_future<void>();
            lock.unlock();
            f.wait();
        }
    }
    arena.release_free_chunks(true);
}

// The goal-directed lower bounds of a device graph whose blocks hold plain weights (k <= 255): k - 1 rounds over a 32-bit distance
// array (lb), one pass for lb+ and the reach flags, and the blocks rewritten into the 8:8 format. A function of the graph alone;
// what it costs (HIP events: device_lower_bounds_ms) is the price of the pruned search, paid once per device graph -- a caller that
// searches once is better off without (mtg_compute_tigs_cfg builds none: 4.5 ms of full-ball search against 2 + 12 ms).
static void build_lower_bounds(Device *d, hipStream_t st) {
    const uint64_t V = d->V;
    if (!V || d->w8) return;
    const unsigned vb = (unsigned)((V + 255) / 256);
    uint32_t *d_D = nullptr;
    uint8_t *d_lb8 = nullptr;
    uint16_t *d_lbx = nullptr;
    hu::device_malloc(&d_D, V * 4);
    hu::device_malloc(&d_lb8, V);
    hu::device_malloc(&d_lbx, V * 2);
    HIP_CHECK(hipEventRecord(d->ev0, st));
    hipLaunchKernelGGL(lb_init_kernel, dim3(vb), dim3(256), 0, st, d->d_odeg, d->d_mirror, V, d_D);
    for (uint32_t r = 0; r < d->K1; r++)
        hipLaunchKernelGGL(lb_round_kernel, dim3(vb), dim3(256), 0, st, d->d_recs, d->d_ext_col, d->d_ext_w, V, r, d->K1, d_D);
    hipLaunchKernelGGL(lb_mirror_kernel, dim3(vb), dim3(256), 0, st, d->d_mirror, d_D, V, d_lb8);
    hipLaunchKernelGGL(build_lbx_kernel, dim3(vb), dim3(256), 0, st, V, d->d_recs, d->d_ext_col, d->d_ext_w, d_lb8, d->K1, d_lbx, d->d_odeg);
    hipLaunchKernelGGL(build_children_kernel<true>, dim3(vb), dim3(256), 0, st, V, d->d_recs, (const uint16_t *)d_lbx);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(d->ev1, st));
    HIP_CHECK(hipStreamSynchronize(st));
    d->lower_bounds_ms = elapsed_ms(d);
    d->w8 = true;
    hu::device_free(d_D);
    hu::device_free(d_lb8);
    hu::device_free(d_lbx);
}
// ... for a device graph that was built without them (a caller that turns out to iterate): the searchable-source list of the
// classification changes with the reach flags, so a classification that was there is repeated.
void device_build_lower_bounds(Device *d, void *stream) {
    HIP_CHECK(hipSetDevice(d->dev));
    if (d->k > 255 || d->w8) return;
    if (!d->d_recs) MTG_DIE("mtg_device_build_lower_bounds: this device copy has given its search arrays back");
    build_lower_bounds(d, (hipStream_t)stream);
    if (d->classified) (void)device_classify(d, stream);
}
double device_lower_bounds_ms(const Device *d) { return d->lower_bounds_ms; }
bool device_has_lower_bounds(const Device *d) { return d->w8; }

Device *device_create(const HostGraph &g, uint64_t k, int device_id, bool lower_bounds) {
    if (k < 1) MTG_DIE("k must be >= 1");
    if (k > 0xFFFFFFFFull) MTG_DIE("k = %llu is not supported by the device stage (32-bit distances)", (unsigned long long)k);
    if (k > 65535) {
        // Edge weights live in 16 bits, clamped to min(w, k) (an edge of weight >= k can never lie on a path within k - 1). With
        // k beyond 16 bits the clamp no longer fits, so every weight itself has to: true for any unitig set (a weight is a
        // number of k-mers of a unitig). Distances beyond 15 / 21 bits skip the enumeration / cooperative levels (run_levels).
        parallel_ranges(g.n_original_edges / 2, [&](uint64_t lo, uint64_t hi) {
            for (uint64_t u = lo; u < hi; u++)
                if (g.w_biedge[u] > 65534)
                    MTG_DIE("k = %llu with a unitig of %llu k-mers: beyond k = 65535 the device stage needs every weight below 65535",
                            (unsigned long long)k, (unsigned long long)g.w_biedge[u]);
        });
    }
    // weights: 16 bits per UNITIG, clamped to min(w, k) (both edges of a unitig weigh the same, host_graph.hpp); a unitig without
    // k-mers aborts here -- the bounded search, its lower bounds and the (distance, node) pop order of the reference's heap all need
    // weights >= 1 (the reference computes weight = len + 1 - k >= 1, bin.rs:369-376)
    const uint64_t U = g.n_original_edges / 2;
    PodVec<uint16_t> wclamp(U);
    parallel_ranges(U, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t u = lo; u < hi; u++) {
            const uint64_t w = g.w_biedge[u];
            if (w == 0)
                MTG_DIE("unitig %llu has weight 0: the bounded search needs weights >= 1 (the reference computes "
                        "weight = len + 1 - k >= 1, bin.rs:369-376)", (unsigned long long)u);
            wclamp[u] = (uint16_t)std::min<uint64_t>(w, k);
        }
    });
    if (device_count() <= device_id) MTG_DIE("no MI355X/HIP device %d available; libmatchtigs has no CPU path", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    Device *d = new Device();
    d->dev = device_id;
    d->k = k;
    if (MTG_POLICY_BOUND_EXCLUSIVE && k < 2) MTG_DIE("k must be >= 2 under the exclusive-bound policy");
    d->K1 = (uint32_t)mtg_policy_search_bound(k);  // the largest distance a target is found at: policy P2 (mtg_policy.h)
    d->V = g.node_count();
    d->E0 = g.n_original_edges;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    d->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    // The device graph is built ON the GPU from the ORIGINAL edges only (the search runs before any dummy edge exists, :678):
    // the host uploads from / to / clamped weight / mirror, the build kernels make the family blocks.
    const uint64_t V = d->V, E = g.n_original_edges;
    struct {
        const bool on = std::getenv("MTG_DEBUG") != nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            (void)hipDeviceSynchronize();
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mtg] device_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } dl;
    dl.lap("clamped weights (host)");
    hipStream_t st = nullptr;
    // One chunk of the device's arena for the whole call unless an earlier call (or the reservation the graph's construction started
    // on a helper thread, device_reserve_async) left one: every allocation below and in the stages that follow is a range of it.
    const size_t call_bytes = device_call_bytes_estimate(V, E, k);
    uint32_t *d_from = nullptr, *d_fill = nullptr, *d_need = nullptr, *d_row0 = nullptr, *d_adj0 = nullptr;
    uint16_t *d_w = nullptr;
    unsigned long long *d_ext_off = nullptr;
    // the long-lived arrays first (they collect at the front of the chunk), the build's scratch after them
    hu::device_malloc(&d->d_recs, std::max<uint64_t>(V, 1) * sizeof(NodeBlock), call_bytes);
    hu::device_malloc(&d->d_odeg, std::max<uint64_t>(V, 1) * 4);
    hu::device_malloc(&d->d_mirror, std::max<uint64_t>(V, 1) * 4);
    // (class bytes, multiplicities, out-node list: taken by the first classification, once the build's scratch arrays are back)
    d->n_cls_blocks = (V + CLS_NODES - 1) / CLS_NODES;
    hu::device_malloc(&d->d_block_counts, std::max<uint64_t>(d->n_cls_blocks, 1) * 2 * 4);  // source counts | positive multiplicities per block
    hu::device_malloc(&d->d_act_blocks, std::max<uint64_t>(d->n_cls_blocks, 1) * 4);
    hu::device_malloc(&d->d_act_total, 8);
    hu::device_malloc(&d->d_counters, C_COUNT * sizeof(unsigned long long));
    hu::device_malloc(&d_from, std::max<uint64_t>(E, 1) * 4);
    // the buckets of the original darts by from-node (row0[V + 1], adj0[E]: what the finishing stages keep with the graph) fall out
    // of build_nodes_kernel's sort; kept while they are small next to the stand-ins of BASELINE configs[4] (finish_device.hip)
    const bool want_buckets = E && (V + 1 + E) * 4 <= (8ull << 30) && !g.device_cache && !(hu::finish_tuning().flags.load() & hu::FT_NO_EDGE_CACHE);
    if (want_buckets) {
        hu::device_malloc(&d_row0, (V + 1) * 4);
        hu::device_malloc(&d_adj0, E * 4);
    }
    hu::device_malloc(&d_w, std::max<uint64_t>(U, 1) * 2);
    hu::device_malloc(&d_fill, std::max<uint64_t>(E, 1) * 4);  // rank of every edge among the out-edges of its from-node (arrival order of the counting pass)
    hu::device_malloc(&d_need, std::max<uint64_t>(V, 1) * 4);
    hu::device_malloc(&d_ext_off, std::max<uint64_t>(V, 1) * 8);
    HIP_CHECK(hipHostMalloc(&d->h_counters, C_COUNT * sizeof(unsigned long long)));
    HIP_CHECK(hipEventCreate(&d->ev0));
    HIP_CHECK(hipEventCreate(&d->ev1));
    for (auto &e : d->ev_r) HIP_CHECK(hipEventCreate(&e));
    dl.lap("allocations");
    // (through the pinned ring: host threads fill the next slice while one crosses PCIe -- the runtime stages a pageable upload on one
    // thread. from + mirror + 2 bytes per unitig: the heads are mirror(from(e ^ 1)), 6.3 B per edge + 4 B per node in all)
    // `from` first: the counting pass and the scans over its degrees (12 ms of kernels at 2^27) need nothing else and run on a stream
    // of their own (non-blocking: `st` is the legacy default stream, which every ordinary stream waits for) while the weights and the
    // mirror array (0.49 GB, 10 ms of PCIe) come up on `st`.
    if (E) hu::upload_sliced(d_from, g.e_from.data(), E * 4, st, device_id);
    dl.lap("upload of from");
    const unsigned eb = (unsigned)((E + 255) / 256), vb = (unsigned)((V + 255) / 256);
    uint64_t ext_total = 0;
    uint32_t *d_bs = nullptr;
    hipStream_t bs = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
    HIP_CHECK(hipMemsetAsync(d->d_odeg, 0, std::max<uint64_t>(V, 1) * 4, bs));
    if (V) {
        if (E) hipLaunchKernelGGL(build_count_kernel, dim3(eb), dim3(256), 0, bs, d_from, E, d->d_odeg, d_fill);
        hipLaunchKernelGGL(build_ext_need_kernel, dim3(vb), dim3(256), 0, bs, d->d_odeg, V, d_need);
        HIP_CHECK(hipGetLastError());
        scan_u32(d, bs, d->replay, d_need, V, d_ext_off, &d->d_counters[C_OVF_LIST]);
        if (want_buckets) {  // row0 = exclusive scan of the out-degrees
            hu::device_malloc(&d_bs, (hu::scan_blocks(V) + 2) * 4);
            hu::scan_u32<uint32_t>(bs, d->d_odeg, V, d_row0, d_bs, d_row0 + V);
        }
    }
    if (E) hu::upload_sliced(d_w, wclamp.data(), U * 2, st, device_id);
    if (V) hu::upload_sliced(d->d_mirror, g.mirror.data(), V * 4, st, device_id);  // (both calls return when the data is there)
    if (V) {
        read_counters(d, bs);
        if (d_bs) hu::device_free(d_bs);
        ext_total = d->h_counters[C_OVF_LIST];
    }
    HIP_CHECK(hipStreamSynchronize(bs));
    HIP_CHECK(hipStreamDestroy(bs));
    dl.lap("count + scans beside the uploads of weights and mirror");
    d->ext_n = ext_total;
    hu::device_malloc(&d->d_ext_col, std::max<uint64_t>(ext_total, 1) * 4);
    hu::device_malloc(&d->d_ext_w, std::max<uint64_t>(ext_total, 1) * 2);
    if (V) {
        if (E) hipLaunchKernelGGL(build_fill_kernel, dim3(eb), dim3(256), 0, st, d_from, E, d->d_odeg, d_ext_off, d_fill, d->d_recs, d->d_ext_col);
        hipLaunchKernelGGL(build_nodes_kernel, dim3(vb), dim3(256), 0, st, V, d->d_odeg, d->d_mirror, d_ext_off, d_from, d_w, d->d_recs,
                           d->d_ext_col, d->d_ext_w, d_row0, d_adj0);
        d->w8 = false;
        if (lower_bounds && k <= 255) build_lower_bounds(d, st);  // (writes the second halves too)
        else hipLaunchKernelGGL(build_children_kernel<false>, dim3(vb), dim3(256), 0, st, V, d->d_recs, (const uint16_t *)nullptr);
        HIP_CHECK(hipGetLastError());
    }
    dl.lap("build kernels (+ lower bounds)");
    // the finishing stages on this GPU start from the same arrays: from, mirror and the buckets stay with the graph
    uint32_t *d_mirror_copy = nullptr;
    hu::device_malloc(&d_mirror_copy, std::max<uint64_t>(V, 1) * 4);
    if (V) HIP_CHECK(hipMemcpyAsync(d_mirror_copy, d->d_mirror, V * 4, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hu::edge_cache_put(g, device_id, d_from, d_mirror_copy);
    if (want_buckets) hu::edge_cache_set_buckets(g, device_id, d_row0, d_adj0);
    for (void *p : {(void *)d_w, (void *)d_fill, (void *)d_need, (void *)d_ext_off}) hu::device_free(p);
    d->graph_bytes = V * sizeof(NodeBlock) + ext_total * 6 + V * 13;
    dl.lap("edge cache + frees");
    return d;
}

// What the stages of a STEP take from the arena beside a device graph that stays (classification, search lists, claim-replay records,
// the finish's dart arrays at their peak), calibrated on G-csr 2^24 / 2^27 / 2^30 (bench.py full_size.arena: 1.77 / 13.4 / 106.5 GB):
// 60 bytes per node + 62 per original edge, + 8 %.
size_t device_step_work_bytes_estimate(uint64_t V, uint64_t E) { return (size_t)((V * 60 + E * 62) / 100 * 108) + (64u << 20); }
// mtg_device_create_opts(MTG_DEVICE_RESERVE_WORK): a caller that will step through the stages with this device graph (classify /
// search / replay / finish, again and again) takes that memory NOW, as ONE chunk, unless the arena has it free already -- the first
// step then makes no driver call (five otherwise, each of which can stall for a second on this pool: DESIGN 2.1). Asked for
// explicitly, so the whole-call limit of the implicit reservations does not apply; what must remain is room for the other
// allocators of the process (a sixth of the device), else nothing is reserved and the arrays come piece by piece as before.
void device_reserve_step_work(Device *d) {
    HIP_CHECK(hipSetDevice(d->dev));
    const size_t want = device_step_work_bytes_estimate(d->V, d->E0);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return; }
    hu::device_arena(d->dev).ensure_free(want, /*explicit_request=*/true, free_b > total_b / 6 ? free_b - total_b / 6 : 0);
}

void device_free(Device *d) {
    if (!d) return;
    (void)hipSetDevice(d->dev);
    void *bufs[] = {d->d_recs, d->d_odeg, d->d_cls, d->d_ext_col, d->d_ext_w, d->d_mult, d->d_mirror, d->d_out_nodes, d->d_block_counts, d->d_counters,
                    d->d_act_index, d->d_act_node, d->d_act_blocks, d->d_act_total};
    for (void *b : bufs) hu::device_free(b);
    for (int i = 0; i < 2; i++) hu::device_free(d->d_ovf[i]);
    hu::device_free(d->d_fix);
    hu::device_free(d->d_fix_dense);
    ReplayWork &w = d->replay;
    void *rb[] = {w.state, w.resv[0], w.resv[1], w.touch, w.src_mirror, w.dense, w.claims, w.pending[0], w.pending[1], w.spill,
                  w.final_off, w.block_sums, w.ctl, w.out};
    for (void *b : rb) hu::device_free(b);
    (void)hipHostFree(w.h_ctl);
    (void)hipHostFree(w.h_out);
    (void)hipHostFree(d->h_counters);
    (void)hipEventDestroy(d->ev0);
    (void)hipEventDestroy(d->ev1);
    for (auto e : d->ev_r) (void)hipEventDestroy(e);
    delete d;
}

uint64_t device_graph_bytes(const Device *d) { return d->graph_bytes; }

uint64_t device_classify(Device *d, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    d->n_sources = 0;
    if (!d->d_cls) {
        hu::device_malloc(&d->d_cls, std::max<uint64_t>(d->V, 1));
        hu::device_malloc(&d->d_mult, std::max<uint64_t>(d->V, 1) * 4);
        hu::device_malloc(&d->d_out_nodes, std::max<uint64_t>(d->V, 1) * 4);
    }
    if (d->V) {
        static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(classify_kernel, dim3((unsigned)d->n_cls_blocks), dim3(CLS_BLOCK), 0, st, d->d_odeg, d->d_mirror, (uint32_t)d->V,
                           d->d_mult, d->d_cls, d->d_block_counts, d->d_block_counts + d->n_cls_blocks, d->d_act_blocks);
        HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(2), dim3(1024), 0, st, d->d_block_counts, (uint32_t)d->n_cls_blocks,
                           &d->d_counters[C_OVF_LIST], d->d_block_counts + d->n_cls_blocks, &d->d_counters[C_DEMAND],
                           d->d_act_blocks, d->d_act_total);
        HIP_CHECK(hipGetLastError());
        const auto t1 = std::chrono::steady_clock::now();
        read_counters(d, st);  // (the number of sources sizes the lists the compaction writes)
        if (dbg) std::fprintf(stderr, "[mtg] classify: two launches issued in %.3f ms, their results back after %.3f ms more\n",
                              std::chrono::duration<double, std::milli>(t1 - t0).count(),
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        d->n_sources = d->h_counters[C_OVF_LIST];
        d->total_demand = d->h_counters[C_DEMAND];
        const uint64_t act_need = d->w8 ? d->n_sources : 0;  // (without the 8:8 format no source carries the flag: the lists stay empty)
        if (d->act_cap < act_need || !d->d_act_index) {
            for (void *p : {(void *)d->d_act_index, (void *)d->d_act_node}) if (p) hu::device_free(p);
            hu::device_malloc(&d->d_act_index, std::max<uint64_t>(act_need, 1) * 4);
            hu::device_malloc(&d->d_act_node, std::max<uint64_t>(act_need, 1) * 4);
            d->act_cap = act_need;
        }
        hipLaunchKernelGGL(compact_sources_kernel, dim3((unsigned)d->n_cls_blocks), dim3(CLS_BLOCK), 0, st, d->d_cls,
                           (uint32_t)d->V, d->d_block_counts, d->d_out_nodes, d->d_act_blocks, d->d_act_index, d->d_act_node);
        HIP_CHECK(hipGetLastError());
    }
    d->classified = true;
    return d->n_sources;
}

void device_classify_download(Device *d, void *stream, uint32_t *out_nodes, int32_t *mult, uint8_t *live) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    if (!d->classified) MTG_DIE("mtg_classify_download: call mtg_classify first");
    if (out_nodes && d->n_sources)
        HIP_CHECK(hipMemcpyAsync(out_nodes, d->d_out_nodes, d->n_sources * 4, hipMemcpyDeviceToHost, st));
    if (mult && d->V) HIP_CHECK(hipMemcpyAsync(mult, d->d_mult, d->V * 4, hipMemcpyDeviceToHost, st));
    if (live && d->V) {
        uint8_t *d_live = nullptr;
        hu::device_malloc(&d_live, d->V);
        hipLaunchKernelGGL(export_live_kernel, dim3((unsigned)((d->V + 255) / 256)), dim3(256), 0, st, d->d_cls, (uint32_t)d->V, d_live);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpyAsync(live, d_live, d->V, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        hu::device_free(d_live);
    }
    HIP_CHECK(hipStreamSynchronize(st));
}

const uint32_t *device_d_out_nodes(const Device *d) { return d->d_out_nodes; }
uint64_t device_n_sources(const Device *d) { return d->n_sources; }

int device_sssp(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, uint64_t *d_pool, uint64_t pool_cap,
                uint64_t *d_cand_start, uint32_t *d_cand_count, uint64_t *pool_needed) {
    HIP_CHECK(hipSetDevice(d->dev));
    return run_levels(d, (hipStream_t)stream, 0, src_begin, src_end, (unsigned long long *)d_pool, pool_cap,
                      (unsigned long long *)d_cand_start, d_cand_count, pool_needed, nullptr);
}

void device_sssp_count(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = src_end - src_begin;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    const double keep_ms = d->last_kernel_ms;
    run_levels(d, (hipStream_t)stream, 1, src_begin, src_end, nullptr, 0, d_start, d_count, nullptr, stats);
    d->last_kernel_ms = keep_ms;
    hu::device_free(d_start);
    hu::device_free(d_count);
}

// the same counters for the search the default plan really runs: only the sources that can reach an in-node, successors pruned by
// their lower bounds (settled = distinct (source, node) pairs visited, relaxed = their out-edges, emitted = the same candidates)
void device_sssp_count_visited(Device *d, void *stream, uint64_t src_begin, uint64_t src_end, mtg_sssp_stats *stats) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = src_end - src_begin;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    const double keep_ms = d->last_kernel_ms;
    run_levels(d, (hipStream_t)stream, 3, src_begin, src_end, nullptr, 0, d_start, d_count, nullptr, stats);
    if (stats) stats->sources = d->h_counters[C_ACTIVE];
    d->last_kernel_ms = keep_ms;
    hu::device_free(d_start);
    hu::device_free(d_count);
}
bool device_prunes(const Device *d) { return enum_prunes(d); }
uint64_t device_last_active_sources(const Device *d) { return d->last_active_sources; }

// greedytigs/mod.rs:647-673 counters in the engine's terms (see mtg_dijkstra_performance_data): one source per workgroup
void device_performance_data(Device *d, void *stream, mtg_dijkstra_performance_data *out) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = d->n_sources;
    unsigned long long *d_start = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, std::max<uint64_t>(n, 1) * 8);
    hu::device_malloc(&d_count, std::max<uint64_t>(n, 1) * 4);
    mtg_sssp_stats st{};
    run_levels(d, (hipStream_t)stream, 2, 0, n, nullptr, 0, d_start, d_count, nullptr, &st);
    hu::device_free(d_start);
    hu::device_free(d_count);
    out->dijkstras = n;
    out->iterations = st.settled_nodes;
    out->heap_pushes = d->h_counters[C_PUSHES];
    out->unnecessary_heap_elements = d->h_counters[C_PUSHES] - st.settled_nodes;
    out->max_max_heap_size = d->h_counters[C_MAX_LOG];
    out->max_max_distance_array_size = d->h_counters[C_MAX_ENT];
    out->sum_max_heap_size = d->h_counters[C_PUSHES];
    out->sum_max_distance_array_size = st.settled_nodes;
}

double device_last_kernel_ms(const Device *d) { return d->last_kernel_ms; }
const char *device_last_level_name(const Device *d, int level) {
    return level >= 0 && level < d->last_n_levels ? d->last_level_name[level].c_str() : "";
}
int device_last_levels(const Device *d, double *ms, uint64_t *sources, int cap) {
    const int n = d->last_n_levels < cap ? d->last_n_levels : cap;
    for (int i = 0; i < n; i++) { ms[i] = d->last_level_ms[i]; sources[i] = d->last_level_sources[i]; }
    return n;
}

// ------------------------------------------------------------------------------------------------
// GPU claim replay (replay_kernels.inc): host driver
// ------------------------------------------------------------------------------------------------
template <typename In>
static void scan_values(hipStream_t st, ReplayWork &w, In in, uint64_t n, unsigned long long *out, unsigned long long *d_total) {
    const uint64_t nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb > w.cap_blocks) {
        if (w.block_sums) hu::device_free(w.block_sums);
        hu::device_malloc(&w.block_sums, nb * 8);
        w.cap_blocks = nb;
    }
    hipLaunchKernelGGL(scan_reduce_kernel<In>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n, w.block_sums);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, w.block_sums, nb, d_total);
    hipLaunchKernelGGL(scan_apply_kernel<In>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, n, w.block_sums, out);
    HIP_CHECK(hipGetLastError());
}
static void scan_u32(Device *d, hipStream_t st, ReplayWork &w, const uint32_t *in, uint64_t n, unsigned long long *out,
                     unsigned long long *d_total) {
    scan_values(st, w, ScanInU32{in}, n, out, d_total);
    (void)d;
}

// Claims for ALL n_sources classified sources (candidate arrays indexed by absolute source index, device pointers).
// Returns the number of pairs; *pairs_out (host, malloc'd) holds them in the reference's push order.
uint64_t device_replay(Device *d, void *stream, uint64_t n_sources, const uint64_t *d_cand_start, const uint32_t *d_cand_count,
                       const uint64_t *d_pool, mtg_pair **pairs_out, int *rounds_out) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    if (!d->classified || n_sources != d->n_sources) MTG_DIE("mtg_replay_claims_device: classify first; n_sources must be all sources");
    ReplayWork &w = d->replay;  // buffers are re-used across calls on this device
    const uint64_t V = d->V, S = n_sources;
    const auto t_replay_begin = std::chrono::steady_clock::now();
    d->last_wall_s[1] = d->last_wall_s[2] = 0;
    struct {  // MTG_DEBUG=1: host-side wall clock of the call's segments
        const bool on = std::getenv("MTG_DEBUG") != nullptr;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        void lap(const char *what) {
            if (!on) return;
            const auto n = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[mtg] replay: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
            t = n;
        }
    } rt;
    if (S == 0) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d->last_replay_rounds = 0;
        d->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    if (V > w.cap_v) {
        if (w.state) hu::device_free(w.state);
        for (int i = 0; i < 2; i++) if (w.resv[i]) { hu::device_free(w.resv[i]); w.resv[i] = nullptr; }
#if MTG_REPLAY_RECORDS
        hu::device_malloc(&w.state, ((V + 1) / 2) * 64);  // one 64-byte record per pair of numeric neighbours: states + both reservation words
#else
        hu::device_malloc(&w.state, (V + 2) * 8);  // (states are read in aligned pairs: the last pair may reach one word beyond V)
        hu::device_malloc(&w.resv[0], V * 8);
        hu::device_malloc(&w.resv[1], V * 8);
#endif
        w.cap_v = V;
        w.tag_base = 0xFFFFFFFFu;  // forces the clear below
    }
    if (S > w.cap_s) {
        if (w.touch) {
            hu::device_free(w.touch); hu::device_free(w.src_mirror); hu::device_free(w.claims);
            hu::device_free(w.pending[0]); hu::device_free(w.pending[1]); hu::device_free(w.final_off);
        }
        hu::device_malloc(&w.touch, S * sizeof(Touch));
        hu::device_malloc(&w.src_mirror, S * 4);
        hu::device_malloc(&w.claims, S * 8);
        hu::device_malloc(&w.pending[0], S * 4);
        hu::device_malloc(&w.pending[1], S * 4);
        hu::device_malloc(&w.final_off, S * 8);
        w.cap_s = S;
    }
    const uint64_t spill_need = std::max<uint64_t>(d->total_demand, 1);  // a source emits at most its demand (classification)
    if (spill_need > w.cap_spill) {
        if (w.spill) hu::device_free(w.spill);
        hu::device_malloc(&w.spill, spill_need * 4);
        w.cap_spill = spill_need;
    }
    if (!w.ctl) {
        hu::device_malloc(&w.ctl, RC_COUNT * 8);
        HIP_CHECK(hipHostMalloc(&w.h_ctl, RC_COUNT * 8));
    }
    // reservation tags decrease with every round of every call, so the two reservation arrays are never cleared; only when
    // the 32-bit tag space is used up (or the arrays are new)
#if MTG_REPLAY_RECORDS
    w.tag_base = 0;  // (the records are written whole below, reservation words included)
#endif
    if ((uint64_t)w.tag_base + REPLAY_MAX_ROUNDS + 128 >= 0xFFFFFFF0ull) {
#if MTG_REPLAY_RECORDS
#else
        HIP_CHECK(hipMemsetAsync(w.resv[0], 0xFF, V * 8, st));
        HIP_CHECK(hipMemsetAsync(w.resv[1], 0xFF, V * 8, st));
#endif
        w.tag_base = 0;
    }
    HIP_CHECK(hipEventRecord(d->ev_r[0], st));
    // working copy of the classification state; per-source outputs start at "nothing claimed"
    {
#if MTG_REPLAY_RECORDS
        const uint64_t n_quarters = ((V + 1) / 2) * 4;
        hipLaunchKernelGGL(replay_record_init_kernel, dim3((unsigned)((n_quarters + 255) / 256)), dim3(256), 0, st, d->d_mirror, d->d_mult, d->d_cls, V, w.state);
#else
        ReplayArgs ia{};
        ia.state = w.state; ia.resv[0] = w.resv[0]; ia.resv[1] = w.resv[1];
        hipLaunchKernelGGL(replay_state_init_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d->d_mirror, d->d_mult, d->d_cls, V, ia);
#endif
        HIP_CHECK(hipGetLastError());
    }
    // the sources that have candidates, in order, with the static words of their admission (Dense)
    uint64_t n_dense = 0;
    {
        static_assert(DENSE_BLOCK == SCAN_BLOCK, "the dense fill uses the scan's block offsets");
        const uint64_t nb = (S + SCAN_BLOCK - 1) / SCAN_BLOCK;
        if (nb > w.cap_blocks) {
            if (w.block_sums) hu::device_free(w.block_sums);
            hu::device_malloc(&w.block_sums, nb * 8);
            w.cap_blocks = nb;
        }
        unsigned long long *total = &d->d_counters[C_OVF_LIST];
        hipLaunchKernelGGL(scan_reduce_kernel<ScanInHasCand>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, ScanInHasCand{d_cand_count}, S, w.block_sums);
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(SCAN_BLOCK), 0, st, w.block_sums, nb, total);
        HIP_CHECK(hipGetLastError());
        read_counters(d, st);
        n_dense = d->h_counters[C_OVF_LIST];
        if (n_dense > w.cap_dense) {
            if (w.dense) hu::device_free(w.dense);
            hu::device_malloc(&w.dense, n_dense * sizeof(Dense));
            w.cap_dense = n_dense;
        }
        hipLaunchKernelGGL(replay_dense_fill_kernel, dim3((unsigned)nb), dim3(DENSE_BLOCK), 0, st, d->d_out_nodes, d_cand_count,
                           (const unsigned long long *)d_cand_start, (const unsigned long long *)d_pool, d->d_mirror, S, w.block_sums, w.dense, w.src_mirror, w.claims);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemsetAsync(w.ctl, 0, RC_COUNT * 8, st));

    ReplayArgs a{};
    a.out_nodes = d->d_out_nodes; a.state = w.state;
    a.cand_start = (const unsigned long long *)d_cand_start; a.cand_count = d_cand_count; a.pool = (const unsigned long long *)d_pool;
    a.resv[0] = w.resv[0]; a.resv[1] = w.resv[1]; a.touch = w.touch; a.src_mirror = w.src_mirror; a.claims = w.claims;
    a.dense = w.dense; a.n_dense = n_dense; a.spill = w.spill; a.pending[0] = w.pending[0]; a.pending[1] = w.pending[1]; a.ctl = w.ctl; a.n_sources = S;
    a.tag_base = w.tag_base; a.max_rounds = REPLAY_MAX_ROUNDS;

    // index-ordered admission windows over the dense list: ~32 K listed sources each, between 4 and 48 of them (a round costs a
    // grid barrier plus one dependent-access chain, so tiny windows are latency bound; huge ones bring the waiting visits back)
    {
        // Round 5: 36 windows instead of 48, the second half of them twice as large -- the late windows meet mostly dead candidates (their
        // retries fall from a third of a window to nothing), so they can be larger and the rounds fewer: 50 -> 39 rounds, rounds kernel
        // 4.41 -> 4.14 ms at 2^27, 1.58 -> 1.46 ms at 2^24 (tools/replay_window_sweep.py; fewer windows of ONE size lose more to
        // retries than they save in rounds: 36 equal windows 4.65 ms)
        uint64_t n_win = std::max<uint64_t>(4, std::min<uint64_t>(36, (n_dense + (1u << 15) - 1) >> 15));
        // (tuning only -- the pair list does not depend on any of it: bits 0-15 of tune_windows = number of windows, bits 16-23 = the
        // sixteenth of them from which they grow, bits 24-31 = by which factor)
        uint64_t grow_16th = n_win >= 8 ? 8 : 16, grow_mul = n_win >= 8 ? 2 : 1;
        if (d->tune_windows & 0xFFFF) { n_win = d->tune_windows & 0xFFFF; grow_16th = 16; grow_mul = 1; }
        if ((d->tune_windows >> 16) & 0xFFFF) { grow_16th = (d->tune_windows >> 16) & 0xFF; grow_mul = std::max<uint64_t>(1, (d->tune_windows >> 24) & 0xFF); }
        const uint64_t g = std::min<uint64_t>(n_win, n_win * grow_16th / 16);  // windows of the base size
        // g windows of `window` sources, the other n_win - g of grow_mul times as many
        a.window = std::max<uint64_t>((n_dense + g + (n_win - g) * grow_mul - 1) / std::max<uint64_t>(g + (n_win - g) * grow_mul, 1), 256);
        a.grow_from = g;
        a.grow_mul = grow_mul;
        {
            const uint64_t base_cover = g * a.window;
            a.n_windows = n_dense <= base_cover ? (n_dense + a.window - 1) / a.window
                                                : g + (n_dense - base_cover + a.window * grow_mul - 1) / (a.window * grow_mul);
        }
        // Bit 32 of the tuning word lets the windows ADAPT on top of this list (replay_kernels.inc: while a round is short and the share
        // of checks sent on does not grow, every third round doubles the window, up to a sixth of the list). Measured, not the default:
        // how often sources block each other is a property of the graph -- on a real compacted de Bruijn graph (100 Mbp) a fifth of
        // the checks are retried whatever the window and eight windows beat thirty-six by 2.4 x, on the G-csr graphs retries explode
        // with the window (8 windows at 2^24: 4.5 x the visits) and do so several rounds AFTER the doubling that caused them, so a
        // controller that serves the first (1.68 -> 1.44 ms) costs the second (2^22: 1.08 -> 1.64 ms). tools/replay_window_sweep.py.
        a.window_cap = (d->tune_windows >> 32) & 1u ? std::max<uint64_t>(a.window, n_dense / 6) : 0;
    }
    const uint64_t widest = a.window_cap ? a.window_cap : a.window * a.grow_mul;
    // one cooperative launch for all rounds: the runtime refuses a grid that cannot be co-resident, so the grid barrier cannot
    // deadlock. Workgroups of 1024 when a round's tiles (a window to admit, about as many sources to check) fill the device, of 256
    // when they do not (replay_kernels.inc); as many workgroups as a round has tiles, at least 64, at most what is co-resident.
    if (w.grid == 0) {
        int coop = 0;
        HIP_CHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, d->dev));
        if (!coop) MTG_DIE("device %d does not support cooperative launches (needed by the claim replay's grid barrier)", d->dev);
        int occ = 0, occ_small = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, replay_rounds_kernel<REPLAY_BLOCK>, REPLAY_BLOCK, 0));
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_small, replay_rounds_kernel<REPLAY_BLOCK_SMALL>, REPLAY_BLOCK_SMALL, 0));
        if (occ < 1 || occ_small < 1) MTG_DIE("replay_rounds_kernel does not fit a compute unit");
        w.grid = (unsigned)d->n_cu * (unsigned)std::min(occ, 2);
        w.grid_small = (unsigned)d->n_cu * (unsigned)std::min(occ_small, 2);
    }
    // (by the WIDEST window since round 5: a real de Bruijn graph of 100 Mbp -- 4.6 M listed sources, windows of 86 K and 172 K -- ran
    // its rounds kernel in 2.60 ms with workgroups of 256 and runs it in 1.68 ms with workgroups of 1024: a third as many arrivals at
    // the grid barrier and tile counters; the G-csr graph of 2^24, windows of 19 K and 39 K, stays with 256)
    bool small = 2 * ((widest + REPLAY_BLOCK - 1) / REPLAY_BLOCK) < (uint64_t)d->n_cu;
    if (d->tune_block) small = d->tune_block < REPLAY_BLOCK;  // (tuning only)
    const uint64_t block = small ? REPLAY_BLOCK_SMALL : REPLAY_BLOCK;
    unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)(small ? w.grid_small : w.grid), (n_dense + block - 1) / block,
                                                                       std::max<uint64_t>(64, 2 * (((a.window_cap ? widest : a.window) + block - 1) / block))}));
    if (d->tune_grid) grid = std::max(1u, std::min(grid, (unsigned)d->tune_grid));  // (tuning only)
    // one workgroup in role_mod admits, the others check (the longer chain). Measured with the per-XCD barrier: 2^27 (the grid fills
    // the device) role_mod 2 / 4 / 6 / 8 = 6.6 / 6.7 / 7.2 / 8.2 ms; 2^24 (64 workgroups) 2.7 / 2.2 / 2.4 / 2.8 ms
    a.role_mod = (!small && grid >= w.grid) ? 2u : 4u;
    if (d->tune_role_mod) a.role_mod = (uint32_t)std::max(1, d->tune_role_mod);  // (tuning only)
    a.plain_barrier = d->tune_plain_barrier ? 1u : 0u;
    void *kargs[] = {&a};
    rt.lap("buffers + launches");
    HIP_CHECK(hipEventRecord(d->ev_r[1], st));
    HIP_CHECK(hipLaunchCooperativeKernel(small ? reinterpret_cast<void *>(replay_rounds_kernel<REPLAY_BLOCK_SMALL>) : reinterpret_cast<void *>(replay_rounds_kernel<REPLAY_BLOCK>),
                                         dim3(grid), dim3((unsigned)block), kargs, 0, st));
    HIP_CHECK(hipEventRecord(d->ev_r[2], st));
    HIP_CHECK(hipMemcpyAsync(w.h_ctl, w.ctl, RC_COUNT * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    rt.lap("state init + rounds kernel");
#ifdef MTG_REPLAY_PROF
    {
        std::vector<unsigned long long> hp((size_t)grid * 64 * 4);
        HIP_CHECK(hipMemcpy(hp.data(), d_prof, hp.size() * 8, hipMemcpyDeviceToHost));
        const int nr = std::min<int>(64, (int)w.h_ctl[RC_ROUNDS]);
        for (int r = 0; r < nr; r += (r < 8 ? 1 : 8)) {
            for (int role = 0; role < 2; role++) {
                double sw = 0, mw = 0, sf = 0, mf = 0, sb = 0, mb = 0; int n = 0;
                for (unsigned b = role; b < grid; b += 2) {
                    const unsigned long long *t = &hp[((size_t)b * 64 + r) * 4];
                    const double wk = (t[1] - t[0]) * 0.01, fl = (t[2] - t[1]) * 0.01, ba = (t[3] - t[2]) * 0.01;
                    sw += wk; mw = std::max(mw, wk); sf += fl; mf = std::max(mf, fl); sb += ba; mb = std::max(mb, ba); n++;
                }
                if (n) std::fprintf(stderr, "[mtg] replay prof: round %2d %s: work mean %.1f max %.1f us, flush mean %.1f max %.1f, barrier wait mean %.1f max %.1f\n",
                                    r, role ? "check" : "admit", sw / n, mw, sf / n, mf, sb / n, mb);
            }
        }
    }
#endif
    if (w.h_ctl[RC_ABORT]) MTG_DIE("claim replay: a workgroup never reached the grid barrier (watchdog)");
    const int rounds = (int)w.h_ctl[RC_ROUNDS];
    w.tag_base += (uint32_t)rounds + 2;
    d->last_replay_visits = 0;
    for (int r = 0; r < rounds && r < RC_TRACE_ROUNDS; r++) d->last_replay_visits += w.h_ctl[RC_TRACE + 2 * r];
    static const bool replay_debug = std::getenv("MTG_DEBUG") != nullptr;
    if (replay_debug) {
        std::fprintf(stderr, "[mtg] replay: %llu listed sources, windows from %llu (cap %llu), widest admitted %llu; %llu checks sent on\n", (unsigned long long)n_dense,
                     (unsigned long long)a.window, (unsigned long long)a.window_cap, (unsigned long long)w.h_ctl[RC_WIDEST], (unsigned long long)w.h_ctl[RC_RETRIED]);
        std::fprintf(stderr, "[mtg] replay: %d rounds, %llu left; per round (pending, us since kernel start):", rounds, (unsigned long long)w.h_ctl[RC_LEFT]);
        for (int r = 0; r < rounds && r < RC_TRACE_ROUNDS; r++)
            std::fprintf(stderr, " (%llu, %.0f)", (unsigned long long)w.h_ctl[RC_TRACE + 2 * r], (double)(w.h_ctl[RC_TRACE + 2 * r + 1] - w.h_ctl[RC_LEFT_PAR + 2]) * 0.01);
        std::fprintf(stderr, "\n");
    }
    uint64_t n_left = w.h_ctl[RC_LEFT];
    if (n_left > 0) {  // very long priority chain: finish the rest in order on one GPU thread
        uint32_t *lst = w.pending[w.h_ctl[RC_LEFT_PAR] & 1];
        std::vector<uint32_t> rest(n_left);
        HIP_CHECK(hipMemcpyAsync(rest.data(), lst, n_left * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        std::sort(rest.begin(), rest.end());
        HIP_CHECK(hipMemcpyAsync(lst, rest.data(), n_left * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(replay_tail_kernel, dim3(1), dim3(64), 0, st, a, lst, n_left);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));
    }
    d->last_replay_rounds = rounds;
    if (rounds_out) *rounds_out = rounds;

    // compaction in source order
    unsigned long long *cnt = &d->d_counters[C_OVF_LIST];
    uint64_t n_pairs = 0;
    if (n_dense) {  // (over the sources with candidates only: nobody else can claim)
        scan_values(st, w, ScanInClaims{w.claims, w.dense}, n_dense, w.final_off, cnt);
        read_counters(d, st);
        n_pairs = d->h_counters[C_OVF_LIST];
    }
    rt.lap("tail + scan");
    d->last_n_pairs = n_pairs;
    if (n_pairs > w.cap_out || !w.out) {
        if (w.out) hu::device_free(w.out);
        hu::device_malloc(&w.out, std::max<uint64_t>(n_pairs, 1) * sizeof(mtg_pair));
        w.cap_out = std::max<uint64_t>(n_pairs, 1);
    }
    if (n_pairs) {
        hipLaunchKernelGGL(replay_compact_kernel, dim3((unsigned)((n_dense + 255) / 256)), dim3(256), 0, st, a, w.final_off, w.out);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipEventRecord(d->ev_r[3], st));
    HIP_CHECK(hipStreamSynchronize(st));
    {
        float a_ms = 0, b_ms = 0;
        HIP_CHECK(hipEventElapsedTime(&a_ms, d->ev_r[1], d->ev_r[2]));
        HIP_CHECK(hipEventElapsedTime(&b_ms, d->ev_r[0], d->ev_r[3]));
        d->last_replay_kernel_ms = a_ms;
        d->last_replay_gpu_ms = b_ms;
    }
    const auto t_replay_done = std::chrono::steady_clock::now();
    d->last_wall_s[1] = std::chrono::duration<double>(t_replay_done - t_replay_begin).count();
    d->last_wall_s[2] = 0;
    if (!pairs_out) {  // the pairs stay in HBM for a finish on this GPU (device_resident_pairs / device_take_pairs)
        rt.lap("compact (pairs stay on the GPU)");
        return n_pairs;
    }
    mtg_pair *host = (mtg_pair *)big_malloc(std::max<uint64_t>(n_pairs, 1) * sizeof(mtg_pair));  // (the caller frees it with free())
    if (!host) MTG_DIE("out of memory");
    if (n_pairs) {
        if (n_pairs < (1u << 18)) {
            HIP_CHECK(hipMemcpyAsync(host, w.out, n_pairs * sizeof(mtg_pair), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        } else {
            // through a pinned staging buffer kept with the device, in slices: while slice i+1 crosses PCIe, a few host threads
            // copy slice i into the caller's (pageable, freshly allocated) array
            if (n_pairs > w.cap_h_out) {
                if (w.h_out) HIP_CHECK(hipHostFree(w.h_out));
                w.cap_h_out = n_pairs + n_pairs / 4;
                // (coherent, the default: host threads read each slice right after hipEventSynchronize on an event recorded behind
                // its copy; non-coherent host memory is only guaranteed visible after a stream / device synchronize)
                HIP_CHECK(hipHostMalloc(&w.h_out, w.cap_h_out * sizeof(mtg_pair), hipHostMallocDefault));
            }
            const uint64_t n_slices = std::min<uint64_t>(8, (n_pairs + (1u << 18) - 1) >> 18);
            const uint64_t slice = (n_pairs + n_slices - 1) / n_slices;
            std::vector<hipEvent_t> ev(n_slices);
            for (uint64_t i = 0; i < n_slices; i++) {
                const uint64_t lo = i * slice, n = std::min(slice, n_pairs - lo);
                HIP_CHECK(hipMemcpyAsync(w.h_out + lo, w.out + lo, n * sizeof(mtg_pair), hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventReleaseToSystem));
                HIP_CHECK(hipEventRecord(ev[i], st));
            }
            const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({8, (uint64_t)std::thread::hardware_concurrency(), n_pairs >> 18}));
            double dbg_wait = 0, dbg_copy = 0;
            auto worker = [&](unsigned t) {
                for (uint64_t i = 0; i < n_slices; i++) {
                    const uint64_t lo = i * slice, n = std::min(slice, n_pairs - lo);
                    const uint64_t a0 = lo + n * t / T, a1 = lo + n * (t + 1) / T;
                    const auto t0 = std::chrono::steady_clock::now();
                    HIP_CHECK(hipEventSynchronize(ev[i]));
                    const auto t1 = std::chrono::steady_clock::now();
                    std::memcpy(host + a0, w.h_out + a0, (a1 - a0) * sizeof(mtg_pair));
                    if (t == 0) {
                        dbg_wait += std::chrono::duration<double, std::milli>(t1 - t0).count();
                        dbg_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
                    }
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; t++) th.emplace_back(worker, t);
            worker(0);
            for (auto &x : th) x.join();
            for (uint64_t i = 0; i < n_slices; i++) HIP_CHECK(hipEventDestroy(ev[i]));
            if (rt.on) std::fprintf(stderr, "[mtg] replay: pair download: %llu slices, %u host threads; thread 0 waited %.3f ms for copies, copied for %.3f ms\n",
                                    (unsigned long long)n_slices, T, dbg_wait, dbg_copy);
        }
    }
    rt.lap("compact + pair download");
    d->last_wall_s[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_replay_done).count();
    *pairs_out = host;
    return n_pairs;
}

int device_set_plan(Device *d, int plan) {
    if (plan >= 0 && plan <= 7) d->plan = (plan & 3) == 1 ? 1 : plan;
    return d->plan;
}
int device_last_replay_rounds(const Device *d) { return d->last_replay_rounds; }
void device_set_replay_tuning(Device *d, uint64_t windows, int block, int grid, int role_mod, int plain_barrier) {
    d->tune_windows = windows; d->tune_block = block; d->tune_grid = grid; d->tune_role_mod = role_mod; d->tune_plain_barrier = plain_barrier != 0;
}
void device_last_pairs_wall_s(const Device *d, double out[3]) { for (int i = 0; i < 3; i++) out[i] = d->last_wall_s[i]; }
void device_last_replay_ms(const Device *d, double out[2]) { out[0] = d->last_replay_kernel_ms; out[1] = d->last_replay_gpu_ms; }
int device_id_of(const Device *d) { return d->dev; }
// was d built from g (same node and edge counts) for bound k - 1? What the resident pairs of d may be finished on.
bool device_matches(const Device *d, const HostGraph &g, uint64_t k) { return d->V == g.node_count() && d->E0 == g.n_original_edges && d->k == k; }
// the pairs of the last claim replay as they lie in HBM (valid until the next replay on this device)
const mtg_pair *device_resident_pairs(const Device *d, uint64_t *n_out) {
    if (n_out) *n_out = d->last_n_pairs;
    return d->last_n_pairs ? d->replay.out : nullptr;
}
// the same, handed over: the caller owns the device array now (device_free_array) -- lets the device graph go before the finish starts
mtg_pair *device_take_pairs(Device *d, uint64_t *n_out) {
    if (n_out) *n_out = d->last_n_pairs;
    mtg_pair *p = d->replay.out;
    d->replay.out = nullptr;
    d->replay.cap_out = 0;
    d->last_n_pairs = 0;
    return p;
}
void device_free_array(int device_id, void *p) {
    hu::device_free_on(device_id, p);
}
// host copy of the resident pairs (malloc'd)
uint64_t device_download_pairs(Device *d, mtg_pair **pairs_out) {
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t n = d->last_n_pairs;
    mtg_pair *host = (mtg_pair *)big_malloc(std::max<uint64_t>(n, 1) * sizeof(mtg_pair));
    if (!host) MTG_DIE("out of memory");
    if (n) HIP_CHECK(hipMemcpy(host, d->replay.out, n * sizeof(mtg_pair), hipMemcpyDeviceToHost));
    *pairs_out = host;
    return n;
}
uint64_t device_last_replay_visits(const Device *d) { return d->last_replay_visits; }

// one-shot path: SSSP candidates for all sources into engine-owned device buffers (pool grown on demand), then the
// claim replay on the GPU; only the matched pairs travel to the host.
// A device copy that is searched once (the consuming call, clib.rs:291): the family blocks (64 of the ~100 bytes per node a call holds
// at its peak) and the search's lists are dead once the candidates exist, and the claim replay's arrays take their place in the arena
// instead of new memory beside them.
void device_set_single_use(Device *d) { d->single_use = true; }
static void device_drop_search_arrays(Device *d) {
    HIP_CHECK(hipDeviceSynchronize());
    for (void **p : {(void **)&d->d_recs, (void **)&d->d_ext_col, (void **)&d->d_ext_w, (void **)&d->d_act_index, (void **)&d->d_act_node,
                     (void **)&d->d_ovf[0], (void **)&d->d_ovf[1], (void **)&d->d_fix, (void **)&d->d_fix_dense}) {
        if (*p) hu::device_arena(d->dev).free(*p, false);
        *p = nullptr;
    }
    d->ovf_cap = 0;
    d->act_cap = 0;
}

uint64_t device_pairs(Device *d, void *stream, mtg_pair **pairs_out, int *rounds_out) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    d->last_wall_s[0] = d->last_wall_s[1] = d->last_wall_s[2] = 0;
    if (!S) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    unsigned long long *d_start = nullptr, *d_pool = nullptr;
    uint32_t *d_count = nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    hu::device_malloc(&d_start, S * 8);
    hu::device_malloc(&d_count, S * 4);
    uint64_t cap = std::max<uint64_t>(S * 2 + std::min<uint64_t>((S + 63) / 64, (uint64_t)d->n_cu * 8) * ENUM_POOL_CHUNK, 1024);  // keys + per-wave chunk slack
    for (;;) {
        hu::device_malloc(&d_pool, cap * 8);
        uint64_t needed = 0;
        if (run_levels(d, st, 0, 0, S, d_pool, cap, d_start, d_count, &needed, nullptr) == 0) break;
        hu::device_free(d_pool);
        d_pool = nullptr;
        cap = needed + needed / 8 + 1024;
    }
    if (d->single_use) device_drop_search_arrays(d);
    d->last_wall_s[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    const uint64_t n = device_replay(d, stream, S, (const uint64_t *)d_start, d_count, (const uint64_t *)d_pool, pairs_out, rounds_out);
    hu::device_free(d_pool);
    hu::device_free(d_start);
    hu::device_free(d_count);
    return n;
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU inside the library: the graph is replicated on every device, the ascending source list is cut into
// contiguous blocks of equal estimated work (1 + out-degree of the source), every device runs the SSSP stage over its block
// from its own host thread, the candidate lists are gathered on the first device with peer copies over xGMI (one copy of the
// counts, one of the starts, one of the keys per device: concatenation in device order IS the replay order), and the claim
// replay runs there. No other exchange exists on the path.
// ------------------------------------------------------------------------------------------------
__global__ void source_work_kernel(const uint32_t *out_nodes, const uint32_t *odeg, uint64_t n, uint32_t *work) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) work[i] = 1u + (odeg[out_nodes[i]] & ~ODEG_REACH);
}
// cut[r] = first source whose exclusive work prefix reaches total * r / parts (r = 1 .. parts-1)
__global__ void work_cuts_kernel(const unsigned long long *prefix, const uint32_t *work, uint64_t n, const unsigned long long *total, int parts,
                                 unsigned long long *cut) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long lo = prefix[i], hi = lo + work[i], tot = *total;
    for (int r = 1; r < parts; r++) {
        const unsigned long long target = tot / (unsigned)parts * (unsigned)r + tot % (unsigned)parts * (unsigned)r / (unsigned)parts;
        if (lo <= target && target < hi) cut[r] = i;
    }
}
__global__ void rebase_starts_kernel(unsigned long long *start, uint64_t n, unsigned long long offset) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) start[i] += offset;
}

std::vector<uint64_t> device_partition_sources(Device *d, void *stream, int parts) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    std::vector<uint64_t> cuts((size_t)parts + 1, 0);
    cuts[(size_t)parts] = S;
    if (parts <= 1 || S == 0) {
        for (int r = 1; r < parts; r++) cuts[(size_t)r] = S;
        return cuts;
    }
    uint32_t *d_work = nullptr;
    unsigned long long *d_prefix = nullptr, *d_cut = nullptr;
    hu::device_malloc(&d_work, S * 4);
    hu::device_malloc(&d_prefix, S * 8);
    hu::device_malloc(&d_cut, (size_t)parts * 8);
    HIP_CHECK(hipMemsetAsync(d_cut, 0, (size_t)parts * 8, st));
    hipLaunchKernelGGL(source_work_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d->d_out_nodes, d->d_odeg, S, d_work);
    scan_u32(d, st, d->replay, d_work, S, d_prefix, &d->d_counters[C_OVF_LIST]);
    hipLaunchKernelGGL(work_cuts_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, d_prefix, d_work, S, &d->d_counters[C_OVF_LIST], parts, d_cut);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> h((size_t)parts);
    HIP_CHECK(hipMemcpyAsync(h.data(), d_cut, (size_t)parts * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    for (int r = 1; r < parts; r++) cuts[(size_t)r] = std::max<uint64_t>(cuts[(size_t)r - 1], h[(size_t)r]);
    hu::device_free(d_work); hu::device_free(d_prefix); hu::device_free(d_cut);
    return cuts;
}

// All devices must hold the same graph and be classified. Returns the pairs of the whole graph (host, malloc'd).
uint64_t device_pairs_multi(Device *const *devs, int n_dev, mtg_pair **pairs_out, int *rounds_out, double *gather_ms_out) {
    if (n_dev < 1) MTG_DIE("device_pairs_multi: no device");
    if (n_dev == 1) return device_pairs(devs[0], nullptr, pairs_out, rounds_out);
    Device *d0 = devs[0];
    const uint64_t S = d0->n_sources;
    for (int i = 1; i < n_dev; i++)
        if (devs[i]->V != d0->V || devs[i]->n_sources != S) MTG_DIE("device_pairs_multi: the devices hold different graphs");
    if (!S) {
        if (pairs_out) *pairs_out = (mtg_pair *)std::malloc(sizeof(mtg_pair));
        d0->last_n_pairs = 0;
        if (rounds_out) *rounds_out = 0;
        return 0;
    }
    const auto t_begin = std::chrono::steady_clock::now();
    const std::vector<uint64_t> cuts = device_partition_sources(d0, nullptr, n_dev);
    struct Part { unsigned long long *start = nullptr, *pool = nullptr; uint32_t *count = nullptr; uint64_t used = 0; };
    std::vector<Part> parts((size_t)n_dev);
    std::vector<std::thread> th;
    for (int i = 0; i < n_dev; i++) {
        th.emplace_back([&, i]() {  // one host thread per device: its block of sources, its own buffers
            Device *d = devs[i];
            Part &p = parts[(size_t)i];
            HIP_CHECK(hipSetDevice(d->dev));
            const uint64_t lo = cuts[(size_t)i], hi = cuts[(size_t)i + 1], n = hi - lo;
            if (!n) return;
            hu::device_malloc(&p.start, n * 8);
            hu::device_malloc(&p.count, n * 4);
            uint64_t cap = std::max<uint64_t>(n * 2 + std::min<uint64_t>((n + 63) / 64, (uint64_t)d->n_cu * 8) * ENUM_POOL_CHUNK, 1024);
            for (;;) {
                hu::device_malloc(&p.pool, cap * 8);
                uint64_t needed = 0;
                if (run_levels(d, nullptr, 0, lo, hi, p.pool, cap, p.start, p.count, &needed, nullptr) == 0) { p.used = needed; break; }
                hu::device_free(p.pool);
                p.pool = nullptr;
                cap = needed + needed / 8 + 1024;
            }
        });
    }
    for (auto &t : th) t.join();
    // gather on the first device
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipSetDevice(d0->dev));
    uint64_t pool_total = 0;
    for (const Part &p : parts) pool_total += p.used;
    unsigned long long *g_start = nullptr, *g_pool = nullptr;
    uint32_t *g_count = nullptr;
    hu::device_malloc(&g_start, S * 8);
    hu::device_malloc(&g_count, S * 4);
    hu::device_malloc(&g_pool, std::max<uint64_t>(pool_total, 1) * 8);
    uint64_t off = 0;
    for (int i = 0; i < n_dev; i++) {
        const Part &p = parts[(size_t)i];
        const uint64_t lo = cuts[(size_t)i], n = cuts[(size_t)i + 1] - lo;
        if (!n) continue;
        auto copy = [&](void *dst, const void *src, size_t bytes) {
            if (!bytes) return;
            if (devs[i]->dev == d0->dev) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, d0->dev, src, devs[i]->dev, bytes, nullptr));
        };
        copy(g_start + lo, p.start, n * 8);
        copy(g_count + lo, p.count, n * 4);
        copy(g_pool + off, p.pool, p.used * 8);
        if (off) hipLaunchKernelGGL(rebase_starts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, g_start + lo, n, (unsigned long long)off);
        off += p.used;
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(nullptr));
    if (gather_ms_out) *gather_ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < n_dev; i++) {
        HIP_CHECK(hipSetDevice(devs[i]->dev));
        if (parts[(size_t)i].start) { hu::device_free(parts[(size_t)i].start); hu::device_free(parts[(size_t)i].count); hu::device_free(parts[(size_t)i].pool); }
    }
    HIP_CHECK(hipSetDevice(d0->dev));
    if (d0->single_use) device_drop_search_arrays(d0);
    d0->last_wall_s[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    const uint64_t n = device_replay(d0, nullptr, S, (const uint64_t *)g_start, g_count, (const uint64_t *)g_pool, pairs_out, rounds_out);
    hu::device_free(g_pool); hu::device_free(g_start); hu::device_free(g_count);
    return n;
}

// convenience for tests: allocates device buffers, grows the pool on demand, downloads to host vectors
void device_candidates_to_host(Device *d, void *stream, std::vector<uint64_t> &cand_start, std::vector<uint32_t> &cand_count,
                               std::vector<uint64_t> &pool) {
    hipStream_t st = (hipStream_t)stream;
    HIP_CHECK(hipSetDevice(d->dev));
    const uint64_t S = d->n_sources;
    cand_start.assign(S, 0);
    cand_count.assign(S, 0);
    pool.clear();
    if (!S) return;
    unsigned long long *d_start = nullptr, *d_pool = nullptr;
    uint32_t *d_count = nullptr;
    hu::device_malloc(&d_start, S * 8);
    hu::device_malloc(&d_count, S * 4);
    uint64_t cap = std::max<uint64_t>(S * 4, 1024);
    for (;;) {
        hu::device_malloc(&d_pool, cap * 8);
        uint64_t needed = 0;
        const int rc = run_levels(d, st, 0, 0, S, d_pool, cap, d_start, d_count, &needed, nullptr);
        if (rc == 0) {
            pool.resize(needed);
            if (needed) HIP_CHECK(hipMemcpyAsync(pool.data(), d_pool, needed * 8, hipMemcpyDeviceToHost, st));
            break;
        }
        hu::device_free(d_pool);
        d_pool = nullptr;
        cap = needed + needed / 8 + 1024;
    }
    HIP_CHECK(hipMemcpyAsync(cand_start.data(), d_start, S * 8, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(cand_count.data(), d_count, S * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    hu::device_free(d_pool);
    hu::device_free(d_start);
    hu::device_free(d_count);
}

}  // namespace mtg
