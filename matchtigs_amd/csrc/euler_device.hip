// euler_device.hip -- Euler bicycles of the Eulerised bigraph on the MI355X.
//
// Alternative to the sequential Hierholzer walk the reference gets from bigraph 5.0.1
// (`compute_minimum_bidirected_eulerian_cycle_decomposition`, call sites
// /root/reference/src/implementation/greedytigs/mod.rs:722 and eulertigs/mod.rs:119; host version with the
// reference's order: euler_fast.cpp). A parallel formulation cannot reproduce the sequential tie-breaks, so the
// ORDER of the walks differs from the reference's; what it guarantees is what the reference's callers rely on:
// one closed biwalk per connected component, every biedge (an edge or its mirror, never both) exactly once,
// consecutive edges adjacent. It is therefore opt-in (mtg_set_euler_mode) and judged on those invariants.
//
// Method (Atallah/Vishkin-style, adapted to mirror pairs; all arrays indexed by directed edge = "dart"; the mirror
// of dart e is e ^ 1):
//   1. bucket the darts by from-node (degree count, scan, slot fill, per-node sort): adj[row[v] + i] = i-th out-dart
//      of v in ascending dart id.
//   2. pair every in-dart of a node with an out-dart, mirror-symmetrically: for dart e = (u -> v), j = slot of
//      e^1 among the out-darts of mirror(v); succ[e] = out(v)[j] (self-mirror v: out(v)[j ^ 1]). By construction
//      succ[succ[e] ^ 1] = e ^ 1, so the closed trails succ defines come in disjoint mirror pairs.
//   3. label the trail PAIRS (a trail together with its mirror trail) by a lock-free union-find over the BIEDGES (root = smallest
//      biedge id): the passage e -> succ[e] and its mirror passage succ[e]^1 -> e^1 join the same two biedges, so one of the two
//      does the union -- half the atomics and half the label arrays of a union-find over darts, same components.
//   4. spanning forest over (biedge components x binodes): rounds of deterministic hooking -- every binode proposes,
//      for each passage whose component differs from its passage 0's, to hang the larger component root below the
//      smaller; a root accepts its smallest proposal (64-bit atomicMin) -- until every binode sees one component.
//      Each accepted proposal selects its passage; a node then rotates the successors of passage 0 and its selected
//      passages (and, mirrored, at the mirror node), which merges all their trails into one. Roots only hang below
//      smaller ids, so the accepted proposals form a forest: every one joins two trails that are still distinct, and
//      the result is exactly one trail pair per connected component.
//   5. rank the final trails: splitters (every 64th dart id + the smallest dart of every component) walk to the next
//      splitter and record the darts they pass, the reduced list is ranked by pointer jumping (two levels), and the
//      recorded segments are copied to their final positions. Only the trail that contains the component's smallest
//      dart is emitted.
// No step depends on thread timing (atomics are only used for counts, minima and idempotent flags), so the same graph
// always gives the same walks. Everything is integer gather/scatter work; no MFMA.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <vector>

#include "device.hpp"
#include "finish_device.hpp"
#include "hip_util.hpp"

namespace mtg {

using namespace hu;

namespace {

// ---- lock-free union-find (links always point to a smaller id, so a root is the smallest id of its set) -----
__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
    uint32_t cur = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur != x) {
        uint32_t prev = x, next;
        while (cur > (next = __hip_atomic_load(&parent[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(&parent[prev], next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
            prev = cur;
            cur = next;
        }
    }
    return cur;
}
// true iff THIS call linked two sets (the successful calls of any concurrent execution form a spanning forest)
__device__ __forceinline__ bool uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    while (a != b) {
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        const uint32_t old = atomicCAS(&parent[a], a, b);  // a > b: hang root a below b
        if (old == a) return true;
        a = uf_find(parent, old);
        b = uf_find(parent, b);
    }
    return false;
}

__device__ __forceinline__ bool bit_of(const uint32_t *bits, uint32_t e) { return (bits[e >> 5] >> (e & 31)) & 1u; }

// ---- step 1: bucket darts by from-node ------------------------------------------------------------------------
// (dart ids are 32-bit and may use all 32 bits: element indices are computed in 64 bits, hu::gid())
// (the count and the dart's place in its bucket come from the same atomic: `rank` is read back, coalesced, by the fill -- one random
// atomic per dart instead of two)
__global__ __launch_bounds__(EB) void degree_rank_kernel(const uint32_t *from, uint64_t n_darts, uint32_t *deg, uint32_t *rank) {
    const uint64_t e = gid();
    if (e < n_darts) rank[e] = atomicAdd(&deg[from[e]], 1u);
}
__global__ __launch_bounds__(EB) void fill_kernel(const uint32_t *from, uint64_t n_darts, const uint32_t *row, const uint32_t *rank,
                                                 uint32_t *adj) {
    const uint64_t e = gid();
    if (e >= n_darts) return;
    adj[row[from[e]] + rank[e]] = (uint32_t)e;
}

// buckets are filled in atomic order; sorting each node's few out-darts by id makes slots (and with them the whole
// decomposition) independent of thread timing
__global__ __launch_bounds__(EB) void sort_buckets_kernel(uint64_t n_nodes, const uint32_t *row, uint32_t *adj, uint32_t *pos) {
    const uint64_t v = gid();
    if (v >= n_nodes) return;
    const uint32_t lo = row[v], hi = row[v + 1];
    for (uint32_t i = lo + 1; i < hi; i++) {  // insertion sort (degrees are tiny in a de Bruijn graph)
        const uint32_t x = adj[i];
        uint32_t j = i;
        while (j > lo && adj[j - 1] > x) {
            adj[j] = adj[j - 1];
            j--;
        }
        adj[j] = x;
    }
    if (pos)
        for (uint32_t i = lo; i < hi; i++) pos[adj[i]] = i - lo;
}

// ---- step 2: mirror-symmetric pairing --------------------------------------------------------------------------
// Dart e = (u -> v) is followed by the out-dart of v in the slot that e ^ 1 = (mirror v -> mirror u) has in the bucket of mirror v
// (for a self-mirror node: the neighbouring slot), which makes the successor function commute with mirroring.
// Evaluated from the nodes' side: the j-th in-dart of v is the mirror of the j-th out-dart of mirror(v) (both buckets in
// ascending dart id), so a node reads its own bucket and its mirror's and writes the successors -- no slot array, no per-dart
// lookups of from / mirror / row (buckets + pairing: 32 -> see DESIGN 4.7 / docs/history).
__global__ __launch_bounds__(EB) void succ_node_kernel(const uint32_t *mirror, uint64_t n_nodes, const uint32_t *row, const uint32_t *adj,
                                                      uint32_t *succ, uint32_t *error) {
    const uint64_t i = gid();
    if (i >= n_nodes) return;
    const uint32_t v = (uint32_t)i, vm = mirror[v];
    const uint32_t lo = row[v], dv = row[v + 1] - lo;
    if (v == vm) {
        if (dv & 1) {
            atomicOr(error, 1u);
            return;
        }
        for (uint32_t j = 0; j < dv; j++) succ[adj[lo + j] ^ 1u] = adj[lo + (j ^ 1u)];
        return;
    }
    const uint32_t lom = row[vm];
    if (dv != row[vm + 1] - lom) {
        atomicOr(error, 1u);  // not Eulerian
        return;
    }
    for (uint32_t j = 0; j < dv; j++) succ[adj[lom + j] ^ 1u] = adj[lo + j];
}
// ---- step 3: labels of the trail pairs, over biedges ---------------------------------------------------------------
// A biedge b has exactly two neighbours in the graph whose components are the trail pairs: succ[2b] >> 1 and succ[2b + 1] >> 1
// (the passage e -> f and its mirror f^1 -> e^1 name the same two biedges, so the predecessors are the same two). Components of
// this 2-regular graph, ECL-CC style: the first pass links every biedge to its SMALLER neighbour without any atomic (one coalesced
// 8-byte load per biedge, no random access: a forest in which every parent is smaller than its child); the second pass only has to
// hook the edges the first could not express -- a biedge both of whose neighbours are smaller brings its second one (a third of
// the biedges of a random order) -- through the lock-free union (roots hang below smaller roots, so the final root of a component
// is its smallest biedge, whatever the thread timing).
__global__ __launch_bounds__(EB) void label_init_kernel(const uint32_t *succ, uint64_t n_biedges, uint32_t *parent) {
    const uint64_t b = gid();
    if (b >= n_biedges) return;
    const uint2 s2 = reinterpret_cast<const uint2 *>(succ)[b];
    const uint32_t n1 = s2.x >> 1, n2 = s2.y >> 1, m = n1 < n2 ? n1 : n2;
    parent[b] = m < (uint32_t)b ? m : (uint32_t)b;
}
__global__ __launch_bounds__(EB) void label_hook_kernel(const uint32_t *succ, uint64_t n_biedges, uint32_t *parent) {
    const uint64_t b = gid();
    if (b >= n_biedges) return;
    const uint2 s2 = reinterpret_cast<const uint2 *>(succ)[b];
    const uint32_t n1 = s2.x >> 1, n2 = s2.y >> 1;
    const uint32_t hi = n1 < n2 ? n2 : n1;  // the neighbour the first pass did not link (when both are smaller than b)
    if (hi < (uint32_t)b && hi != (n1 < n2 ? n1 : n2)) uf_union(parent, (uint32_t)b, hi);
}
// In place: parent[e] = root of e for EVERY e when the kernel ends. The find must not compress here: a path-halving store of
// another thread (parent[e] = some ancestor it read earlier) could land after this thread's parent[e] = root and leave a
// non-root behind -- harmless for a union-find that is only ever read through uf_find, wrong for an array read as labels.
// Every slot is written by its own thread only; readers passing through see the old parent or the root, both ancestors.
// (parent2, optional: the second union-find -- over the LABELS, i.e. the roots of this one -- starts as the identity on them; a
// non-root is never looked up there, so only the roots are written: no pass of its own over all biedges)
__global__ __launch_bounds__(EB) void flatten_kernel(uint32_t *parent, uint64_t n, uint32_t *parent2) {
    const uint64_t e = gid();
    if (e >= n) return;
    uint32_t cur = (uint32_t)e, next;
    while ((next = __hip_atomic_load(&parent[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != cur) cur = next;
    __hip_atomic_store(&parent[e], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (parent2 && cur == (uint32_t)e) parent2[e] = (uint32_t)e;
}
// the same for the second union-find after a hooking round: only the labels (comp[b] == b) are ever looked up in it, and only the
// few that hang below another label change -- one streaming pass over comp instead of a chase and a store per biedge
__global__ __launch_bounds__(EB) void flatten_labels_kernel(const uint32_t *comp, uint32_t *parent2, uint64_t n) {
    const uint64_t e = gid();
    if (e >= n || comp[e] != (uint32_t)e) return;
    uint32_t cur = (uint32_t)e, next;
    while ((next = __hip_atomic_load(&parent2[cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != cur) cur = next;
    if (cur != (uint32_t)e) __hip_atomic_store(&parent2[e], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ---- step 4: spanning forest (deterministic hooking rounds) + successor rotation ------------------------------
// Component labels and roots are biedge ids (the smallest biedge of the trail pair / of the merged component).
// Round: every binode proposes, for each passage i >= 1 whose component root differs from passage 0's, to hang the
// LARGER root below the smaller one; a root keeps the smallest proposal (smaller target root, then smaller passage
// dart) by a 64-bit atomicMin, so the outcome does not depend on thread timing. Roots only ever hang below smaller ids,
// so the accepted proposals of all rounds form a forest over the components.
template <typename F>
__device__ __forceinline__ void for_each_passage(const uint32_t *mirror, const uint32_t *row, const uint32_t *adj, uint32_t v, F &&f) {
    const uint32_t vm = mirror[v];
    if (vm < v) return;  // the lower node of a pair works for both
    const uint32_t d = row[v + 1] - row[v];
    const bool self = vm == v;
    const uint32_t n_pass = self ? d / 2 : d;
    // passage i: in-dart a_i -> out-dart b_i at v (and, mirrored, b_i^1 -> a_i^1 at mirror v)
    for (uint32_t i = 0; i < n_pass; i++) {
        const uint32_t a = (self ? adj[row[v] + 2 * i] : adj[row[vm] + i]) ^ 1u;
        const uint32_t b = self ? adj[row[v] + 2 * i + 1] : adj[row[v] + i];
        f(i, a, b);
    }
}
__global__ __launch_bounds__(EB) void propose_kernel(const uint32_t *mirror, uint64_t n_nodes, const uint32_t *row, const uint32_t *adj,
                                                    const uint32_t *comp, uint32_t *parent2, unsigned long long *best, uint32_t *proposed,
                                                    uint8_t *active, int first_round) {
    const uint64_t vi = gid();
    if (vi >= n_nodes) return;
    const uint32_t v = (uint32_t)vi;
    // components only ever merge: a binode whose passages were all in one component in an earlier round has nothing to propose in
    // any later one (the last round, which only finds out that nobody proposes, then reads one byte per node)
    if (!first_round && !active[v]) return;
    {   // a binode with a single passage has nothing to merge (and its mirror node is handled by the lower node of the pair)
        const uint32_t vm = mirror[v];
        const uint32_t d = row[v + 1] - row[v];
        if (vm < v || (vm == v ? d / 2 : d) < 2) { active[v] = 0; return; }
    }
    bool mine = false;
    uint32_t r0 = 0;
    for_each_passage(mirror, row, adj, v, [&](uint32_t i, uint32_t, uint32_t b) {
        // (first round: the second union-find is still the identity -- the label IS the root, one gather instead of two)
        const uint32_t r = first_round ? comp[b >> 1] : uf_find(parent2, comp[b >> 1]);
        if (i == 0) {
            r0 = r;
            return;
        }
        if (r == r0) return;
        const uint32_t hi = r > r0 ? r : r0, lo = r > r0 ? r0 : r;
        const unsigned long long key = ((unsigned long long)lo << 32) | b;
        // a long trail receives one proposal per shared binode: filter the losers before they queue up on its word
        if (key < __hip_atomic_load(&best[hi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&best[hi], key);
        *proposed = 1u;  // (every proposing thread stores the same value: a round without proposals ends the hooking)
        mine = true;
    });
    active[v] = mine ? 1 : 0;
}
// `best` holds ~0 everywhere except at the roots that received a proposal this round; the kernel resets what it consumes,
// so the array is cleared once per call and not once per round
__global__ __launch_bounds__(EB) void hook_kernel(uint64_t n_biedges, uint32_t *parent2, unsigned long long *best, uint32_t *selected) {
    const uint64_t r = gid();
    if (r >= n_biedges) return;
    const unsigned long long p = best[r];
    if (p == ~0ull) return;
    best[r] = ~0ull;
    parent2[r] = (uint32_t)(p >> 32);  // r was a root when it was proposed for, and only this thread writes it
    const uint32_t x = (uint32_t)p;    // the passage whose out-dart this is joins its node's rotation (one bit per dart: 23 MB at 2^27,
    atomicOr(&selected[x >> 5], 1u << (x & 31u));  // resident in L2 for the rotation's lookups, where a word per dart was 0.75 GB to clear and to gather from)
}
__global__ __launch_bounds__(EB) void rotate_kernel(const uint32_t *mirror, uint64_t n_nodes, const uint32_t *row, const uint32_t *adj,
                                                   const uint32_t *selected, uint32_t *succ) {
    const uint64_t vi = gid();
    if (vi >= n_nodes) return;
    {   // (single-passage binodes: nothing can have been selected)
        const uint32_t v = (uint32_t)vi, vm = mirror[v], d = row[v + 1] - row[v];
        if (vm < v || (vm == v ? d / 2 : d) < 2) return;
    }
    uint32_t a_prev = 0, b0 = 0;
    bool any = false;
    for_each_passage(mirror, row, adj, (uint32_t)vi, [&](uint32_t i, uint32_t a, uint32_t b) {
        if (i == 0) {
            a_prev = a;
            b0 = b;
            return;
        }
        if (!bit_of(selected, b)) return;
        succ[a_prev] = b;
        succ[b ^ 1] = a_prev ^ 1;
        a_prev = a;
        any = true;
    });
    if (any) {
        succ[a_prev] = b0;
        succ[b0 ^ 1] = a_prev ^ 1;
    }
}

// ---- step 5: ranking ---------------------------------------------------------------------------------------------

// Splitters (round 4, second form): every 64th dart id (dart ids are unrelated to the order of the trails, so these are as evenly
// spread as a hash) and the smallest dart of every connected component ("root": the closed walk starts there, and only the trail
// that carries it is emitted). Splitter i < M0 = ceil(E / 64) is dart 64 i; the roots that are not a multiple of 64 follow, in
// ascending order (`extra`, X of them) -- no per-dart flag array, no scan and no compaction over the darts, and the index of a
// splitter is arithmetic (extras: a binary search in their short list). Biedge b is a root iff it is the label of its own trail pair
// and of its own component: comp[b] == b and parent2[b] == b (both arrays are flat here) -- two streaming passes over the biedges:
//   root_count_kernel  extras per workgroup chunk (-> scan over the chunk counts) + the root bitmap (one bit per dart)
//   root_write_kernel  the extras' darts, ranked inside the chunk
constexpr int ROOT_PER = 8, ROOT_CHUNK = EB * ROOT_PER;
__device__ __forceinline__ bool is_root_biedge(const uint32_t *comp, const uint32_t *parent2, uint64_t b) { return comp[b] == (uint32_t)b && parent2[b] == (uint32_t)b; }
__global__ __launch_bounds__(EB) void root_count_kernel(const uint32_t *comp, const uint32_t *parent2, uint64_t n_biedges, uint32_t *chunk_extras,
                                                       uint32_t *root_bits) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int p = 0; p < ROOT_PER; p++) {
        const uint64_t b = (uint64_t)blockIdx.x * ROOT_CHUNK + (uint64_t)p * EB + threadIdx.x;
        if (b < n_biedges && is_root_biedge(comp, parent2, b)) {
            const uint64_t e = 2 * b;
            atomicOr(&root_bits[e >> 5], 1u << (e & 31));  // (a handful per graph: one per connected component)
            c += (e & 63u) ? 1u : 0u;
        }
    }
    for (int dd = 32; dd >= 1; dd >>= 1) c += __shfl_down(c, dd);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) chunk_extras[blockIdx.x] = s_cnt;
}
__global__ __launch_bounds__(EB) void root_write_kernel(const uint32_t *comp, const uint32_t *parent2, uint64_t n_biedges, const uint32_t *chunk_offsets,
                                                       uint32_t *extra) {
    __shared__ uint32_t wave_cnt[ROOT_PER][EB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long bal[ROOT_PER];
#pragma unroll
    for (int p = 0; p < ROOT_PER; p++) {
        const uint64_t b = (uint64_t)blockIdx.x * ROOT_CHUNK + (uint64_t)p * EB + threadIdx.x;
        bal[p] = __ballot(b < n_biedges && ((2 * b) & 63u) && is_root_biedge(comp, parent2, b));
        if (lane == 0) wave_cnt[p][wv] = (uint32_t)__popcll(bal[p]);
    }
    __syncthreads();
    uint32_t off = chunk_offsets[blockIdx.x];
#pragma unroll
    for (int p = 0; p < ROOT_PER; p++) {  // ascending: lane, wave, pass, chunk
        for (int j = 0; j < EB / 64; j++) {
            if (j == wv && ((bal[p] >> lane) & 1ull))
                extra[off + (uint32_t)__popcll(bal[p] & ((1ull << lane) - 1ull))] = (uint32_t)(2 * ((uint64_t)blockIdx.x * ROOT_CHUNK + (uint64_t)p * EB + threadIdx.x));
            off += wave_cnt[p][j];
        }
    }
}
// the splitters' darts, by index (M0 regular ones, then the extras)
__global__ __launch_bounds__(EB) void split_list_kernel(uint32_t m0, const uint32_t *extra, uint32_t n_split, uint32_t *splitters) {
    const uint64_t i = gid();
    if (i < n_split) splitters[i] = i < m0 ? (uint32_t)(i * 64) : extra[i - m0];
}
struct SplitIndex {  // dart -> index of the splitter it is
    uint32_t m0, n_extra;
    const uint32_t *extra;
    __device__ __forceinline__ uint32_t operator()(uint32_t x) const {
        if (!(x & 63u)) return x >> 6;
        uint32_t lo = 0, hi = n_extra;  // (x is an extra root: present)
        while (lo + 1 < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (extra[mid] <= x) lo = mid; else hi = mid;
        }
        return m0 + lo;
    }
};
// While dart ids leave bit 31 free (fewer than 2^31 darts) the walks do not look the splitter bitmap up at every step: the
// PREDECESSOR of every splitter carries the mark in bit 31 of its successor word -- pred(x) = succ[x ^ 1] ^ 1, by the mirror symmetry
// of the pairing -- set by one thread per splitter. One random load per step instead of two (walk_measure 6.8 -> see DESIGN 4.7 / docs/history).
constexpr uint32_t SUCC_MARK = 0x80000000u;
__global__ __launch_bounds__(EB) void mark_pred_kernel(uint32_t *succ, const uint32_t *splitters, uint32_t n_split) {
    const uint64_t i = gid();
    if (i >= n_split) return;
    const uint32_t x = splitters[i];
    const uint32_t p = (succ[x ^ 1u] & ~SUCC_MARK) ^ 1u;  // (the word of x ^ 1 may carry a mark of its own already)
    atomicOr(&succ[p], SUCC_MARK);
}
// The measuring walk also RECORDS the darts it passes, so that no second chase through `succ` is needed to write the closed walks
// (round 4; until then walk_write_kernel walked every emitted segment again: E / 2 more dependent random loads). The 64 walkers of
// a wave advance in lockstep; in every step the darts of the lanes that are still walking are stored back to back (ballot rank) in
// the wave's current chunk of the sequence buffer -- coalesced stores, exactly one word per dart. Chunks of SEQ_CHUNK words come
// from a bump cursor (one atomic per chunk and wave), their numbers go to the wave's row of chunk_tab; seq_copy_kernel replays the
// same arithmetic from the segment lengths alone (which lanes are active in step j, where the row starts) and moves every emitted
// dart to its final place. Where a chunk lands depends on thread timing, the result does not.
constexpr uint32_t SEQ_CHUNK = 1024;  // words
constexpr int SEQ_MAXC = 48;          // chunks per wave (a wave's 64 segments are ~4096 darts on average: 4-5 chunks)
__host__ __device__ inline uint64_t seq_capacity_chunks(uint64_t n_darts, uint64_t n_waves) { return n_darts / (SEQ_CHUNK - 64) + n_waves + 16; }
template <bool MARKED, bool RECORD>
__global__ __launch_bounds__(EB) void walk_measure_kernel(const uint32_t *succ, const uint32_t *splitters, SplitIndex sidx,
                                                         const uint32_t *root_bits, uint32_t n_split,
                                                         uint64_t n_darts, uint32_t *seg_len, uint32_t *next_split, uint32_t *jump,
                                                         uint32_t *dist, uint32_t *error, uint32_t *seq, unsigned long long *seq_cursor,
                                                         uint64_t seq_chunks, uint32_t *chunk_tab, uint32_t max_chunks) {
    const uint64_t i = gid();
    const bool valid = i < n_split;
    const uint32_t s = valid ? splitters[i] : 0u;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = i >> 6;
    uint32_t cur = s, x = 0, len = 0;
    bool act = valid;
    uint32_t off = 0, end = 0, n_chunks = 0;  // wave-uniform: fill of the wave's current chunk
    uint64_t base = 0;
    for (;;) {
        const unsigned long long m = __ballot(act);
        if (!m) break;
        if constexpr (RECORD) {
            const uint32_t n_act = (uint32_t)__popcll(m);
            if (off + n_act > end) {  // (wave-uniform) a new chunk
                unsigned long long c = 0;
                if (lane == 0) {
                    c = atomicAdd(seq_cursor, 1ull);
                    if (n_chunks >= max_chunks || c >= seq_chunks) { atomicOr(error, 16u); c = 0; }  // (the caller falls back to the second walk)
                    else chunk_tab[wave * SEQ_MAXC + n_chunks] = (uint32_t)c;
                }
                c = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(c >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)c);
                base = c * SEQ_CHUNK;
                n_chunks++;
                off = 0;
                end = SEQ_CHUNK;
            }
            if (act) seq[base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = cur;
            off += n_act;
        }
        if (act) {
            const uint32_t w = succ[cur];
            const uint32_t nx = MARKED ? (w & ~SUCC_MARK) : w;
            const bool stop = MARKED ? (w & SUCC_MARK) != 0u : (!(w & 63u) || bit_of(root_bits, w));
            if (++len == 0 || len > n_darts) {  // cannot happen for a permutation; guards against a corrupted successor array
                atomicOr(error, 2u);
                act = false;
            }
            if (stop) { x = nx; act = false; }
            else cur = nx;
        }
    }
    if (!valid) return;
    const uint32_t nxt = sidx(x);
    seg_len[i] = len;
    next_split[i] = nxt;
    const bool root = bit_of(root_bits, s);
    jump[i] = root ? (uint32_t)i : nxt;  // roots are the terminals of the reduced lists
    dist[i] = root ? 0u : len;
}
// every emitted segment from the recorded sequence to its place in the closed walk (see walk_measure_kernel)
__global__ __launch_bounds__(EB) void seq_copy_kernel(const uint32_t *seq, const uint32_t *chunk_tab, const uint32_t *rflag, const uint32_t *ridx,
                                                     const uint32_t *jump, const uint32_t *dist, const uint32_t *seg_len, const uint32_t *cyc_len,
                                                     const uint32_t *cyc_base, uint32_t n_split, uint32_t *out) {
    const uint64_t i = gid();
    const bool valid = i < n_split;
    const uint64_t wave = i >> 6;
    const uint32_t len = valid ? seg_len[i] : 0u;
    bool emit = false;
    uint32_t p = 0;
    if (valid) {
        const uint32_t t = jump[i];
        emit = rflag[t] != 0u;  // (a mirror trail has no root on it: not emitted)
        if (emit) {
            const uint32_t r = ridx[t];
            p = cyc_base[r] + (i == t ? 0u : cyc_len[r] - dist[i]);
        }
    }
    uint32_t off = 0, end = 0, n_chunks = 0;
    uint64_t base = 0;
    // (a wave runs as many steps as its longest segment, ~300 for a mean of 64: the loop is bound by the latency of its loads, so four
    // rows are requested before the first is stored)
    constexpr int U = 4;
    for (uint32_t j0 = 0;; j0 += U) {
        uint32_t v[U];
        bool wr[U];
        bool done = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool act = len > j0 + u;
            const unsigned long long m = __ballot(act);
            wr[u] = false;
            v[u] = 0;
            if (!m) { done = true; continue; }  // (lengths only shrink the active set: once empty, empty for good)
            const uint32_t n_act = (uint32_t)__popcll(m);
            if (off + n_act > end) {
                base = (uint64_t)chunk_tab[wave * SEQ_MAXC + n_chunks] * SEQ_CHUNK;
                n_chunks++;
                off = 0;
                end = SEQ_CHUNK;
            }
            wr[u] = act && emit;
            if (wr[u]) v[u] = seq[base + off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))];
            off += n_act;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (wr[u]) out[p + j0 + u] = v[u];
        if (done) break;
    }
}
__global__ __launch_bounds__(EB) void wyllie_kernel(const uint32_t *jump_in, const uint32_t *dist_in, uint32_t n, uint32_t *jump_out,
                                                   uint32_t *dist_out) {
    const uint64_t i = gid();
    if (i >= n) return;
    const uint32_t j = jump_in[i];
    jump_out[i] = jump_in[j];
    dist_out[i] = dist_in[i] + dist_in[j];
}
// ---- two-level ranking of the reduced list (round 4) ----
// Pointer jumping over all M splitters takes ceil(log2 2M) launches of a latency-bound kernel (23 x 0.09 ms at 2^27). Instead:
// every 32nd splitter (by hash) and every root is a LEVEL-2 splitter; one walker per level-2 splitter sums the segment lengths
// up to the next one (32 dependent steps over arrays that sit in L2), pointer jumping runs over the ~M/32 level-2 elements only,
// and a second walk hands every splitter of a level-2 segment its terminal root and its distance to it.
__device__ __forceinline__ bool hash_splitter2(uint32_t i) { return ((i * 0x85EBCA77u) >> 27) == 0; }
__global__ __launch_bounds__(EB) void split2_flag_kernel(const uint32_t *rflag, uint32_t n_split, uint32_t *flag2) {
    const uint64_t i = gid();
    if (i < n_split) flag2[i] = (rflag[i] || hash_splitter2((uint32_t)i)) ? 1u : 0u;
}
__global__ __launch_bounds__(EB) void split2_compact_kernel(const uint32_t *flag2, const uint32_t *idx2, uint32_t n_split, uint32_t *list2) {
    const uint64_t i = gid();
    if (i < n_split && flag2[i]) list2[idx2[i]] = (uint32_t)i;
}
// per level-2 splitter s (index a in list2): total length of its level-2 segment and the next level-2 splitter; a root is the
// terminal of its list (jump to itself, distance 0) but still owns the segment that follows it
__global__ __launch_bounds__(EB) void walk2_measure_kernel(const uint32_t *list2, uint32_t n2, const uint32_t *flag2, const uint32_t *idx2,
                                                          const uint32_t *rflag, const uint32_t *next_split, const uint32_t *seg_len,
                                                          uint32_t n_split, uint32_t *seg2_len, uint32_t *next2, uint32_t *jump2, uint32_t *dist2,
                                                          uint32_t *error) {
    const uint64_t a = gid();
    if (a >= n2) return;
    const uint32_t s = list2[a];
    uint32_t x = s, total = 0, steps = 0;
    do {
        total += seg_len[x];
        x = next_split[x];
        if (++steps > n_split) { atomicOr(error, 2u); break; }  // (cannot happen: the reduced list is a permutation)
    } while (!flag2[x]);
    const uint32_t nx = idx2[x];
    seg2_len[a] = total;
    next2[a] = nx;
    const bool root = rflag[s] != 0;
    jump2[a] = root ? (uint32_t)a : nx;
    dist2[a] = root ? 0u : total;
}
// every splitter of the level-2 segment of s: its terminal root (as a splitter index) and its distance to it
__global__ __launch_bounds__(EB) void walk2_write_kernel(const uint32_t *list2, uint32_t n2, const uint32_t *flag2, const uint32_t *rflag,
                                                        const uint32_t *next_split, const uint32_t *seg_len, const uint32_t *seg2_len,
                                                        const uint32_t *next2, const uint32_t *jump2, const uint32_t *dist2, uint32_t *jump,
                                                        uint32_t *dist) {
    const uint64_t a = gid();
    if (a >= n2) return;
    const uint32_t s = list2[a];
    const uint32_t t2 = next2[a];                        // the level-2 splitter that ends this segment
    const uint32_t term = list2[jump2[t2]];              // ... its terminal (t2 itself when it is a root), as a splitter index
    const uint32_t tail = dist2[t2];                     // ... and its distance to it (0 for a root)
    uint32_t x = s, before = 0;
    const uint32_t total = seg2_len[a];
    do {
        const bool root = rflag[x] != 0;                 // (only x == s can be a root: roots are level-2 splitters)
        jump[x] = root ? x : term;
        dist[x] = root ? 0u : total - before + tail;
        before += seg_len[x];
        x = next_split[x];
    } while (!flag2[x]);
}

// per root splitter (ascending dart id): length of its trail
__global__ __launch_bounds__(EB) void root_flag_kernel(const uint32_t *splitters, const uint32_t *root_bits, uint32_t n_split, uint32_t *rflag) {
    const uint64_t i = gid();
    if (i < n_split) rflag[i] = bit_of(root_bits, splitters[i]) ? 1u : 0u;
}
__global__ __launch_bounds__(EB) void root_len_kernel(const uint32_t *rflag, const uint32_t *ridx, const uint32_t *seg_len,
                                                     const uint32_t *next_split, const uint32_t *dist, uint32_t n_split, uint32_t *cyc_len) {
    const uint64_t i = gid();
    if (i >= n_split || !rflag[i]) return;
    cyc_len[ridx[i]] = seg_len[i] + dist[next_split[i]];  // next == i (single splitter): dist 0
}
template <bool MARKED>
__global__ __launch_bounds__(EB) void walk_write_kernel(const uint32_t *succ, const uint32_t *splitters, const uint32_t *rflag,
                                                       const uint32_t *ridx, const uint32_t *jump, const uint32_t *dist,
                                                       const uint32_t *seg_len, const uint32_t *cyc_len, const uint32_t *cyc_base,
                                                       uint32_t n_split, uint32_t *out) {
    const uint64_t i = gid();
    if (i >= n_split) return;
    const uint32_t t = jump[i];
    if (!rflag[t]) return;  // mirror trail (no root on it): not emitted
    const uint32_t r = ridx[t];
    uint32_t p = cyc_base[r] + (i == t ? 0u : cyc_len[r] - dist[i]);
    uint32_t x = splitters[i];
    const uint32_t len = seg_len[i];
    for (uint32_t j = 0; j < len; j++) {
        out[p++] = x;
        x = MARKED ? (succ[x] & ~SUCC_MARK) : succ[x];
    }
}

}  // namespace

// (tests: the bitmap form of the splitter test, which graphs with 2^31 darts or more always use, on small graphs too)
static std::atomic<int> g_force_bitmap{0}, g_flat_ranking{0}, g_second_walk{0}, g_small_tables{0};
// (bit 0: bitmap form of the splitter test; bit 1: pointer jumping over all splitters; bit 2: write the closed walks by a second
// walk through the successor array instead of from the sequence the measuring walk recorded; bit 3: one-entry chunk tables, so
// that every wave of a large graph outgrows its table and the fallback to the second walk runs)
void device_euler_force_bitmap(int on) {
    g_force_bitmap.store((on & 1) ? 1 : 0); g_flat_ranking.store((on & 2) ? 1 : 0); g_second_walk.store((on & 4) ? 1 : 0);
    g_small_tables.store((on & 8) ? 1 : 0);
}

// adj[row[v] + i] = i-th out-dart of v in ascending dart id; pos[e] = slot of e in its bucket (pos may be null)
// The buckets of darts [0, E) from the kept buckets of the original darts [0, E0) (row0 / adj0) and fresh ones of the dummy darts
// [E0, E): every dummy id is larger than every original one, so a node's bucket is its original darts followed by its dummy darts.
// Bucketing 56 M dummy darts and one streaming merge instead of bucketing 186 M darts (2^27).
namespace {
__global__ __launch_bounds__(EB) void degree_rank_offset_kernel(const uint32_t *from, uint64_t first, uint64_t n, uint32_t *deg, uint32_t *rank) {
    const uint64_t i = gid();
    if (i < n) rank[i] = atomicAdd(&deg[from[first + i]], 1u);
}
__global__ __launch_bounds__(EB) void fill_offset_kernel(const uint32_t *from, uint64_t first, uint64_t n, const uint32_t *row, const uint32_t *rank,
                                                        uint32_t *adj) {
    const uint64_t i = gid();
    if (i < n) adj[row[from[first + i]] + rank[i]] = (uint32_t)(first + i);
}
__global__ __launch_bounds__(EB) void merge_buckets_kernel(uint64_t n_nodes, const uint32_t *row0, const uint32_t *adj0, const uint32_t *row_d,
                                                          const uint32_t *adj_d, uint32_t *row, uint32_t *adj) {
    const uint64_t v = gid();
    if (v > n_nodes) return;
    const uint32_t lo0 = row0[v], lod = row_d[v];
    row[v] = lo0 + lod;
    if (v == n_nodes) return;
    const uint32_t d0 = row0[v + 1] - lo0, dd = row_d[v + 1] - lod;
    uint32_t o = lo0 + lod;
    for (uint32_t j = 0; j < d0; j++) adj[o++] = adj0[lo0 + j];
    for (uint32_t j = 0; j < dd; j++) adj[o++] = adj_d[lod + j];
}
}  // namespace
// ---- the arithmetic part of the dummy buckets (ZipBuckets, finish_device.hpp) ----
namespace {
struct ZipRange { uint32_t a_lo, a_hi, b_lo, b_hi; };  // steps whose even (a) / odd (b) dart leaves the node
__device__ __forceinline__ ZipRange zip_range(const ZipBuckets &z, const uint32_t *mirror, uint32_t v) {
    ZipRange r{0u, 0u, 0u, 0u};
    const uint32_t co = z.cout[v];
    if (co) {  // a_node holds v at [N - p_out[v] - cout[v], + cout[v]) (out-nodes descending)
        const int64_t a0 = (int64_t)z.n_units - (int64_t)z.p_out[v] - (int64_t)co;
        const int64_t lo = a0 > 0 ? a0 : 0, hi = a0 + co < (int64_t)z.n_steps ? a0 + co : (int64_t)z.n_steps;
        if (hi > lo) { r.a_lo = (uint32_t)lo; r.a_hi = (uint32_t)hi; }
    }
    const uint32_t t = mirror[v], ci = z.cin[t];
    if (ci) {  // b_node holds t at [p_in[t], + cin[t]) (in-nodes ascending); step s reads b_node[s + delta]
        const int64_t b0 = (int64_t)z.p_in[t] - (int64_t)z.delta;
        const int64_t lo = b0 > 0 ? b0 : 0, hi = b0 + ci < (int64_t)z.n_steps ? b0 + ci : (int64_t)z.n_steps;
        if (hi > lo) { r.b_lo = (uint32_t)lo; r.b_hi = (uint32_t)hi; }
    }
    return r;
}
__global__ __launch_bounds__(EB) void zip_degree_kernel(uint64_t n_nodes, ZipBuckets z, const uint32_t *mirror, uint32_t *zdeg) {
    const uint64_t v = gid();
    if (v > n_nodes) return;
    uint32_t d = 0;
    if (v < n_nodes) {
        const ZipRange r = zip_range(z, mirror, (uint32_t)v);
        d = (r.a_hi - r.a_lo) + (r.b_hi - r.b_lo);
    }
    zdeg[v] = d;
}
// bucket of v = its original darts, its bucketed dummy darts below the zip region (matched pairs, self-mirror phase), its zip darts
// (two arithmetic runs merged by id), its bucketed dummy darts above it (the sequential tail): ascending dart id throughout
__global__ __launch_bounds__(EB) void merge_zip_buckets_kernel(uint64_t n_nodes, const uint32_t *row0, const uint32_t *adj0, const uint32_t *row_g,
                                                              const uint32_t *adj_g, const uint32_t *row_z, ZipBuckets z, const uint32_t *mirror,
                                                              uint32_t *row, uint32_t *adj) {
    const uint64_t v = gid();
    if (v > n_nodes) return;
    const uint32_t lo0 = row0[v], log_ = row_g[v];
    uint32_t o = lo0 + log_ + row_z[v];
    row[v] = o;
    if (v == n_nodes) return;
    const uint32_t d0 = row0[v + 1] - lo0, dg = row_g[v + 1] - log_;
    for (uint32_t j = 0; j < d0; j++) adj[o++] = adj0[lo0 + j];
    uint32_t j = 0;
    for (; j < dg; j++) {
        const uint32_t e = adj_g[log_ + j];
        if ((uint64_t)e >= z.zip_first) break;
        adj[o++] = e;
    }
    const ZipRange r = zip_range(z, mirror, (uint32_t)v);
    uint32_t ia = r.a_lo, ib = r.b_lo;
    while (ia < r.a_hi || ib < r.b_hi) {
        // even dart of step ia against odd dart of step ib: 2 ia < 2 ib + 1  <=>  ia <= ib
        const bool take_a = ia < r.a_hi && (ib >= r.b_hi || ia <= ib);
        adj[o++] = (uint32_t)(z.zip_first + 2ull * (take_a ? ia : ib) + (take_a ? 0u : 1u));
        if (take_a) ia++; else ib++;
    }
    for (; j < dg; j++) adj[o++] = adj_g[log_ + j];
}
}  // namespace
void device_build_buckets_merged(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E0, uint64_t E, uint64_t V,
                                 const uint32_t *d_row0, const uint32_t *d_adj0, uint32_t *d_row, uint32_t *d_adj, const ZipBuckets *zip) {
    const bool use_zip = zip && zip->n_steps > 0;
    // the dummy darts that are bucketed with atomics: all of them, or -- beside the arithmetic zip region -- those below and above it
    const uint64_t zip_lo = use_zip ? zip->zip_first : E, zip_hi = use_zip ? zip->zip_first + 2ull * zip->n_steps : E;
    const uint64_t n1 = zip_lo - E0, n2 = E - zip_hi, n_d = n1 + n2;
    Buf b_rowd, b_adjd, b_rank, b_bsum, b_tot, b_rowz;
    uint32_t *d_rowd = b_rowd.alloc<uint32_t>(st, V + 1), *d_adjd = b_adjd.alloc<uint32_t>(st, std::max<uint64_t>(n_d, 1)),
             *d_rank = b_rank.alloc<uint32_t>(st, std::max<uint64_t>(n_d, 1));
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(V + 1) + 1), *d_tot = b_tot.alloc<uint32_t>(st, 1);
    HIP_CHECK(hipMemsetAsync(d_rowd, 0, (V + 1) * 4, st));
    if (n1) degree_rank_offset_kernel<<<grid_for(n1), EB, 0, st>>>(d_from, E0, n1, d_rowd, d_rank);
    if (n2) degree_rank_offset_kernel<<<grid_for(n2), EB, 0, st>>>(d_from, zip_hi, n2, d_rowd, d_rank + n1);
    scan_u32<uint32_t>(st, d_rowd, V + 1, d_rowd, d_bsum, d_tot);
    if (n1) fill_offset_kernel<<<grid_for(n1), EB, 0, st>>>(d_from, E0, n1, d_rowd, d_rank, d_adjd);
    if (n2) fill_offset_kernel<<<grid_for(n2), EB, 0, st>>>(d_from, zip_hi, n2, d_rowd, d_rank + n1, d_adjd);
    sort_buckets_kernel<<<grid_for(V), EB, 0, st>>>(V, d_rowd, d_adjd, nullptr);
    if (use_zip) {
        uint32_t *d_rowz = b_rowz.alloc<uint32_t>(st, V + 1);
        zip_degree_kernel<<<grid_for(V + 1), EB, 0, st>>>(V, *zip, d_mirror, d_rowz);
        scan_u32<uint32_t>(st, d_rowz, V + 1, d_rowz, d_bsum, d_tot);
        merge_zip_buckets_kernel<<<grid_for(V + 1), EB, 0, st>>>(V, d_row0, d_adj0, d_rowd, d_adjd, d_rowz, *zip, d_mirror, d_row, d_adj);
    } else {
        merge_buckets_kernel<<<grid_for(V + 1), EB, 0, st>>>(V, d_row0, d_adj0, d_rowd, d_adjd, d_row, d_adj);
    }
    HIP_CHECK(hipGetLastError());
}

void device_pairing(hipStream_t st, const uint32_t *d_mirror, uint64_t V, const uint32_t *d_row, const uint32_t *d_adj, uint32_t *d_succ, uint32_t *d_error) {
    succ_node_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_succ, d_error);
    HIP_CHECK(hipGetLastError());
}

// (d_scratch: E words of scratch, or null)
void device_build_buckets(hipStream_t st, const uint32_t *d_from, uint64_t E, uint64_t V, uint32_t *d_row, uint32_t *d_adj, uint32_t *d_pos,
                          uint32_t *d_scratch) {
    Buf b_rank, b_bsum, b_tot;
    uint32_t *d_rank = d_scratch ? d_scratch : b_rank.alloc<uint32_t>(st, E);
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(V + 1) + 1);
    uint32_t *d_tot = b_tot.alloc<uint32_t>(st, 1);
    HIP_CHECK(hipMemsetAsync(d_row, 0, (V + 1) * 4, st));
    degree_rank_kernel<<<grid_for(E), EB, 0, st>>>(d_from, E, d_row, d_rank);
    scan_u32<uint32_t>(st, d_row, V + 1, d_row, d_bsum, d_tot);
    fill_kernel<<<grid_for(E), EB, 0, st>>>(d_from, E, d_row, d_rank, d_adj);
    sort_buckets_kernel<<<grid_for(V), EB, 0, st>>>(V, d_row, d_adj, d_pos);
    HIP_CHECK(hipGetLastError());
}

// The decomposition over device arrays: from[E] (dart e leaves from[e]; its mirror is e ^ 1) and mirror[V]. Results: the closed
// walks back to back in b_out (u32[E / 2], dart ids), their lengths in b_clen and start offsets in b_cbase (u32[*n_cycles]).
void device_euler_decompose(hipStream_t st, const uint32_t *d_from, const uint32_t *d_mirror, uint64_t E, uint64_t V, Buf &b_out, Buf &b_clen,
                            Buf &b_cbase, uint32_t *n_cycles, double *kernel_ms_out, const uint32_t *d_row0, const uint32_t *d_adj0, uint64_t E0,
                            const ZipBuckets *zip) {
    if (E == 0 || (E & 1) || E >= 0xFFFFFFFFull) MTG_DIE("device_euler_decompose: %llu darts (must be even and fit 32-bit ids)", (unsigned long long)E);
    hipEvent_t ev0, ev1;
    HIP_CHECK(hipEventCreate(&ev0));
    HIP_CHECK(hipEventCreate(&ev1));
    HIP_CHECK(hipEventRecord(ev0, st));

    Buf b_row, b_adj, b_pos2, b_succ, b_comp, b_flag, b_rbits, b_best, b_bsum, b_small, b_active;
    uint32_t *d_row = b_row.alloc<uint32_t>(st, V + 1);
    uint32_t *d_adj = b_adj.alloc<uint32_t>(st, E);
    const uint64_t n_b = E / 2;  // biedges
    // (scratch of the bucket build over all darts when there are no kept buckets, then the second union-find, over biedges)
    uint32_t *d_pos2 = b_pos2.alloc<uint32_t>(st, d_row0 ? n_b : E);
    uint32_t *d_succ = b_succ.alloc<uint32_t>(st, E);
    uint32_t *d_comp = b_comp.alloc<uint32_t>(st, n_b);  // union-find over biedges -> labels of the trail pairs, in place
    const uint64_t n_words = (E + 31) / 32;
    uint32_t *d_flag = b_flag.alloc<uint32_t>(st, n_words);  // `selected`: one bit per dart
    uint32_t *d_rbits = b_rbits.alloc<uint32_t>(st, n_words);
    unsigned long long *d_best = b_best.alloc<unsigned long long>(st, n_b);
    uint8_t *d_active = b_active.alloc<uint8_t>(st, V);
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(E) + 2);
    uint32_t *d_small = b_small.alloc<uint32_t>(st, 8);  // [0] error, [1..3] scan totals, [4] a binode proposed in this round, [5] level-2 splitters
    uint32_t *d_error = d_small, *d_total = d_small + 1;
    HIP_CHECK(hipMemsetAsync(d_small, 0, 32, st));
    static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        HIP_CHECK(hipStreamSynchronize(st));
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] euler decompose:   %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t_lap).count());
        t_lap = n;
    };
    lap("allocations");

    // 1. buckets
    if (d_row0) device_build_buckets_merged(st, d_from, d_mirror, E0, E, V, d_row0, d_adj0, d_row, d_adj, zip);
    else device_build_buckets(st, d_from, E, V, d_row, d_adj, nullptr, d_pos2);  // (d_pos2 is free until the trail labels)
    lap("buckets");
    // 2. pairing, 3. trail labels
    succ_node_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_succ, d_error);
    uint32_t h_small[8];
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[0]) MTG_DIE("device_euler_cycles: the graph is not Eulerian (greedytigs/mod.rs:708)");
    lap("pairing");
    label_init_kernel<<<grid_for(n_b), EB, 0, st>>>(d_succ, n_b, d_comp);
    label_hook_kernel<<<grid_for(n_b), EB, 0, st>>>(d_succ, n_b, d_comp);
    uint32_t *d_parent2 = d_pos2;
    flatten_kernel<<<grid_for(n_b), EB, 0, st>>>(d_comp, n_b, d_parent2);
    lap("trail labels");
    // 4. merge the trails of every connected component
    HIP_CHECK(hipMemsetAsync(d_flag, 0, n_words * 4, st));  // `selected`
    HIP_CHECK(hipMemsetAsync(d_best, 0xFF, n_b * 8, st));
    int hook_rounds = 0;
    for (;; hook_rounds++) {  // (a round in which no binode sees two components is the last: nothing to hook, nothing to flatten)
        if (hook_rounds > 64) MTG_DIE("device_euler_cycles: internal error (component hooking does not converge)");
        HIP_CHECK(hipMemsetAsync(d_small + 4, 0, 4, st));
        propose_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_comp, d_parent2, d_best, d_small + 4, d_active, hook_rounds == 0 ? 1 : 0);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (!h_small[4]) break;
        hook_kernel<<<grid_for(n_b), EB, 0, st>>>(n_b, d_parent2, d_best, d_flag);
        flatten_labels_kernel<<<grid_for(n_b), EB, 0, st>>>(d_comp, d_parent2, n_b);
    }
    lap("hooking rounds");
    b_best.release();
    b_active.release();
    rotate_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_flag, d_succ);
    // 5. ranking: splitters = every 64th dart + the components' roots (see root_count_kernel)
    const uint64_t n_root_chunks = (n_b + ROOT_CHUNK - 1) / ROOT_CHUNK;
    Buf b_xcnt, b_extra;
    uint32_t *d_xcnt = b_xcnt.alloc<uint32_t>(st, n_root_chunks);
    HIP_CHECK(hipMemsetAsync(d_rbits, 0, n_words * 4, st));
    root_count_kernel<<<(unsigned)n_root_chunks, EB, 0, st>>>(d_comp, d_parent2, n_b, d_xcnt, d_rbits);
    scan_u32<uint32_t>(st, d_xcnt, n_root_chunks, d_xcnt, d_bsum, d_total);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t M0 = (E + 63) / 64, X = h_small[1];  // regular splitters, extra roots
    if (M0 + X >= 0xFFFFFFFFull) MTG_DIE("device_euler_cycles: too many splitters");
    const uint32_t M = (uint32_t)(M0 + X);
    uint32_t *d_extra = b_extra.alloc<uint32_t>(st, std::max<uint64_t>(X, 1));
    if (X) root_write_kernel<<<(unsigned)n_root_chunks, EB, 0, st>>>(d_comp, d_parent2, n_b, d_xcnt, d_extra);
    HIP_CHECK(hipGetLastError());
    lap("rotate + roots");
    b_row.release();
    b_adj.release();
    b_flag.release();
    b_comp.release();
    b_pos2.release();
    b_xcnt.release();

    Buf b_split, b_seglen, b_next, b_jump0, b_dist0, b_jump1, b_dist1, b_rflag, b_ridx;
    uint32_t *d_split = b_split.alloc<uint32_t>(st, M);
    uint32_t *d_seglen = b_seglen.alloc<uint32_t>(st, M);
    uint32_t *d_next = b_next.alloc<uint32_t>(st, M);
    uint32_t *d_jump[2] = {b_jump0.alloc<uint32_t>(st, M), b_jump1.alloc<uint32_t>(st, M)};
    uint32_t *d_dist[2] = {b_dist0.alloc<uint32_t>(st, M), b_dist1.alloc<uint32_t>(st, M)};
    uint32_t *d_rflag = b_rflag.alloc<uint32_t>(st, M);
    uint32_t *d_ridx = b_ridx.alloc<uint32_t>(st, M);
    split_list_kernel<<<grid_for(M), EB, 0, st>>>((uint32_t)M0, d_extra, M, d_split);
    const SplitIndex sidx{(uint32_t)M0, (uint32_t)X, d_extra};
    const bool marked = E < 0x80000000ull && !g_force_bitmap;  // bit 31 of a successor word is free: it marks "my successor is a splitter"
    // the measuring walk records the darts it passes (see walk_measure_kernel): sequence buffer, its chunk cursor, the waves' chunk tables
    const bool record = !g_second_walk;
    const uint64_t n_waves = ((uint64_t)M + 63) / 64, seq_chunks = record ? seq_capacity_chunks(E, n_waves) : 0;
    Buf b_seq, b_ctab, b_cursor;
    uint32_t *d_seq = record ? b_seq.alloc<uint32_t>(st, seq_chunks * SEQ_CHUNK) : nullptr;
    uint32_t *d_ctab = record ? b_ctab.alloc<uint32_t>(st, n_waves * SEQ_MAXC) : nullptr;
    unsigned long long *d_cursor = b_cursor.alloc<unsigned long long>(st, 1);
    HIP_CHECK(hipMemsetAsync(d_cursor, 0, 8, st));
    if (marked) mark_pred_kernel<<<grid_for(M), EB, 0, st>>>(d_succ, d_split, M);
    {
        auto fn = marked ? (record ? walk_measure_kernel<true, true> : walk_measure_kernel<true, false>)
                         : (record ? walk_measure_kernel<false, true> : walk_measure_kernel<false, false>);
        fn<<<grid_for(M), EB, 0, st>>>(d_succ, d_split, sidx, d_rbits, M, E, d_seglen, d_next, d_jump[0], d_dist[0], d_error, d_seq, d_cursor,
                                       seq_chunks, d_ctab, g_small_tables ? 1u : (uint32_t)SEQ_MAXC);
    }
    int cur = 0;
    root_flag_kernel<<<grid_for(M), EB, 0, st>>>(d_split, d_rbits, M, d_rflag);
    scan_u32<uint32_t>(st, d_rflag, M, d_ridx, d_bsum, d_total + 1);
    if (g_flat_ranking || M < (1u << 16)) {  // pointer jumping over all splitters (small inputs; tests hold the two forms to the same walks)
        for (uint64_t span = 1; span < (uint64_t)M * 2; span <<= 1) {  // after r rounds a pointer spans 2^r reduced elements
            wyllie_kernel<<<grid_for(M), EB, 0, st>>>(d_jump[cur], d_dist[cur], M, d_jump[cur ^ 1], d_dist[cur ^ 1]);
            cur ^= 1;
        }
    } else {  // two levels (see split2_flag_kernel)
        Buf b_flag2, b_idx2, b_list2, b_s2len, b_next2, b_j2a, b_j2b, b_d2a, b_d2b;
        uint32_t *d_flag2 = b_flag2.alloc<uint32_t>(st, M), *d_idx2 = b_idx2.alloc<uint32_t>(st, M);
        split2_flag_kernel<<<grid_for(M), EB, 0, st>>>(d_rflag, M, d_flag2);
        scan_u32<uint32_t>(st, d_flag2, M, d_idx2, d_bsum, d_small + 5);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        const uint32_t M2 = h_small[5];
        uint32_t *d_list2 = b_list2.alloc<uint32_t>(st, M2), *d_s2len = b_s2len.alloc<uint32_t>(st, M2), *d_next2 = b_next2.alloc<uint32_t>(st, M2);
        uint32_t *d_j2[2] = {b_j2a.alloc<uint32_t>(st, M2), b_j2b.alloc<uint32_t>(st, M2)};
        uint32_t *d_d2[2] = {b_d2a.alloc<uint32_t>(st, M2), b_d2b.alloc<uint32_t>(st, M2)};
        if (M2) {
            split2_compact_kernel<<<grid_for(M), EB, 0, st>>>(d_flag2, d_idx2, M, d_list2);
            walk2_measure_kernel<<<grid_for(M2), EB, 0, st>>>(d_list2, M2, d_flag2, d_idx2, d_rflag, d_next, d_seglen, M, d_s2len, d_next2, d_j2[0], d_d2[0], d_error);
            int c2 = 0;
            for (uint64_t span = 1; span < (uint64_t)M2 * 2; span <<= 1) {
                wyllie_kernel<<<grid_for(M2), EB, 0, st>>>(d_j2[c2], d_d2[c2], M2, d_j2[c2 ^ 1], d_d2[c2 ^ 1]);
                c2 ^= 1;
            }
            // splitters on trails without a level-2 splitter (short mirror trails) keep the values walk_measure gave them: never a root
            walk2_write_kernel<<<grid_for(M2), EB, 0, st>>>(d_list2, M2, d_flag2, d_rflag, d_next, d_seglen, d_s2len, d_next2, d_j2[c2], d_d2[c2], d_jump[cur], d_dist[cur]);
        }
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipStreamSynchronize(st));  // (the level-2 arrays go back to the block cache when this scope ends)
    }
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[0] & ~16u) MTG_DIE("device_euler_cycles: internal error (successor array is not a permutation)");
    const bool from_sequence = record && !(h_small[0] & 16u);  // (16: a wave outgrew its chunk table -- the second walk writes instead)
    lap("segment walks + ranking");
    const uint32_t R = h_small[2];  // connected components = closed walks
    if (std::getenv("MTG_DEBUG")) std::fprintf(stderr, "[mtg] euler decompose: E %llu V %llu hook rounds %d splitters %u components %u\n", (unsigned long long)E, (unsigned long long)V, hook_rounds, M, R);
    b_extra.release();
    uint32_t *d_clen = b_clen.alloc<uint32_t>(st, R);
    uint32_t *d_cbase = b_cbase.alloc<uint32_t>(st, R);
    uint32_t *d_out = b_out.alloc<uint32_t>(st, E / 2);
    root_len_kernel<<<grid_for(M), EB, 0, st>>>(d_rflag, d_ridx, d_seglen, d_next, d_dist[cur], M, d_clen);
    scan_u32<uint32_t>(st, d_clen, R, d_cbase, d_bsum, d_total + 2);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[3] != E / 2)
        MTG_DIE("device_euler_cycles: internal error (closed walks cover %u of %llu biedges)", h_small[3], (unsigned long long)(E / 2));
    if (from_sequence) seq_copy_kernel<<<grid_for(M), EB, 0, st>>>(d_seq, d_ctab, d_rflag, d_ridx, d_jump[cur], d_dist[cur], d_seglen, d_clen, d_cbase, M, d_out);
    else if (marked) walk_write_kernel<true><<<grid_for(M), EB, 0, st>>>(d_succ, d_split, d_rflag, d_ridx, d_jump[cur], d_dist[cur], d_seglen, d_clen, d_cbase, M, d_out);
    else walk_write_kernel<false><<<grid_for(M), EB, 0, st>>>(d_succ, d_split, d_rflag, d_ridx, d_jump[cur], d_dist[cur], d_seglen, d_clen, d_cbase, M, d_out);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventRecord(ev1, st));
    HIP_CHECK(hipStreamSynchronize(st));
    lap("write walks");
    *n_cycles = R;
    if (kernel_ms_out) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        *kernel_ms_out = ms;
    }
    HIP_CHECK(hipEventDestroy(ev0));
    HIP_CHECK(hipEventDestroy(ev1));
}

// Host-graph form (mtg_euler_cycles_device): uploads from / mirror of the graph as it stands, downloads the closed walks.
Walks device_euler_cycles(const HostGraph &g, int device_id, double *kernel_ms_out) {
    const uint64_t E = g.edge_count(), V = g.node_count();
    Walks result;
    if (E == 0) return result;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device_id)
        MTG_DIE("device_euler_cycles: no MI355X device %d visible (there is no CPU fallback for this mode)", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    hipStream_t st = finish_stream(device_id);
    Buf b_from, b_mirror, b_out, b_clen, b_cbase;
    uint32_t *d_from = b_from.alloc<uint32_t>(st, E);
    uint32_t *d_mirror = b_mirror.alloc<uint32_t>(st, V);
    HIP_CHECK(hipMemcpyAsync(d_from, g.e_from.data(), E * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(d_mirror, g.mirror.data(), V * 4, hipMemcpyHostToDevice, st));
    uint32_t R = 0;
    device_euler_decompose(st, d_from, d_mirror, E, V, b_out, b_clen, b_cbase, &R, kernel_ms_out, nullptr, nullptr, 0);
    result.edges.resize(E / 2);
    std::vector<uint32_t> clen(R);
    HIP_CHECK(hipMemcpyAsync(result.edges.data(), b_out.as<uint32_t>(), (E / 2) * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(clen.data(), b_clen.as<uint32_t>(), (uint64_t)R * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    result.limits.resize(R);
    uint64_t run = 0;
    for (uint32_t r = 0; r < R; r++) {
        run += clen[r];
        result.limits[r] = run;
    }
    b_from.release(); b_mirror.release(); b_out.release(); b_clen.release(); b_cbase.release();
    HIP_CHECK(hipStreamSynchronize(st));
    finish_trim(device_id, E * 40);
    return result;
}

void device_warm_euler_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(degree_rank_kernel));
}

}  // namespace mtg
