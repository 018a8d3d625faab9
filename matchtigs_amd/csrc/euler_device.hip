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
//   3. label the trails (lock-free union-find over darts, root = smallest dart id); a biedge component is a
//      trail together with its mirror trail.
//   4. spanning forest over (biedge components x binodes): rounds of deterministic hooking -- every binode proposes,
//      for each passage whose component differs from its passage 0's, to hang the larger component root below the
//      smaller; a root accepts its smallest proposal (64-bit atomicMin) -- until every binode sees one component.
//      Each accepted proposal selects its passage; a node then rotates the successors of passage 0 and its selected
//      passages (and, mirrored, at the mirror node), which merges all their trails into one. Roots only hang below
//      smaller ids, so the accepted proposals form a forest: every one joins two trails that are still distinct, and
//      the result is exactly one trail pair per connected component.
//   5. rank the final trails: random splitters (1/64 of the darts + the smallest dart of every component) walk to
//      the next splitter, the reduced list is ranked by pointer jumping, and a second walk writes the darts at
//      their final positions. Only the trail that contains the component's smallest dart is emitted.
// No step depends on thread timing (atomics are only used for counts, minima and idempotent flags), so the same graph
// always gives the same walks. Everything is integer gather/scatter work; no MFMA.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <vector>

#include "device.hpp"

namespace mtg {

#define HIP_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) MTG_DIE("HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
    } while (0)

namespace {

constexpr int EB = 256;  // threads per block for the element-wise kernels
constexpr uint32_t SPLIT_FLAG = 0x80000000u;

inline unsigned grid_for(uint64_t n, int block = EB) { return (unsigned)((n + block - 1) / block); }

// ---- exclusive scan u32 -> u32, three phases, 2048 items per block -----------------------------------------
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = EB * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
    __shared__ uint32_t wave_sum[EB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int w = 0; w < EB / 64; w++) {
        if (w < wave) base += wave_sum[w];
        all += wave_sum[w];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

__global__ __launch_bounds__(EB) void scan_reduce_kernel(const uint32_t *in, uint64_t n, uint32_t *block_sums) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < n) s += in[base + i];
    uint32_t total;
    block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(EB) void scan_block_sums_kernel(uint32_t *block_sums, uint32_t n_blocks, uint32_t *total_out) {
    uint32_t carry = 0;
    for (uint32_t start = 0; start < n_blocks; start += EB) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, &total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(EB) void scan_apply_kernel(const uint32_t *in, uint64_t n, const uint32_t *block_offsets, uint32_t *out) {
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    uint32_t total;
    uint32_t run = block_offsets[blockIdx.x] + block_exclusive_scan(s, &total);
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

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

// ---- step 1: bucket darts by from-node ------------------------------------------------------------------------
__global__ __launch_bounds__(EB) void degree_kernel(const uint32_t *from, uint32_t n_darts, uint32_t *deg) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e < n_darts) atomicAdd(&deg[from[e]], 1u);
}
__global__ __launch_bounds__(EB) void fill_kernel(const uint32_t *from, uint32_t n_darts, const uint32_t *row, uint32_t *cursor,
                                                 uint32_t *adj, uint32_t *pos) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e >= n_darts) return;
    const uint32_t u = from[e];
    const uint32_t p = atomicAdd(&cursor[u], 1u);
    adj[row[u] + p] = e;
    pos[e] = p;
}

// buckets are filled in atomic order; sorting each node's few out-darts by id makes slots (and with them the whole
// decomposition) independent of thread timing
__global__ __launch_bounds__(EB) void sort_buckets_kernel(uint32_t n_nodes, const uint32_t *row, uint32_t *adj, uint32_t *pos) {
    const uint32_t v = blockIdx.x * EB + threadIdx.x;
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
    for (uint32_t i = lo; i < hi; i++) pos[adj[i]] = i - lo;
}

// ---- step 2: mirror-symmetric pairing --------------------------------------------------------------------------
__global__ __launch_bounds__(EB) void succ_kernel(const uint32_t *from, const uint32_t *mirror, uint32_t n_darts, const uint32_t *row,
                                                 const uint32_t *adj, const uint32_t *pos, uint32_t *succ, uint32_t *parent,
                                                 uint32_t *error) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e >= n_darts) return;
    const uint32_t vm = from[e ^ 1];  // e = (u -> v)  <=>  e^1 = (mirror v -> mirror u)
    const uint32_t v = mirror[vm];
    const uint32_t dv = row[v + 1] - row[v];
    uint32_t j = pos[e ^ 1];
    if (v == vm) {
        if (dv & 1) {
            atomicOr(error, 1u);
            return;
        }
        j ^= 1;
    } else if (dv != row[vm + 1] - row[vm]) {
        atomicOr(error, 1u);  // not Eulerian
        return;
    }
    succ[e] = adj[row[v] + j];
    parent[e] = e;
}

// ---- step 3: trail labels ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(EB) void union_succ_kernel(const uint32_t *succ, uint32_t n_darts, uint32_t *parent) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e < n_darts) uf_union(parent, e, succ[e]);
}
__global__ __launch_bounds__(EB) void flatten_kernel(uint32_t *parent, uint32_t n, uint32_t *label) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e < n) label[e] = uf_find(parent, e);
}
// comp[e] = smaller trail label of the pair {trail of e, trail of e^1}; second union-find starts as identity
__global__ __launch_bounds__(EB) void comp_kernel(const uint32_t *label, uint32_t n_darts, uint32_t *comp, uint32_t *parent2) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e >= n_darts) return;
    const uint32_t a = label[e], b = label[e ^ 1];
    comp[e] = a < b ? a : b;
    parent2[e] = e;
}

// ---- step 4: spanning forest (deterministic hooking rounds) + successor rotation ------------------------------
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
__global__ __launch_bounds__(EB) void propose_kernel(const uint32_t *mirror, uint32_t n_nodes, const uint32_t *row, const uint32_t *adj,
                                                    const uint32_t *comp, uint32_t *parent2, unsigned long long *best) {
    const uint32_t v = blockIdx.x * EB + threadIdx.x;
    if (v >= n_nodes) return;
    uint32_t r0 = 0;
    for_each_passage(mirror, row, adj, v, [&](uint32_t i, uint32_t, uint32_t b) {
        const uint32_t r = uf_find(parent2, comp[b]);
        if (i == 0) {
            r0 = r;
            return;
        }
        if (r == r0) return;
        const uint32_t hi = r > r0 ? r : r0, lo = r > r0 ? r0 : r;
        const unsigned long long key = ((unsigned long long)lo << 32) | b;
        // a long trail receives one proposal per shared binode: filter the losers before they queue up on its word
        if (key < __hip_atomic_load(&best[hi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&best[hi], key);
    });
}
__global__ __launch_bounds__(EB) void hook_kernel(uint32_t n_darts, uint32_t *parent2, const unsigned long long *best, uint32_t *selected,
                                                 uint32_t *changed) {
    const uint32_t r = blockIdx.x * EB + threadIdx.x;
    if (r >= n_darts) return;
    const unsigned long long p = best[r];
    if (p == ~0ull) return;
    parent2[r] = (uint32_t)(p >> 32);  // r was a root when it was proposed for, and only this thread writes it
    selected[(uint32_t)p] = 1u;        // the passage whose out-dart this is joins its node's rotation
    *changed = 1u;
}
__global__ __launch_bounds__(EB) void rotate_kernel(const uint32_t *mirror, uint32_t n_nodes, const uint32_t *row, const uint32_t *adj,
                                                   const uint32_t *selected, uint32_t *succ) {
    const uint32_t v = blockIdx.x * EB + threadIdx.x;
    if (v >= n_nodes) return;
    uint32_t a_prev = 0, b0 = 0;
    bool any = false;
    for_each_passage(mirror, row, adj, v, [&](uint32_t i, uint32_t a, uint32_t b) {
        if (i == 0) {
            a_prev = a;
            b0 = b;
            return;
        }
        if (!selected[b]) return;
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
__device__ __forceinline__ bool hash_splitter(uint32_t e) { return ((e * 0x9E3779B1u) >> 26) == 0; }

// flag[e] = 1 for splitters; rootflag packed in bit 1
__global__ __launch_bounds__(EB) void splitter_flag_kernel(const uint32_t *comp, uint32_t *parent2, uint32_t n_darts, uint32_t *flag,
                                                          uint8_t *is_root) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e >= n_darts) return;
    const bool root = uf_find(parent2, comp[e]) == e;  // smallest dart of its connected component
    is_root[e] = root;
    flag[e] = (root || hash_splitter(e)) ? 1u : 0u;
}
__global__ __launch_bounds__(EB) void splitter_compact_kernel(const uint32_t *flag, const uint32_t *sidx, uint32_t n_darts, uint32_t *splitters) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e < n_darts && flag[e]) splitters[sidx[e]] = e;
}
// succ gets bit 31 set where the successor is a splitter (n_darts < 2^31 is checked on the host)
__global__ __launch_bounds__(EB) void succ_mark_kernel(uint32_t *succ, const uint32_t *flag, uint32_t n_darts) {
    const uint32_t e = blockIdx.x * EB + threadIdx.x;
    if (e >= n_darts) return;
    const uint32_t s = succ[e];
    if (flag[s]) succ[e] = s | SPLIT_FLAG;
}
__global__ __launch_bounds__(EB) void walk_measure_kernel(const uint32_t *succ, const uint32_t *splitters, const uint32_t *sidx,
                                                         const uint8_t *is_root, uint32_t n_split, uint32_t n_darts, uint32_t *seg_len,
                                                         uint32_t *next_split, uint32_t *jump, uint32_t *dist, uint32_t *error) {
    const uint32_t i = blockIdx.x * EB + threadIdx.x;
    if (i >= n_split) return;
    const uint32_t s = splitters[i];
    uint32_t x = succ[s], len = 1;
    while (!(x & SPLIT_FLAG)) {
        x = succ[x];
        if (++len > n_darts) {  // cannot happen for a permutation; guards against a corrupted successor array
            atomicOr(error, 2u);
            break;
        }
    }
    const uint32_t nxt = sidx[x & ~SPLIT_FLAG];
    seg_len[i] = len;
    next_split[i] = nxt;
    const bool root = is_root[s];
    jump[i] = root ? i : nxt;  // roots are the terminals of the reduced lists
    dist[i] = root ? 0u : len;
}
__global__ __launch_bounds__(EB) void wyllie_kernel(const uint32_t *jump_in, const uint32_t *dist_in, uint32_t n, uint32_t *jump_out,
                                                   uint32_t *dist_out) {
    const uint32_t i = blockIdx.x * EB + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = jump_in[i];
    jump_out[i] = jump_in[j];
    dist_out[i] = dist_in[i] + dist_in[j];
}
// per root splitter (ascending dart id): length of its trail
__global__ __launch_bounds__(EB) void root_flag_kernel(const uint32_t *splitters, const uint8_t *is_root, uint32_t n_split, uint32_t *rflag) {
    const uint32_t i = blockIdx.x * EB + threadIdx.x;
    if (i < n_split) rflag[i] = is_root[splitters[i]];
}
__global__ __launch_bounds__(EB) void root_len_kernel(const uint32_t *rflag, const uint32_t *ridx, const uint32_t *seg_len,
                                                     const uint32_t *next_split, const uint32_t *dist, uint32_t n_split, uint32_t *cyc_len) {
    const uint32_t i = blockIdx.x * EB + threadIdx.x;
    if (i >= n_split || !rflag[i]) return;
    cyc_len[ridx[i]] = seg_len[i] + dist[next_split[i]];  // next == i (single splitter): dist 0
}
__global__ __launch_bounds__(EB) void walk_write_kernel(const uint32_t *succ, const uint32_t *splitters, const uint32_t *rflag,
                                                       const uint32_t *ridx, const uint32_t *jump, const uint32_t *dist,
                                                       const uint32_t *seg_len, const uint32_t *cyc_len, const uint32_t *cyc_base,
                                                       uint32_t n_split, uint32_t *out) {
    const uint32_t i = blockIdx.x * EB + threadIdx.x;
    if (i >= n_split) return;
    const uint32_t t = jump[i];
    if (!rflag[t]) return;  // mirror trail (no root on it): not emitted
    const uint32_t r = ridx[t];
    uint32_t p = cyc_base[r] + (i == t ? 0u : cyc_len[r] - dist[i]);
    uint32_t x = splitters[i];
    const uint32_t len = seg_len[i];
    for (uint32_t j = 0; j < len; j++) {
        out[p++] = x;
        x = succ[x] & ~SPLIT_FLAG;
    }
}

// Stream-ordered allocations from the device's default memory pool, which is told to keep freed memory: the ~25
// work arrays of a call cost milliseconds to map afresh, and a driver that Eulerises graph after graph reuses them.
thread_local hipStream_t g_alloc_stream = nullptr;
struct Buf {
    void *p = nullptr;
    ~Buf() {
        if (p) (void)hipFreeAsync(p, g_alloc_stream);
    }
    template <typename T>
    T *alloc(uint64_t n) {
        HIP_CHECK(hipMallocAsync(&p, (n ? n : 1) * sizeof(T), g_alloc_stream));
        return (T *)p;
    }
};

// out may alias in
void scan_u32(hipStream_t st, const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *block_sums, uint32_t *d_total) {
    const uint32_t nb = (uint32_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
    if (nb == 0) {
        HIP_CHECK(hipMemsetAsync(d_total, 0, 4, st));
        return;
    }
    scan_reduce_kernel<<<nb, EB, 0, st>>>(in, n, block_sums);
    scan_block_sums_kernel<<<1, EB, 0, st>>>(block_sums, nb, d_total);
    scan_apply_kernel<<<nb, EB, 0, st>>>(in, n, block_sums, out);
}

}  // namespace

Walks device_euler_cycles(const HostGraph &g, int device_id, double *kernel_ms_out) {
    const uint64_t E64 = g.edge_count(), V64 = g.node_count();
    Walks result;
    if (E64 == 0) return result;
    if (E64 >= 0x7FFFFFFFull || (E64 & 1)) MTG_DIE("device_euler_cycles: %llu directed edges do not fit the 31-bit dart ids", (unsigned long long)E64);
    const uint32_t E = (uint32_t)E64, V = (uint32_t)V64;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device_id)
        MTG_DIE("device_euler_cycles: no MI355X device %d visible (there is no CPU fallback for this mode)", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    static hipStream_t streams[64] = {nullptr};
    if (device_id >= 64) MTG_DIE("device_euler_cycles: device id %d out of range", device_id);
    if (!streams[device_id]) {
        HIP_CHECK(hipStreamCreate(&streams[device_id]));
        hipMemPool_t pool;
        HIP_CHECK(hipDeviceGetDefaultMemPool(&pool, device_id));
        uint64_t keep = UINT64_MAX;
        HIP_CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    }
    hipStream_t st = streams[device_id];
    g_alloc_stream = st;
    hipEvent_t ev0, ev1;
    HIP_CHECK(hipEventCreate(&ev0));
    HIP_CHECK(hipEventCreate(&ev1));

    Buf b_best, b_from, b_mirror, b_row, b_cursor, b_adj, b_pos, b_succ, b_parent, b_label, b_comp, b_parent2, b_flag, b_sidx, b_isroot,
        b_bsum, b_small;
    uint32_t *d_from = b_from.alloc<uint32_t>(E);
    uint32_t *d_mirror = b_mirror.alloc<uint32_t>(V);
    uint32_t *d_row = b_row.alloc<uint32_t>((uint64_t)V + 1);
    uint32_t *d_cursor = b_cursor.alloc<uint32_t>(V);
    uint32_t *d_adj = b_adj.alloc<uint32_t>(E);
    uint32_t *d_pos = b_pos.alloc<uint32_t>(E);
    uint32_t *d_succ = b_succ.alloc<uint32_t>(E);
    uint32_t *d_parent = b_parent.alloc<uint32_t>(E);
    uint32_t *d_label = b_label.alloc<uint32_t>(E);
    uint32_t *d_comp = b_comp.alloc<uint32_t>(E);
    uint32_t *d_parent2 = b_parent2.alloc<uint32_t>(E);
    uint32_t *d_flag = b_flag.alloc<uint32_t>(E);
    uint32_t *d_sidx = b_sidx.alloc<uint32_t>(E);
    uint8_t *d_isroot = b_isroot.alloc<uint8_t>(E);
    unsigned long long *d_best = b_best.alloc<unsigned long long>(E);
    const uint64_t max_scan = std::max<uint64_t>(E, (uint64_t)V + 1);
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(max_scan / SCAN_CHUNK + 2);
    uint32_t *d_small = b_small.alloc<uint32_t>(8);  // [0] error, [1..] scan totals
    uint32_t *d_error = d_small, *d_total = d_small + 1;

    HIP_CHECK(hipMemcpyAsync(d_from, g.e_from.data(), (uint64_t)E * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(d_mirror, g.mirror.data(), (uint64_t)V * 4, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemsetAsync(d_small, 0, 32, st));
    HIP_CHECK(hipEventRecord(ev0, st));

    // 1. buckets
    HIP_CHECK(hipMemsetAsync(d_row, 0, ((uint64_t)V + 1) * 4, st));
    HIP_CHECK(hipMemsetAsync(d_cursor, 0, (uint64_t)V * 4, st));
    degree_kernel<<<grid_for(E), EB, 0, st>>>(d_from, E, d_row);
    scan_u32(st, d_row, (uint64_t)V + 1, d_row, d_bsum, d_total);
    fill_kernel<<<grid_for(E), EB, 0, st>>>(d_from, E, d_row, d_cursor, d_adj, d_pos);
    sort_buckets_kernel<<<grid_for(V), EB, 0, st>>>(V, d_row, d_adj, d_pos);
    // 2. pairing, 3. trail labels
    succ_kernel<<<grid_for(E), EB, 0, st>>>(d_from, d_mirror, E, d_row, d_adj, d_pos, d_succ, d_parent, d_error);
    uint32_t h_small[8];
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[0]) MTG_DIE("device_euler_cycles: the graph is not Eulerian (greedytigs/mod.rs:708)");
    union_succ_kernel<<<grid_for(E), EB, 0, st>>>(d_succ, E, d_parent);
    flatten_kernel<<<grid_for(E), EB, 0, st>>>(d_parent, E, d_label);
    comp_kernel<<<grid_for(E), EB, 0, st>>>(d_label, E, d_comp, d_parent2);
    // 4. merge the trails of every connected component
    HIP_CHECK(hipMemsetAsync(d_flag, 0, (uint64_t)E * 4, st));  // `selected`, reused as the splitter flags afterwards
    int hook_rounds = 0;
    for (;; hook_rounds++) {
        if (hook_rounds > 64) MTG_DIE("device_euler_cycles: internal error (component hooking does not converge)");
        HIP_CHECK(hipMemsetAsync(d_best, 0xFF, (uint64_t)E * 8, st));
        HIP_CHECK(hipMemsetAsync(d_small + 4, 0, 4, st));
        propose_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_comp, d_parent2, d_best);
        hook_kernel<<<grid_for(E), EB, 0, st>>>(E, d_parent2, d_best, d_flag, d_small + 4);
        flatten_kernel<<<grid_for(E), EB, 0, st>>>(d_parent2, E, d_parent2);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (!h_small[4]) break;
    }
    rotate_kernel<<<grid_for(V), EB, 0, st>>>(d_mirror, V, d_row, d_adj, d_flag, d_succ);
    // 5. ranking
    splitter_flag_kernel<<<grid_for(E), EB, 0, st>>>(d_comp, d_parent2, E, d_flag, d_isroot);
    scan_u32(st, d_flag, E, d_sidx, d_bsum, d_total);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const uint32_t M = h_small[1];  // splitters
    if (M == 0) MTG_DIE("device_euler_cycles: internal error (no splitters)");

    Buf b_split, b_seglen, b_next, b_jump0, b_dist0, b_jump1, b_dist1, b_rflag, b_ridx, b_clen, b_cbase, b_out;
    uint32_t *d_split = b_split.alloc<uint32_t>(M);
    uint32_t *d_seglen = b_seglen.alloc<uint32_t>(M);
    uint32_t *d_next = b_next.alloc<uint32_t>(M);
    uint32_t *d_jump[2] = {b_jump0.alloc<uint32_t>(M), b_jump1.alloc<uint32_t>(M)};
    uint32_t *d_dist[2] = {b_dist0.alloc<uint32_t>(M), b_dist1.alloc<uint32_t>(M)};
    uint32_t *d_rflag = b_rflag.alloc<uint32_t>(M);
    uint32_t *d_ridx = b_ridx.alloc<uint32_t>(M);
    splitter_compact_kernel<<<grid_for(E), EB, 0, st>>>(d_flag, d_sidx, E, d_split);
    succ_mark_kernel<<<grid_for(E), EB, 0, st>>>(d_succ, d_flag, E);
    walk_measure_kernel<<<grid_for(M), EB, 0, st>>>(d_succ, d_split, d_sidx, d_isroot, M, E, d_seglen, d_next, d_jump[0], d_dist[0], d_error);
    int cur = 0;
    for (uint64_t span = 1; span < (uint64_t)M * 2; span <<= 1) {  // after r rounds a pointer spans 2^r reduced elements
        wyllie_kernel<<<grid_for(M), EB, 0, st>>>(d_jump[cur], d_dist[cur], M, d_jump[cur ^ 1], d_dist[cur ^ 1]);
        cur ^= 1;
    }
    root_flag_kernel<<<grid_for(M), EB, 0, st>>>(d_split, d_isroot, M, d_rflag);
    scan_u32(st, d_rflag, M, d_ridx, d_bsum, d_total + 1);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[0]) MTG_DIE("device_euler_cycles: internal error (successor array is not a permutation)");
    const uint32_t R = h_small[2];  // connected components = closed walks
    uint32_t *d_clen = b_clen.alloc<uint32_t>(R);
    uint32_t *d_cbase = b_cbase.alloc<uint32_t>(R);
    uint32_t *d_out = b_out.alloc<uint32_t>(E / 2);
    root_len_kernel<<<grid_for(M), EB, 0, st>>>(d_rflag, d_ridx, d_seglen, d_next, d_dist[cur], M, d_clen);
    scan_u32(st, d_clen, R, d_cbase, d_bsum, d_total + 2);
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 32, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    if (h_small[3] != E / 2)
        MTG_DIE("device_euler_cycles: internal error (closed walks cover %u of %u biedges)", h_small[3], E / 2);
    walk_write_kernel<<<grid_for(M), EB, 0, st>>>(d_succ, d_split, d_rflag, d_ridx, d_jump[cur], d_dist[cur], d_seglen, d_clen, d_cbase, M,
                                                   d_out);
    HIP_CHECK(hipEventRecord(ev1, st));

    result.edges.resize(E / 2);
    std::vector<uint32_t> clen(R);
    HIP_CHECK(hipMemcpyAsync(result.edges.data(), d_out, (uint64_t)(E / 2) * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(clen.data(), d_clen, (uint64_t)R * 4, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    result.limits.resize(R);
    uint64_t run = 0;
    for (uint32_t r = 0; r < R; r++) {
        run += clen[r];
        result.limits[r] = run;
    }
    if (kernel_ms_out) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
        *kernel_ms_out = ms;
    }
    HIP_CHECK(hipEventDestroy(ev0));
    HIP_CHECK(hipEventDestroy(ev1));
    return result;
}

}  // namespace mtg
