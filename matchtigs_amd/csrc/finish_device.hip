// finish_device.hip -- everything after the claim loop on the MI355X: dummy biedge insertion, the Euleriser, the Euler
// bicycles (or the records of the reference-order host walk) and the cutter, over flat dart arrays in HBM.
//
//   stage                         reference lines (all under /root/reference/src/implementation/)
//   matched-pair darts            greedytigs/mod.rs:678-689
//   Euleriser                     mod.rs:392-649 (+ choose_in_node_from_iterator :252-285): the SAME sequence of breaking edges
//   Euler bicycles                call greedytigs/mod.rs:722 / eulertigs/mod.rs:119 -> euler_device.hip (parallel, own order) or
//                                 the host walk in the reference's order over records built here (euler_lean.cpp)
//   rotate + cut                  greedytigs/mod.rs:726-789 == eulertigs/mod.rs:123-186
//
// A dart is a directed edge; dart e leaves from[e], its mirror dart is e ^ 1, its head is mirror[from[e ^ 1]]. Darts
// [0, E0) are the original edges, [E0, E0 + 2P) the matched pairs in claim order, the breaking edges follow in the order the
// reference adds them -- so "dummy" and "breaking" are id comparisons (like the host cutter, host_pipeline.cpp).
//
// The Euleriser in parallel. After the self-mirror phase the reference repeats: o = LARGEST node that still misses out-edges,
// t = SMALLEST node that still misses in-edges (with one exception rule), add o -> t and its mirror, and decrement the four
// counters of o, t, mirror(t), mirror(o). Every missing edge of an out-node o is also a missing edge of the in-node mirror(o), so
// the state is one list of "units" (binode, copy) seen in two orders: order A = by out-node descending, order B = by in-node
// ascending (copies of one binode in opposite order). A step removes the first live unit of A and the first live unit of B, and
// the live set is always {A-rank >= a, B-rank >= b}: the whole state is two cursors. While neither cursor has to skip, step s
// simply pairs A[s] with B[s + d] (d = 1 if the odd self-mirror took B[0]), and "no skip up to step s" is the per-unit test
// B-rank(A[s]) > s + d and A-rank(B[s + d]) > s -- a parallel map plus a min-reduction for the first irregular step s*. In
// graphs whose mirror nodes have neighbouring ids (every de Bruijn graph builder, and the generator here) order B is order A
// reversed, the cursors meet in the middle and s* is the last step; otherwise the steps from s* on are replayed sequentially on
// the host over the residual counters, which is the reference's loop itself. Either way the edges and their order are the
// reference's (tests: against make_eulerian and the oracle's ordered-map form, including scrambled mirror numberings).
//
// Everything is integer streaming / gather work on arrays over darts and nodes; HBM-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <numeric>
#include <vector>

#include "../../include/mtg_policy.h"
#include "device.hpp"
#include "finish_device.hpp"
#include "hugebuf.hpp"
#include "parallel.hpp"

namespace mtg {

using namespace hu;

Walks euler_cycles_generic(const HostGraph &g);

namespace {

// ---- degrees and imbalance ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(EB) void degree_kernel(const uint32_t *from, uint64_t n, uint32_t *deg) {
    const uint64_t e = gid();
    if (e < n) atomicAdd(&deg[from[e]], 1u);
}
__global__ __launch_bounds__(EB) void row_degree_kernel(uint64_t n_nodes, const uint32_t *row, uint32_t *deg) {  // (from kept buckets)
    const uint64_t v = gid();
    if (v < n_nodes) deg[v] = row[v + 1] - row[v];
}
__global__ __launch_bounds__(EB) void pair_degree_kernel(const mtg_pair *pairs, uint64_t n, const uint32_t *mirror, uint32_t *deg) {
    const uint64_t i = gid();
    if (i >= n) return;
    atomicAdd(&deg[pairs[i].out_node], 1u);
    atomicAdd(&deg[mirror[pairs[i].in_node]], 1u);
}
__global__ __launch_bounds__(EB) void pair_darts_kernel(const mtg_pair *pairs, uint64_t n, const uint32_t *mirror, uint32_t *from_pairs,
                                                       uint32_t *pair_w) {
    const uint64_t i = gid();
    if (i >= n) return;
    from_pairs[2 * i] = pairs[i].out_node;           // out -> in
    from_pairs[2 * i + 1] = mirror[pairs[i].in_node];  // mirror(in) -> mirror(out)
    pair_w[i] = (uint32_t)pairs[i].distance;
}
// find_non_eulerian_binodes_with_differences (mod.rs:408-427): cin = missing in-edges (diff > 0), cout = missing out-edges
// (diff < 0), sm = self-mirror node with odd degree (difference-0 entry)
struct Need { uint32_t ci, co, sm; };
__device__ __forceinline__ Need need_of(uint32_t v, const uint32_t *mirror, const uint32_t *deg) {
    const uint32_t mv = mirror[v], d = deg[v];
    Need nd{0u, 0u, 0u};
    if (mv == v) nd.sm = d & 1u;
    else {
        const uint32_t dm = deg[mv];
        if (d > dm) nd.ci = d - dm;
        else nd.co = dm - d;
    }
    return nd;
}
// Round 4: two passes over the nodes instead of twelve kernels (need, three scans of three kernels, self-mirror compaction,
// expansion). A workgroup takes NEED_CHUNK consecutive nodes, 8 per thread: pass 1 sums the three counters per chunk (three small
// scans over the chunk sums follow), pass 2 recomputes them, ranks them inside the chunk and writes everything that depends on the
// ranks: the counters and their exclusive prefixes (the zip's per-unit tests gather them), the self-mirror list, and the two unit
// orders -- order A: out-nodes descending (A-offset of o = N - p_out[o] - cout[o]), order B: in-nodes ascending (B-offset of
// t = p_in[t]).
constexpr int NEED_PER = 8, NEED_CHUNK = EB * NEED_PER;
// (node of pass i and thread t of a chunk: chunk * NEED_CHUNK + i * EB + t -- consecutive lanes, consecutive nodes: coalesced loads
// and stores; ascending node order = pass, wave, lane)
__global__ __launch_bounds__(EB) void need_count_kernel(uint64_t n_nodes, const uint32_t *mirror, const uint32_t *deg, uint32_t *chunk_in,
                                                       uint32_t *chunk_out, uint32_t *chunk_sm) {
    __shared__ uint32_t s_in, s_out, s_sm;
    if (threadIdx.x == 0) { s_in = 0; s_out = 0; s_sm = 0; }
    __syncthreads();
    uint32_t si = 0, so = 0, ss = 0;
#pragma unroll
    for (int i = 0; i < NEED_PER; i++) {
        const uint64_t v = (uint64_t)blockIdx.x * NEED_CHUNK + (uint64_t)i * EB + threadIdx.x;
        if (v < n_nodes) {
            const Need nd = need_of((uint32_t)v, mirror, deg);
            si += nd.ci; so += nd.co; ss += nd.sm;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { si += __shfl_down(si, off, 64); so += __shfl_down(so, off, 64); ss += __shfl_down(ss, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_in, si); atomicAdd(&s_out, so); atomicAdd(&s_sm, ss); }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_in[blockIdx.x] = s_in; chunk_out[blockIdx.x] = s_out; chunk_sm[blockIdx.x] = s_sm; }
}
// (chunk_*: exclusive scans of the chunk sums; n_units = N, the total of the in- (= out-) counters, read from device memory)
__global__ __launch_bounds__(EB) void need_emit_kernel(uint64_t n_nodes, const uint32_t *mirror, const uint32_t *deg, const uint32_t *chunk_in,
                                                      const uint32_t *chunk_out, const uint32_t *chunk_sm, const uint32_t *n_units_ptr, uint32_t *cin,
                                                      uint32_t *cout, uint32_t *p_in, uint32_t *p_out, uint32_t *sm_list, uint32_t *a_node,
                                                      uint32_t *b_node) {
    __shared__ uint32_t w_in[NEED_PER][EB / 64], w_out[NEED_PER][EB / 64], w_sm[NEED_PER][EB / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    Need nd[NEED_PER];
    uint32_t xi[NEED_PER], xo[NEED_PER], xs[NEED_PER];  // exclusive prefixes inside the wave
#pragma unroll
    for (int i = 0; i < NEED_PER; i++) {
        const uint64_t v = (uint64_t)blockIdx.x * NEED_CHUNK + (uint64_t)i * EB + threadIdx.x;
        nd[i] = Need{0u, 0u, 0u};
        if (v < n_nodes) nd[i] = need_of((uint32_t)v, mirror, deg);
        uint32_t a = nd[i].ci, b = nd[i].co, c = nd[i].sm;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ua = __shfl_up(a, off, 64), ub = __shfl_up(b, off, 64), uc = __shfl_up(c, off, 64);
            if (lane >= off) { a += ua; b += ub; c += uc; }
        }
        xi[i] = a - nd[i].ci; xo[i] = b - nd[i].co; xs[i] = c - nd[i].sm;
        if (lane == 63) { w_in[i][wv] = a; w_out[i][wv] = b; w_sm[i][wv] = c; }
    }
    __syncthreads();
    const uint32_t n_units = *n_units_ptr;
    uint32_t ri = chunk_in[blockIdx.x], ro = chunk_out[blockIdx.x], rs = chunk_sm[blockIdx.x];  // running offsets: pass, then wave
#pragma unroll
    for (int i = 0; i < NEED_PER; i++) {
        uint32_t bi = ri, bo = ro, bs = rs;
        for (int w = 0; w < EB / 64; w++) {
            if (w < wv) { bi += w_in[i][w]; bo += w_out[i][w]; bs += w_sm[i][w]; }
            ri += w_in[i][w]; ro += w_out[i][w]; rs += w_sm[i][w];
        }
        const uint64_t v64 = (uint64_t)blockIdx.x * NEED_CHUNK + (uint64_t)i * EB + threadIdx.x;
        if (v64 >= n_nodes) continue;
        const uint32_t v = (uint32_t)v64, ci = nd[i].ci, co = nd[i].co;
        const uint32_t pi = bi + xi[i], po = bo + xo[i];
        cin[v] = ci; cout[v] = co; p_in[v] = pi; p_out[v] = po;
        for (uint32_t j = 0; j < ci; j++) b_node[pi + j] = v;
        if (co) {
            const uint32_t base = n_units - po - co;
            for (uint32_t j = 0; j < co; j++) a_node[base + j] = v;
        }
        if (nd[i].sm) sm_list[bs + xs[i]] = v;
    }
}
// first irregular step (see the header): min over s of "a cursor would have to skip at step s"
__global__ __launch_bounds__(EB) void zip_check_kernel(uint32_t n_steps, uint32_t delta, uint32_t n_units, const uint32_t *mirror,
                                                      const uint32_t *cin, const uint32_t *cout, const uint32_t *p_in, const uint32_t *p_out,
                                                      const uint32_t *a_node, const uint32_t *b_node, uint32_t *s_star) {
    const uint64_t s = gid();
    if (s >= n_steps) return;
    bool bad = false;
    {
        const uint32_t o = a_node[s], c = cout[o];
        const uint64_t a_off = (uint64_t)n_units - p_out[o] - c;
        const uint64_t j = s - a_off;                                     // copy of o at A-rank s
        const uint64_t b_rank = (uint64_t)p_in[mirror[o]] + (c - 1 - j);  // copies sit in B in the opposite order
        bad |= b_rank <= s + delta;
    }
    {
        const uint32_t t = b_node[s + delta], c = cin[t];
        const uint64_t jb = (s + delta) - p_in[t];
        const uint64_t j = c - 1 - jb;
        const uint32_t o = mirror[t];
        const uint64_t a_rank = ((uint64_t)n_units - p_out[o] - c) + j;
        bad |= a_rank <= s;
    }
    if (bad) atomicMin(s_star, (uint32_t)s);
}
// breaking biedge b = b0 + s: from[2b] = out-node, from[2b + 1] = mirror(in-node)   (mod.rs:572-577)
__global__ __launch_bounds__(EB) void zip_emit_kernel(uint32_t n_steps, uint32_t delta, const uint32_t *mirror, const uint32_t *a_node,
                                                     const uint32_t *b_node, uint32_t *from_brk) {
    const uint64_t s = gid();
    if (s >= n_steps) return;
    from_brk[2 * s] = a_node[s];
    from_brk[2 * s + 1] = mirror[b_node[s + delta]];
}
// self-mirror phase (mod.rs:481-524): pairs (sm[2p] -> sm[2p+1]); the odd one out takes the first in-node
__global__ __launch_bounds__(EB) void sm_emit_kernel(uint32_t n_sm, const uint32_t *sm_list, const uint32_t *b_node, const uint32_t *mirror,
                                                    uint32_t *from_brk) {
    const uint64_t p = gid();
    if (2 * p + 1 < n_sm) {
        from_brk[2 * p] = sm_list[2 * p];
        from_brk[2 * p + 1] = sm_list[2 * p + 1];  // mirror(in) with in a self-mirror node
    } else if (2 * p + 1 == n_sm) {
        from_brk[2 * p] = sm_list[2 * p];
        from_brk[2 * p + 1] = mirror[b_node[0]];
    }
}
// units of every out-node that the parallel prefix [0, s*) left alive
__global__ __launch_bounds__(EB) void residual_kernel(uint64_t n_nodes, uint32_t s_star, uint32_t delta, uint32_t n_units, const uint32_t *mirror,
                                                     const uint32_t *cout, const uint32_t *p_in, const uint32_t *p_out, uint32_t *resid,
                                                     uint32_t *rflag, uint32_t *error) {
    const uint64_t v = gid();
    if (v >= n_nodes) return;
    const uint32_t c = cout[v];
    uint32_t r = 0;
    if (c) {
        const int64_t a_off = (int64_t)n_units - p_out[v] - c, b_off = p_in[mirror[v]];
        const int64_t ka = std::min<int64_t>(c, std::max<int64_t>(0, (int64_t)s_star - a_off));
        const int64_t kb = std::min<int64_t>(c, std::max<int64_t>(0, (int64_t)s_star + delta - b_off));
        if (ka + kb > c) atomicOr(error, 4u);
        else r = (uint32_t)(c - ka - kb);
    }
    resid[v] = r;
    rflag[v] = r ? 1u : 0u;
}
__global__ __launch_bounds__(EB) void residual_compact_kernel(uint64_t n_nodes, const uint32_t *resid, const uint32_t *rpos, uint32_t *r_node,
                                                             uint32_t *r_cnt) {
    const uint64_t v = gid();
    if (v >= n_nodes || !resid[v]) return;
    r_node[rpos[v]] = (uint32_t)v;
    r_cnt[rpos[v]] = resid[v];
}
__global__ __launch_bounds__(EB) void head_kernel(const uint32_t *from, const uint32_t *mirror, uint64_t lo, uint64_t hi, uint32_t *to_out) {
    const uint64_t e = lo + gid();
    if (e < hi) to_out[e - lo] = mirror[from[e ^ 1]];
}

// ---- records of the reference-order host walk (euler_lean.cpp) ---------------------------------------------------
__global__ __launch_bounds__(EB) void lean_ext_kernel(uint64_t n_nodes, const uint32_t *row, uint32_t *ext_need, uint32_t *error) {
    const uint64_t v = gid();
    if (v >= n_nodes) return;
    const uint32_t d = row[v + 1] - row[v];
    ext_need[v] = d > 3 ? d - 3 : 0;
    if (d > 65535) atomicOr(error, 8u);
}
__global__ __launch_bounds__(EB) void lean_build_kernel(uint64_t n_nodes, const uint32_t *row, const uint32_t *adj, const uint32_t *from,
                                                       const uint32_t *mirror, const uint32_t *ext_off, LeanNode *nodes, uint32_t *ext_eid,
                                                       uint32_t *ext_to) {
    const uint64_t v = gid();
    if (v >= n_nodes) return;
    const uint32_t lo = row[v], d = row[v + 1] - lo;
    LeanNode r;
    r.deg = (uint16_t)(d > 65535 ? 65535 : d);
    r.pos = 0;
    r.ext_begin = ext_off[v];
    for (uint32_t p = 0; p < 3; p++) {
        r.eid[p] = NONE;
        r.to[p] = NONE;
    }
    for (uint32_t p = 0; p < d; p++) {  // position p of the iteration order (default: the p-th NEWEST out-dart, petgraph's order): policy P3 (mtg_policy.h); buckets ascend
        const uint32_t e = adj[lo + mtg_policy_adjacency_index(p, d)];
        const uint32_t t = mirror[from[e ^ 1]];
        if (p < 3) {
            r.eid[p] = e;
            r.to[p] = t;
        } else {
            ext_eid[r.ext_begin + p - 3] = e;
            ext_to[r.ext_begin + p - 3] = t;
        }
    }
    nodes[v] = r;
}

// The 256-byte records of the latency-optimised walk (euler_fast.cpp: own adjacency + the first three positions of every head +
// the first two positions of every head's heads) from the 32-byte ones: per node up to 3 + 9 gathers of 32-byte records, which
// the GPU does ~100 x faster than the host's threads (phases B and C there: 1.4 s at the bench size).
__global__ __launch_bounds__(EB) void wide_build_kernel(uint64_t n_nodes, const LeanNode *lean, EulerNode3 *wide) {
    const uint64_t v = gid();
    if (v >= n_nodes) return;
    const LeanNode l = lean[v];
    EulerNode3 r;
    for (int i = 0; i < 3; i++) {
        r.eid[i] = l.eid[i];
        r.to[i] = l.to[i];
    }
    for (int j = 0; j < 3; j++)  // (unused copies hold NONE, as in the 128-byte records: no uninitialised words reach the host)
        for (int q = 0; q < 3; q++) {
            r.sub_eid[j][q] = NONE;
            r.sub_to[j][q] = NONE;
            for (int t = 0; t < 2; t++) { r.sub2_eid[j][q][t] = NONE; r.sub2_to[j][q][t] = NONE; }
        }
    r.deg = l.deg;
    r.pos = 0;
    r.pad = 0;
    r.ext_begin = l.ext_begin;
    const uint32_t d = l.deg < 3 ? l.deg : 3;
    uint32_t info = 0, info2 = 0;
    for (uint32_t j = 0; j < d; j++) {
        const LeanNode w = lean[l.to[j]];
        const uint32_t c = w.deg < 3 ? w.deg : 3;
        info |= (c | (w.deg > 3 ? 4u : 0u)) << (3 * j);
        for (uint32_t q = 0; q < c; q++) {
            r.sub_eid[j][q] = w.eid[q];
            r.sub_to[j][q] = w.to[q];
            const LeanNode x = lean[w.to[q]];
            const uint32_t c2 = x.deg < 2 ? x.deg : 2;
            info2 |= (c2 | (x.deg > 2 ? 4u : 0u)) << (3 * (3 * j + q));
            for (uint32_t t = 0; t < c2; t++) {
                r.sub2_eid[j][q][t] = x.eid[t];
                r.sub2_to[j][q][t] = x.to[t];
            }
        }
    }
    r.sub_info = (uint16_t)info;
    r.sub2_info = info2;
    wide[v] = r;
}

// the 128-byte records (own adjacency + the first three positions of every head): one gather level
__global__ __launch_bounds__(EB) void mid_build_slice_kernel(uint64_t first, uint64_t n_slice, uint64_t n_nodes, const LeanNode *lean, EulerNode2 *mid) {
    const uint64_t s = gid();  // (records [first, first + n_slice) go to mid[0 .. n_slice): the slice that is downloaded next)
    if (s >= n_slice) return;
    const uint64_t v = first + s;
    if (v >= n_nodes) return;
    const LeanNode l = lean[v];
    EulerNode2 r;
    for (int i = 0; i < 3; i++) {
        r.eid[i] = l.eid[i];
        r.to[i] = l.to[i];
    }
    r.deg = l.deg;
    r.pos = 0;
    r.pad = 0;
    r.sub2_info = 0;
    r.ext_begin = l.ext_begin;
    for (int i = 0; i < 4; i++) r.pad2[i] = 0;
    const uint32_t d = l.deg < 3 ? l.deg : 3;
    uint32_t info = 0;
    for (uint32_t j = 0; j < 3; j++)
        for (uint32_t q = 0; q < 3; q++) { r.sub_eid[j][q] = NONE; r.sub_to[j][q] = NONE; }
    for (uint32_t j = 0; j < d; j++) {
        const LeanNode w = lean[l.to[j]];
        const uint32_t c = w.deg < 3 ? w.deg : 3;
        info |= (c | (w.deg > 3 ? 4u : 0u)) << (3 * j);
        for (uint32_t q = 0; q < c; q++) {
            r.sub_eid[j][q] = w.eid[q];
            r.sub_to[j][q] = w.to[q];
        }
    }
    r.sub_info = (uint16_t)info;
    mid[s] = r;
}

// ---- rotate + cut (greedytigs/mod.rs:726-789) ----------------------------------------------------------------------
// Round 4: three passes over the closed walks instead of fifteen kernels. A workgroup takes CUT_CHUNK consecutive positions of
// the back-to-back cycles; which cycle a position belongs to comes from the slice of the cycle offsets that overlaps the chunk
// (found by two binary searches, held in LDS: a chunk of 2048 positions meets at most 2049 cycles) -- no per-position head flags,
// no scan over the positions. The rotated cycle is never materialised: rotated position j of cycle c reads cyc[base + (j + r) % len].
//   rotation_kernel  the rotation point of every cycle (one 64-bit atomicMax per cycle)
//   cut_count_kernel kept edges and tig ends per chunk (-> two scans over the chunk counts)
//   cut_emit_kernel  the same flags again, ranked inside the chunk, written to their final places
struct CutIds {
    uint32_t n_orig;     // darts >= n_orig are dummies
    uint32_t first_brk;  // darts >= first_brk are breaking edges (weight k); matched dummies in between have pair_w
};
constexpr int CUT_PER = 8, CUT_CHUNK = EB * CUT_PER;

struct CycleSlice {
    uint32_t base[CUT_CHUNK + 2];  // offsets of the cycles c0 .. c0 + n - 1 that overlap the chunk (ascending)
    uint32_t c0, n;
};
// cycles overlapping positions [q0, q1): c0 = the cycle of q0, n = their number (every cycle has at least one position)
__device__ __forceinline__ void load_cycle_slice(CycleSlice &sl, const uint32_t *cbase, uint32_t n_cycles, uint64_t q0, uint64_t q1) {
    if (threadIdx.x == 0) {
        auto cycle_of = [&](uint64_t q) {  // largest c with cbase[c] <= q
            uint32_t lo = 0, hi = n_cycles;
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (cbase[mid] <= q) lo = mid; else hi = mid;
            }
            return lo;
        };
        sl.c0 = cycle_of(q0);
        sl.n = cycle_of(q1 - 1) - sl.c0 + 1;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < sl.n; i += EB) sl.base[i] = cbase[sl.c0 + i];
    __syncthreads();
}
__device__ __forceinline__ uint32_t slice_cycle_of(const CycleSlice &sl, uint64_t q) {  // index into the slice
    uint32_t lo = 0, hi = sl.n;
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (sl.base[mid] <= q) lo = mid; else hi = mid;
    }
    return lo;
}
// rotation point (:737-748): the first dummy that is strictly longer than every dummy before it = the first occurrence of the
// largest dummy weight = max over dummies of (weight, -position). One 64-bit atomicMax per cycle; a wave that lies inside one
// cycle reduces first, and a key that cannot win is filtered by a plain load (a giant cycle's 10^7 breaking edges would
// otherwise queue up on one word).
__global__ __launch_bounds__(EB) void rotation_kernel(const uint32_t *cyc, uint64_t n, const uint32_t *cbase, uint32_t n_cycles, CutIds ids, uint32_t k,
                                                     const uint32_t *pair_w, unsigned long long *rotkey) {
    __shared__ CycleSlice sl;
    const uint64_t q0 = (uint64_t)blockIdx.x * CUT_CHUNK, q1 = q0 + CUT_CHUNK < n ? q0 + CUT_CHUNK : n;
    load_cycle_slice(sl, cbase, n_cycles, q0, q1);
#pragma unroll
    for (int i = 0; i < CUT_PER; i++) {
        const uint64_t p = q0 + (uint64_t)i * EB + threadIdx.x;
        unsigned long long key = 0;
        uint32_t c = 0xFFFFFFFFu;
        if (p < n) {
            const uint32_t ci = slice_cycle_of(sl, p);
            c = sl.c0 + ci;
            const uint32_t e = cyc[p];
            if (e >= ids.n_orig) {
                const uint32_t w = e >= ids.first_brk ? k : pair_w[(e - ids.n_orig) >> 1];
                if (w > 0) key = ((unsigned long long)w << 32) | (0xFFFFFFFFu - (uint32_t)(p - sl.base[ci]));
            }
        }
        const uint32_t c_first = __shfl(c, 0, 64);
        if (__all(c == c_first || p >= n)) {  // the whole wave is in one cycle
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(key, off, 64);
                key = o > key ? o : key;
            }
            if ((threadIdx.x & 63) != 0) key = 0;
            c = c_first;
        }
        if (key && key > __hip_atomic_load(&rotkey[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&rotkey[c], key);
    }
}
// cut(j) = breaking edge, or any dummy at rotated index 0 (:767-769) or at the last index (the tail rule :779-788 drops a
// trailing dummy, which is the same as cutting there). keep = !cut; a kept edge ends a tig if it is the cycle's last edge or its
// successor is cut.
__device__ __forceinline__ bool is_cut(uint32_t e, uint32_t j, uint32_t len, const CutIds &ids) {
    return e >= ids.first_brk || (e >= ids.n_orig && (j == 0 || j + 1 == len));
}
// edge at rotated position q (and whether it is kept / ends a tig); cycles rotated by their rotation point (:746-748)
struct CutAt { uint32_t e; bool keep, end; };
__device__ __forceinline__ CutAt cut_at(const CycleSlice &sl, uint64_t q, const uint32_t *cyc, const uint32_t *clen, const unsigned long long *rotkey,
                                        const CutIds &ids) {
    const uint32_t ci = slice_cycle_of(sl, q), c = sl.c0 + ci;
    const uint32_t base = sl.base[ci], len = clen[c];
    const unsigned long long key = rotkey[c];
    const uint32_t r = key ? 0xFFFFFFFFu - (uint32_t)key : 0u;
    const uint32_t j = (uint32_t)(q - base);
    auto rot_at = [&](uint32_t jj) {  // cyc[base + (jj + r) % len] without the division (r < len, jj < len)
        const uint32_t t = len - r;   // positions jj < t come from jj + r, the others from jj - t
        return cyc[(uint64_t)base + (jj < t ? jj + r : jj - t)];
    };
    CutAt a;
    a.e = rot_at(j);
    a.keep = !is_cut(a.e, j, len, ids);
    a.end = a.keep && (j + 1 == len || is_cut(rot_at(j + 1), j + 1, len, ids));
    return a;
}
__global__ __launch_bounds__(EB) void cut_count_kernel(const uint32_t *cyc, uint64_t n, const uint32_t *cbase, const uint32_t *clen, uint32_t n_cycles,
                                                      const unsigned long long *rotkey, CutIds ids, uint32_t *chunk_keep, uint32_t *chunk_end) {
    __shared__ CycleSlice sl;
    __shared__ uint32_t s_keep, s_end;
    if (threadIdx.x == 0) { s_keep = 0; s_end = 0; }
    const uint64_t q0 = (uint64_t)blockIdx.x * CUT_CHUNK, q1 = q0 + CUT_CHUNK < n ? q0 + CUT_CHUNK : n;
    load_cycle_slice(sl, cbase, n_cycles, q0, q1);
    uint32_t nk = 0, ne = 0;
#pragma unroll
    for (int i = 0; i < CUT_PER; i++) {
        const uint64_t q = q0 + (uint64_t)i * EB + threadIdx.x;
        if (q >= n) continue;
        const CutAt a = cut_at(sl, q, cyc, clen, rotkey, ids);
        nk += a.keep ? 1u : 0u;
        ne += a.end ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) { nk += __shfl_down(nk, off, 64); ne += __shfl_down(ne, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_keep, nk); atomicAdd(&s_end, ne); }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_keep[blockIdx.x] = s_keep; chunk_end[blockIdx.x] = s_end; }
}
// (chunk_keep / chunk_end: exclusive scans of the counts. Limits travel as 32-bit words: fewer than 2^31 biedges.)
__global__ __launch_bounds__(EB) void cut_emit_kernel(const uint32_t *cyc, uint64_t n, const uint32_t *cbase, const uint32_t *clen, uint32_t n_cycles,
                                                     const unsigned long long *rotkey, CutIds ids, const uint32_t *chunk_keep, const uint32_t *chunk_end,
                                                     uint32_t *tig_edges, uint32_t *tig_limits) {
    __shared__ CycleSlice sl;
    __shared__ uint32_t w_keep[CUT_PER][EB / 64], w_end[CUT_PER][EB / 64];
    const uint64_t q0 = (uint64_t)blockIdx.x * CUT_CHUNK, q1 = q0 + CUT_CHUNK < n ? q0 + CUT_CHUNK : n;
    load_cycle_slice(sl, cbase, n_cycles, q0, q1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t e[CUT_PER];
    unsigned long long bk[CUT_PER], be[CUT_PER];
#pragma unroll
    for (int i = 0; i < CUT_PER; i++) {
        const uint64_t q = q0 + (uint64_t)i * EB + threadIdx.x;
        CutAt a{0u, false, false};
        if (q < n) a = cut_at(sl, q, cyc, clen, rotkey, ids);
        e[i] = a.e;
        bk[i] = __ballot(a.keep);
        be[i] = __ballot(a.end);
        if (lane == 0) { w_keep[i][wv] = (uint32_t)__popcll(bk[i]); w_end[i][wv] = (uint32_t)__popcll(be[i]); }
    }
    __syncthreads();
    uint32_t ko = chunk_keep[blockIdx.x], eo = chunk_end[blockIdx.x];  // ranks in position order: pass, wave, lane
#pragma unroll
    for (int i = 0; i < CUT_PER; i++) {
        for (int w = 0; w < EB / 64; w++) {
            if (w == wv && ((bk[i] >> lane) & 1ull)) {
                const uint32_t kp = ko + (uint32_t)__popcll(bk[i] & ((1ull << lane) - 1ull));
                tig_edges[kp] = e[i];
                if ((be[i] >> lane) & 1ull) tig_limits[eo + (uint32_t)__popcll(be[i] & ((1ull << lane) - 1ull))] = kp + 1;
            }
            ko += w_keep[i][w];
            eo += w_end[i][w];
        }
    }
}

struct Lap {
    const bool on = std::getenv("MTG_DEBUG") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double lap(const char *what) {
        const auto n = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(n - t).count();
        if (on) std::fprintf(stderr, "[mtg] device_finish: %-34s %8.3f ms\n", what, s * 1e3);
        t = n;
        return s;
    }
};

// The steps from s* on, sequentially, over the residual units (= the reference's loop, mod.rs:526-645, on compact arrays):
// r_node ascending out-nodes with r_cnt live units each; the in-entry of unit i is mirror(r_node[i]).
void eulerise_tail(const HostGraph &g, const std::vector<uint32_t> &r_node, std::vector<uint32_t> &cnt, std::vector<uint32_t> &brk_out,
                   std::vector<uint32_t> &brk_in) {
    const size_t n = r_node.size();
    std::vector<uint32_t> in_ord(n);  // indices by in-node ascending
    std::iota(in_ord.begin(), in_ord.end(), 0u);
    std::sort(in_ord.begin(), in_ord.end(), [&](uint32_t a, uint32_t b) { return g.mirror[r_node[a]] < g.mirror[r_node[b]]; });
    int64_t out_cur = (int64_t)n - 1;
    size_t in_cur = 0;
    auto first_in = [&](size_t from) {
        while (from < n && cnt[in_ord[from]] == 0) from++;
        return from;
    };
    for (;;) {
        while (out_cur >= 0 && cnt[(size_t)out_cur] == 0) out_cur--;
        if (out_cur < 0) break;
        const uint32_t oi = (uint32_t)out_cur;
        in_cur = first_in(in_cur);
        if (in_cur >= n) MTG_DIE("in_node_iterator.next().unwrap() on an empty map (implementation/mod.rs:262)");
        size_t pick = in_cur;
        if (in_ord[pick] == oi && cnt[oi] < 2) {  // t == mirror(o) and OUT[o] > -2 (:263): take the second key
            pick = first_in(pick + 1);
            if (pick >= n) MTG_DIE("No further in_nodes left (implementation/mod.rs:553)");
        }
        const uint32_t ti = in_ord[pick];
        brk_out.push_back(r_node[oi]);
        brk_in.push_back(g.mirror[r_node[ti]]);
        cnt[oi]--;  // OUT[o] += 1 and IN[mirror o] -= 1 (:582, :628-644)
        if (cnt[ti] == 0) MTG_DIE("internal error: Euleriser tail picked an exhausted in-node");
        cnt[ti]--;  // IN[t] -= 1 and OUT[mirror t] += 1 (:583, :609-627)
    }
    if (first_in(in_cur) < n) MTG_DIE("in_node_differences not empty after Eulerisation (implementation/mod.rs:648)");
}

}  // namespace

// Whole finish on device `device_id` for a graph that holds only its original edges: inserts the matched pairs (may be none:
// eulertigs), Eulerises, decomposes (euler_mode MTG_EULER_DEVICE: on the GPU; MTG_EULER_HOST_REFERENCE_ORDER: host walk over
// GPU-built records, the reference's order) and cuts. Appends the dummy edges to g's edge arrays (unlinked). times_out[0..5] =
// seconds of {upload + insertion + Euleriser, host-graph materialisation, Euler decomposition, cut + download}, [4] = kernel ms
// of the device decomposition, [5] = number of breaking biedges.
// host memory this process can still take: MemAvailable, and what the cgroup leaves (v2: memory.max - memory.current)
static uint64_t host_available_bytes() {
    uint64_t avail = ~0ull;
    if (FILE *f = std::fopen("/proc/meminfo", "r")) {
        char line[256];
        unsigned long long kb = 0;
        while (std::fgets(line, sizeof line, f))
            if (std::sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { avail = (uint64_t)kb << 10; break; }
        std::fclose(f);
    }
    unsigned long long lim = 0, cur = 0;
    bool have_lim = false, have_cur = false;
    if (FILE *f = std::fopen("/sys/fs/cgroup/memory.max", "r")) { have_lim = std::fscanf(f, "%llu", &lim) == 1; std::fclose(f); }  // ("max" does not parse: no limit)
    if (FILE *f = std::fopen("/sys/fs/cgroup/memory.current", "r")) { have_cur = std::fscanf(f, "%llu", &cur) == 1; std::fclose(f); }
    if (have_lim && have_cur && lim > cur) avail = std::min<uint64_t>(avail, lim - cur);
    return avail;
}

// What the library keeps between calls on a GPU, given back on request (mtg_release_device_memory / mtg_graph_release_device_cache):
// the chunks of the device's arena that no live array sits in, and a graph's device copy of its original edges with their buckets.
void device_release_memory(int device_id) {
    if (device_id < 0 || device_id >= device_count()) return;
    HIP_CHECK(hipSetDevice(device_id));
    HIP_CHECK(hipStreamSynchronize(finish_stream(device_id)));
    device_arena(device_id).release_free_chunks(true);
}
uint64_t device_memory_held(int device_id) {
    if (device_id < 0 || device_id >= 64) return 0;
    return device_arena(device_id).reclaimable_bytes();
}
void device_set_finish_tuning(int records, int flags, long record_delay_us) {
    finish_tuning().records.store(records);
    finish_tuning().flags.store(flags);
    finish_tuning().record_delay_us.store(record_delay_us < 0 ? 0 : record_delay_us);
}
void device_warm_finish_kernels() {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(head_kernel));
}
void device_release_graph_cache(const HostGraph &g) {
    std::lock_guard<std::mutex> l(edge_cache_mutex());
    g.device_cache.reset();
}

// second stream per device: downloads that run beside the finish stream's kernels
static hipStream_t finish_side_stream(int device_id) {
    static hipStream_t streams[64] = {nullptr};
    static std::mutex mu;
    if (device_id < 0 || device_id >= 64) MTG_DIE("device id %d out of range", device_id);
    std::lock_guard<std::mutex> l(mu);
    if (!streams[device_id]) {
        HIP_CHECK(hipSetDevice(device_id));
        HIP_CHECK(hipStreamCreateWithFlags(&streams[device_id], hipStreamNonBlocking));
    }
    return streams[device_id];
}

// Measurement aid (mtg_set_finish_tuning flag 8): while the host walks the Euler cycles in the reference's order the GPU idles for
// seconds; with this a trivial kernel runs every 2 ms, which shows what the idle state costs the first kernels of the next step.
__global__ void keep_awake_kernel(uint32_t *p) { if (p && threadIdx.x == 1024) *p = 0; }
struct KeepAwake {
    std::atomic<bool> stop{false};
    std::thread th;
    KeepAwake(bool on, int device_id) {
        if (!on) return;
        th = std::thread([this, device_id]() {
            HIP_CHECK(hipSetDevice(device_id));
            hipStream_t s = finish_side_stream(device_id);
            while (!stop.load(std::memory_order_relaxed)) {
                keep_awake_kernel<<<1, 64, 0, s>>>(nullptr);
                (void)hipStreamSynchronize(s);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        });
    }
    ~KeepAwake() {
        stop.store(true);
        if (th.joinable()) th.join();
    }
};

ResidentTigs::~ResidentTigs() {
    hu::device_free_on(device, d_edges);
    hu::device_free_on(device, d_limits);
}
void ResidentTigs::download(Walks &w) const {
    int cur = 0;
    (void)hipGetDevice(&cur);
    HIP_CHECK(hipSetDevice(device));
    hipStream_t st = finish_stream(device);
    w.edges.resize(n_edges);
    w.limits.resize(n_tigs);
    download_sliced(w.edges.data(), d_edges, n_edges * 4, st, device);
    download_sliced_widen(w.limits.data(), d_limits, n_tigs, st, device);
    (void)hipSetDevice(cur);
}

Walks device_finish(HostGraph &g, const Pair *pairs, uint64_t n_pairs, uint64_t k, int device_id, int euler_mode, double times_out[12],
                    const mtg_pair *d_pairs_resident, TigSink *sink, ResidentTigs **resident_out) {
    if (sink) resident_out = nullptr;
    if (resident_out) *resident_out = nullptr;
    const uint64_t V = g.node_count(), E0 = g.n_original_edges;
    if (n_pairs && !pairs && !d_pairs_resident) MTG_DIE("device_finish: null pairs");
    if (g.edge_count() != E0) MTG_DIE("device_finish: the graph already holds dummy edges");
    if (k < 1 || k > 0xFFFFFFFFull) MTG_DIE("device_finish: k out of range");
    Walks tigs;
    for (int i = 0; i < 12 && times_out; i++) times_out[i] = 0;
    if (E0 == 0 && n_pairs == 0) return tigs;
    if (device_count() <= device_id) MTG_DIE("device_finish: no MI355X/HIP device %d (there is no CPU fallback for this path)", device_id);
    HIP_CHECK(hipSetDevice(device_id));
    hipStream_t st = finish_stream(device_id);
    Lap lap;
    static_assert(sizeof(Pair) == sizeof(mtg_pair), "pair layout");
    // (the stage's arrays are ranges of the device's arena: what an earlier stage of the call gave back, else one more chunk now)
    device_arena(device_id).ensure_free((E0 + 2 * n_pairs + E0 / 3) * 24 + V * 28);
    // GPU time of the stages (HIP events on the finish stream; read at the end): [0,1] insertion + Euleriser kernels, [2,3] buckets +
    // walk records (reference-order mode), [4,5] rotate + cut kernels; the decomposition has its own pair (device_euler_decompose)
    struct StageEvents {
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool set[6] = {false, false, false, false, false, false};
        void mark(int i, hipStream_t s) {
            if (!ev[i]) HIP_CHECK(hipEventCreate(&ev[i]));
            HIP_CHECK(hipEventRecord(ev[i], s));
            set[i] = true;
        }
        double ms(int a, int b) {
            if (!set[a] || !set[b]) return 0.0;
            float v = 0;
            HIP_CHECK(hipEventElapsedTime(&v, ev[a], ev[b]));
            return v;
        }
        ~StageEvents() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
    } sev;

    // ---- original darts + mirror: left on this GPU by an earlier stage of the graph (HostGraph::device_cache), else uploaded
    // now and left there for the next call; the pairs ----
    Buf b_pairs, b_deg, b_cin, b_cout, b_pin, b_pout, b_bsum, b_small;
    Buf b_from0_tmp, b_mirror_tmp;  // only when the graph's cache lives on another GPU
    const uint32_t *d_from0 = nullptr, *d_mirror = nullptr;
    if (const DeviceEdgeCache *cache = edge_cache_get(g, device_id)) {
        d_from0 = (const uint32_t *)cache->d_from;
        d_mirror = (const uint32_t *)cache->d_mirror;
    } else {
        const bool keep = !g.device_cache;
        uint32_t *up_from = nullptr, *up_mirror = nullptr;
        if (keep) {
            hu::device_malloc(&up_from, std::max<uint64_t>(E0, 1) * 4);
            hu::device_malloc(&up_mirror, std::max<uint64_t>(V, 1) * 4);
        } else {
            up_from = b_from0_tmp.alloc<uint32_t>(st, E0);
            up_mirror = b_mirror_tmp.alloc<uint32_t>(st, V);
        }
        if (E0) HIP_CHECK(hipMemcpyAsync(up_from, g.e_from.data(), E0 * 4, hipMemcpyHostToDevice, st));
        if (V) HIP_CHECK(hipMemcpyAsync(up_mirror, g.mirror.data(), V * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (keep) edge_cache_put(g, device_id, up_from, up_mirror);  // (if another thread got there first, ours are freed: read again)
        if (const DeviceEdgeCache *cache = keep ? edge_cache_get(g, device_id) : nullptr) {
            d_from0 = (const uint32_t *)cache->d_from;
            d_mirror = (const uint32_t *)cache->d_mirror;
        } else if (keep) MTG_DIE("device_finish: the graph's device cache changed hands during the call");
        else {
            d_from0 = up_from;
            d_mirror = up_mirror;
        }
    }
    // The buckets of the original darts (a static function of the graph, like the device copy of its edges) are built by the first
    // call on a graph and kept with its cache while they are small enough (8 GB) not to matter next to the stand-ins of BASELINE
    // configs[4]: the out-degrees come from their row array, and the buckets of the Eulerised darts from a merge (euler_device.hip).
    const uint32_t *d_row0 = nullptr, *d_adj0 = nullptr;
    if (const DeviceEdgeCache *cache = edge_cache_get(g, device_id)) {
        if (!cache->d_row0 && (V + 1 + E0) * 4 <= (8ull << 30) && E0 && !(finish_tuning().flags.load() & FT_NO_EDGE_CACHE)) {
            uint32_t *row0 = nullptr, *adj0 = nullptr;
            device_malloc(&row0, (V + 1) * 4);
            device_malloc(&adj0, E0 * 4);
            device_build_buckets(st, d_from0, E0, V, row0, adj0, nullptr);
            HIP_CHECK(hipStreamSynchronize(st));
            edge_cache_set_buckets(g, device_id, row0, adj0);
            cache = edge_cache_get(g, device_id);
        }
        if (cache && cache->d_row0) {
            d_row0 = (const uint32_t *)cache->d_row0;
            d_adj0 = (const uint32_t *)cache->d_adj0;
        }
    }
    sev.mark(0, st);
    // the pairs: where the claim replay left them in this GPU's HBM, else uploaded
    const mtg_pair *d_pairs = d_pairs_resident;
    if (!d_pairs) {
        mtg_pair *up = b_pairs.alloc<mtg_pair>(st, n_pairs);
        if (n_pairs) HIP_CHECK(hipMemcpyAsync(up, pairs, n_pairs * sizeof(mtg_pair), hipMemcpyHostToDevice, st));
        d_pairs = up;
    }
    uint32_t *d_deg = b_deg.alloc<uint32_t>(st, V);
    if (d_row0) row_degree_kernel<<<grid_for(V), EB, 0, st>>>(V, d_row0, d_deg);
    else {
        HIP_CHECK(hipMemsetAsync(d_deg, 0, V * 4, st));
        if (E0) degree_kernel<<<grid_for(E0), EB, 0, st>>>(d_from0, E0, d_deg);
    }
    if (n_pairs) pair_degree_kernel<<<grid_for(n_pairs), EB, 0, st>>>(d_pairs, n_pairs, d_mirror, d_deg);

    // ---- imbalance, unit orders ----
    uint32_t *d_cin = b_cin.alloc<uint32_t>(st, V), *d_cout = b_cout.alloc<uint32_t>(st, V);
    uint32_t *d_pin = b_pin.alloc<uint32_t>(st, V), *d_pout = b_pout.alloc<uint32_t>(st, V);
    uint32_t *d_bsum = b_bsum.alloc<uint32_t>(st, scan_blocks(std::max<uint64_t>(V, (E0 + 2 * n_pairs) / 2 + 1) * 2) + 2);
    uint32_t *d_small = b_small.alloc<uint32_t>(st, 16);  // [0] error, [1] N_in, [2] N_out, [3] n_sm, [4] s*, [5] residual nodes, [6..] cut totals
    HIP_CHECK(hipMemsetAsync(d_small, 0, 64, st));
    const uint64_t need_chunks = (V + NEED_CHUNK - 1) / NEED_CHUNK;
    Buf b_chunks;
    uint32_t *d_chunks = b_chunks.alloc<uint32_t>(st, 3 * need_chunks);  // per chunk of nodes: missing in-edges | missing out-edges | odd self-mirror nodes
    need_count_kernel<<<(unsigned)need_chunks, EB, 0, st>>>(V, d_mirror, d_deg, d_chunks, d_chunks + need_chunks, d_chunks + 2 * need_chunks);
    scan_u32<uint32_t>(st, d_chunks, need_chunks, d_chunks, d_bsum, d_small + 1);
    scan_u32<uint32_t>(st, d_chunks + need_chunks, need_chunks, d_chunks + need_chunks, d_bsum, d_small + 2);
    scan_u32<uint32_t>(st, d_chunks + 2 * need_chunks, need_chunks, d_chunks + 2 * need_chunks, d_bsum, d_small + 3);
    uint32_t h_small[16];
    HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    double acc0 = 0;
    if (lap.on) acc0 += lap.lap("  upload, degrees, imbalance counts");
    const uint32_t N = h_small[1], n_sm = h_small[3];
    if (h_small[2] != N) MTG_DIE("device_finish: internal error (missing in-edges %u != missing out-edges %u)", N, h_small[2]);
    const uint32_t delta = n_sm & 1u, n_sm_edges = (n_sm + 1) / 2;
    if (delta && N == 0)
        MTG_DIE("Have an uneven number of self-mirrors, but no other nodes with missing in edges. (implementation/mod.rs:496-498)");
    // every step removes two units; an odd rest cannot be paired: the reference's iterator runs dry (mod.rs:262)
    const uint32_t n_steps = (N - delta) / 2;
    const uint64_t brk_cap = (uint64_t)n_sm_edges + n_steps + ((N - delta) & 1u) + 2;
    const uint64_t E_cap = E0 + 2 * n_pairs + 2 * brk_cap;
    if (E_cap >= 0xFFFFFFFEull) MTG_DIE("device_finish: %llu darts do not fit 32-bit ids", (unsigned long long)E_cap);

    Buf b_from, b_pw, b_sm, b_anode, b_bnode;
    uint32_t *d_from = b_from.alloc<uint32_t>(st, E_cap);
    uint32_t *d_pw = b_pw.alloc<uint32_t>(st, n_pairs);
    if (E0) HIP_CHECK(hipMemcpyAsync(d_from, d_from0, E0 * 4, hipMemcpyDeviceToDevice, st));
    if (n_pairs) pair_darts_kernel<<<grid_for(n_pairs), EB, 0, st>>>(d_pairs, n_pairs, d_mirror, d_from + E0, d_pw);
    const uint64_t first_brk = E0 + 2 * n_pairs;
    uint32_t *d_sm = b_sm.alloc<uint32_t>(st, n_sm);
    uint32_t *d_anode = b_anode.alloc<uint32_t>(st, N), *d_bnode = b_bnode.alloc<uint32_t>(st, N);
    need_emit_kernel<<<(unsigned)need_chunks, EB, 0, st>>>(V, d_mirror, d_deg, d_chunks, d_chunks + need_chunks, d_chunks + 2 * need_chunks, d_small + 1,
                                                           d_cin, d_cout, d_pin, d_pout, d_sm, d_anode, d_bnode);
    if (n_sm) sm_emit_kernel<<<grid_for(n_sm_edges), EB, 0, st>>>(n_sm, d_sm, d_bnode, d_mirror, d_from + first_brk);
    uint32_t s_star = n_steps;
    if (n_steps) {
        HIP_CHECK(hipMemcpyAsync(d_small + 4, &s_star, 4, hipMemcpyHostToDevice, st));
        zip_check_kernel<<<grid_for(n_steps), EB, 0, st>>>(n_steps, delta, N, d_mirror, d_cin, d_cout, d_pin, d_pout, d_anode, d_bnode, d_small + 4);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        s_star = h_small[4];
        if (s_star) zip_emit_kernel<<<grid_for(s_star), EB, 0, st>>>(s_star, delta, d_mirror, d_anode, d_bnode, d_from + first_brk + 2 * (uint64_t)n_sm_edges);
    }
    HIP_CHECK(hipGetLastError());
    if (lap.on) { HIP_CHECK(hipStreamSynchronize(st)); acc0 += lap.lap("  unit orders, zip check + emit"); }
    uint64_t n_brk = (uint64_t)n_sm_edges + s_star;
    if ((uint64_t)N - delta - 2 * (uint64_t)s_star > 0) {  // irregular rest: the reference's loop over the residual counters
        Buf b_resid, b_rflag, b_rpos, b_rnode, b_rcnt;
        uint32_t *d_resid = b_resid.alloc<uint32_t>(st, V), *d_rflag = b_rflag.alloc<uint32_t>(st, V), *d_rpos = b_rpos.alloc<uint32_t>(st, V);
        residual_kernel<<<grid_for(V), EB, 0, st>>>(V, s_star, delta, N, d_mirror, d_cout, d_pin, d_pout, d_resid, d_rflag, d_small);
        scan_u32<uint32_t>(st, d_rflag, V, d_rpos, d_bsum, d_small + 5);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (h_small[0]) MTG_DIE("device_finish: internal error (Euleriser prefix removed a unit twice)");
        const uint32_t n_res = h_small[5];
        uint32_t *d_rnode = b_rnode.alloc<uint32_t>(st, n_res), *d_rcnt = b_rcnt.alloc<uint32_t>(st, n_res);
        residual_compact_kernel<<<grid_for(V), EB, 0, st>>>(V, d_resid, d_rpos, d_rnode, d_rcnt);
        std::vector<uint32_t> r_node(n_res), r_cnt(n_res), t_out, t_in;
        HIP_CHECK(hipMemcpyAsync(r_node.data(), d_rnode, (uint64_t)n_res * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(r_cnt.data(), d_rcnt, (uint64_t)n_res * 4, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        eulerise_tail(g, r_node, r_cnt, t_out, t_in);
        if (n_brk + t_out.size() > brk_cap) MTG_DIE("device_finish: internal error (more breaking edges than units)");
        std::vector<uint32_t> t_from(2 * t_out.size());
        for (size_t i = 0; i < t_out.size(); i++) {
            t_from[2 * i] = t_out[i];
            t_from[2 * i + 1] = g.mirror[t_in[i]];
        }
        if (!t_from.empty()) HIP_CHECK(hipMemcpyAsync(d_from + first_brk + 2 * n_brk, t_from.data(), t_from.size() * 4, hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        n_brk += t_out.size();
    }
    if (lap.on)
        std::fprintf(stderr, "[mtg] device_finish:   Euleriser: %u units, %u self-mirror nodes, %u of %u steps by the parallel prefix, %llu by the sequential tail\n",
                     N, n_sm, s_star, n_steps, (unsigned long long)(n_brk - n_sm_edges - s_star));
    const uint64_t E = first_brk + 2 * n_brk, n_dummy = E - E0;
    // (the counters and their prefixes stay until the buckets are built: the breaking darts of the regular steps are bucketed
    // arithmetically from them, finish_device.hpp)
    const ZipBuckets zip{first_brk + 2 * (uint64_t)n_sm_edges, s_star, delta, N, d_cin, d_cout, d_pin, d_pout};
    b_deg.release(); b_chunks.release();
    b_sm.release(); b_anode.release(); b_bnode.release(); b_pairs.release();
    HIP_CHECK(hipStreamSynchronize(st));
    acc0 += lap.lap("upload + insertion + Euleriser");
    if (times_out) { times_out[0] = acc0; times_out[5] = (double)n_brk; }

    // ---- the dummy edges join the host graph's edge arrays (unlinked: a host stage that walks adjacency links them first).
    // With the device decomposition nothing on the host needs them before the call returns: the 33 bytes per dart are then written
    // by a thread of their own while the GPU decomposes and cuts (joined at the end). ----
    Buf b_to;
    std::thread append_thread;
    std::atomic<bool> pw_ready{false};  // the host copy of the resident pairs' weights is complete (a sink flattens with it)
    hipEvent_t ev_heads = nullptr;
    std::unique_ptr<uint32_t[]> h_pw;  // resident pairs: their weights for the host graph (4 of the 16 bytes of a pair); not zero-filled
    {
        bool long_pair = false;  // (resident pairs come from the claim loop: every distance is <= k - 1)
        for (uint64_t i = 0; pairs && i < n_pairs && !long_pair; i++) long_pair = pairs[i].distance >= k;
        uint32_t *d_to = b_to.alloc<uint32_t>(st, n_dummy);
        if (n_dummy) head_kernel<<<grid_for(n_dummy), EB, 0, st>>>(d_from, d_mirror, E0, E, d_to);
        HIP_CHECK(hipGetLastError());
        sev.mark(1, st);
        g.append_unlinked(n_dummy);
        if (!pairs) h_pw.reset(new uint32_t[std::max<uint64_t>(n_pairs, 1)]);
        uint32_t *h_pw_p = h_pw.get();
        auto fill = [&g, pairs, h_pw_p, n_pairs, k, E0, n_dummy]() {  // weights, dummy ids (1-based, :681 / mod.rs:573)
            parallel_ranges(n_dummy / 2, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; i++) {
                    const uint64_t w = i < n_pairs ? (pairs ? pairs[i].distance : (uint64_t)h_pw_p[i]) : k;
                    g.w_biedge[E0 / 2 + i] = w;  // (the payload is per biedge: host_graph.hpp)
                    g.dummy_tail[i] = i + 1;
                }
            });
        };
        // copies into the (pageable) host arrays keep the calling thread until they are done: 0.45 GB = 10 ms at 2^27
        auto download = [&g, h_pw_p, d_from, d_to, d_pw, pairs, n_pairs, E0, n_dummy](hipStream_t s) {
            if (!pairs && n_pairs) HIP_CHECK(hipMemcpyAsync(h_pw_p, d_pw, n_pairs * 4, hipMemcpyDeviceToHost, s));
            if (n_dummy) {
                HIP_CHECK(hipMemcpyAsync(g.e_from.data() + E0, d_from + E0, n_dummy * 4, hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipMemcpyAsync(g.e_to.data() + E0, d_to, n_dummy * 4, hipMemcpyDeviceToHost, s));
            }
            HIP_CHECK(hipStreamSynchronize(s));
        };
        if (euler_mode == MTG_EULER_DEVICE && n_dummy >= (1u << 20)) {
            // nothing on the host needs the dummy edges before the call returns: a thread of its own brings them down on the
            // device's side stream and writes the 33 bytes per dart while the GPU decomposes and cuts (joined at the end)
            HIP_CHECK(hipEventCreateWithFlags(&ev_heads, hipEventDisableTiming));
            HIP_CHECK(hipEventRecord(ev_heads, st));  // (behind the dart, weight and head kernels)
            // (through the pinned ring, whose slices host threads copy out: 0.3 ms better than the runtime's own staging of a copy
            // into pageable memory. Either way the copies are wide KERNELS on this platform -- __amd_rocclr_copyBuffer in the traces --
            // and the decomposition's first kernels run two to six times longer beside them, `tools/kernel_timeline.py`; a copy
            // kernel of our own with 16 to 256 workgroups was measured and is worse at every width: 29.5-34 ms for the
            // decomposition against 26.5.)
            uint32_t *h_from = g.e_from.data() + E0, *h_to = g.e_to.data() + E0;
            append_thread = std::thread([fill, ev_heads, device_id, h_from, h_to, h_pw_p, d_from, d_to, d_pw, pairs, n_pairs, E0, n_dummy, &pw_ready]() {
                HIP_CHECK(hipSetDevice(device_id));
                hipStream_t side = finish_side_stream(device_id);
                HIP_CHECK(hipStreamWaitEvent(side, ev_heads, 0));
                if (!pairs && n_pairs) download_sliced(h_pw_p, d_pw, n_pairs * 4, side, device_id);
                pw_ready.store(true, std::memory_order_release);
                // (the weights and dummy ids need nothing but the pair weights: written beside the two downloads, which wait for PCIe --
                // since the tigs come straight from the pairing the GPU stages are through in 14 ms at 2^27, and downloads followed by
                // the fill took longer than that)
                std::thread filler(fill);
                download_sliced(h_from, d_from + E0, n_dummy * 4, side, device_id);
                download_sliced(h_to, d_to, n_dummy * 4, side, device_id);
                filler.join();
            });
        } else {
            download(st);
            pw_ready.store(true, std::memory_order_release);
            fill();
        }
        g.first_breaking_edge = first_brk;
        g.breaking_weight = k;
        g.dummies_canonical = !long_pair;
    }
    // The result arrays are known up to a bound once the Euleriser is through (kept edges <= original biedges + matched pairs; tigs
    // <= dummy biedges + closed walks): they are reserved now and touched by a thread of their own while the GPU decomposes and
    // cuts -- a one-shot caller's tigs land in memory that has never been touched (0.47 GB at 2^27: the cold step spent 50 ms of
    // page faults inside the download), a caller that iterates gets recycled blocks and the touching costs nothing.
    std::thread prefault_thread;
    {
        const uint64_t bound_edges = E0 / 2 + n_pairs, bound_tigs = n_dummy / 2 + 4096;
        if (!sink && !resident_out) {
            tigs.edges.resize(bound_edges);  // (PodVec: no element is written; shrunk to the real sizes after the cut)
            tigs.limits.resize(bound_tigs);
        }
        // (a sink's arrays are sized as clib.rs:332-348: 2 E0 / 2 E0 / E0 entries. Its first array is touched here, the second by the
        // same threads through the offset below; a caller that passes touched memory loses nothing)
        if (!resident_out && !(sink && sink->pretoucher) && (bound_edges * 4 + bound_tigs * 8) >= (64u << 20)) {
            char *pe = sink ? reinterpret_cast<char *>(sink->edge_out) : reinterpret_cast<char *>(tigs.edges.data());
            char *pl = sink ? reinterpret_cast<char *>(sink->limits_out) : reinterpret_cast<char *>(tigs.limits.data());
            char *pi = sink ? reinterpret_cast<char *>(sink->insert_out) : nullptr;
            const uint64_t be = sink ? bound_edges * 8 : bound_edges * 4, bl = std::min<uint64_t>(bound_tigs, std::max<uint64_t>(E0, 1)) * 8;
            prefault_thread = std::thread([pe, pl, pi, be, bl]() {
                if (pi) parallel_ranges((be + 4095) / 4096, [&](uint64_t lo, uint64_t hi) {
                    for (uint64_t pg = lo; pg < hi; pg++) *(volatile char *)(pi + pg * 4096) = 0;
                });
                parallel_ranges((be + bl + 4095) / 4096, [&](uint64_t lo, uint64_t hi) {
                    for (uint64_t pg = lo; pg < hi; pg++) {
                        const uint64_t off = pg * 4096;
                        volatile char *p = off < be ? pe + off : pl + (off - be);
                        if (off < be || off - be < bl) *p = 0;
                    }
                });
            });
        }
    }
    if (times_out) times_out[1] = lap.lap("host graph: dummy edges");
    if (E == 0) {
        if (prefault_thread.joinable()) prefault_thread.join();
        tigs.edges.resize(0);
        tigs.limits.resize(0);
        return tigs;
    }

    // ---- Euler bicycles ----
    Buf b_cyc, b_clen, b_cbase;
    uint32_t n_cycles = 0;
    double kernel_ms = 0, acc2 = 0;
    // (device order: the tigs straight from the pairing when the graph allows it -- cut_first_device.hip; the closed walks otherwise)
    Buf b_te, b_tl;
    uint64_t n_kept = 0, n_tigs = 0;
    bool have_tigs = false;
    CutFirstStats cf_stats;
    if (euler_mode == MTG_EULER_DEVICE) {
        if (!(finish_tuning().flags.load() & FT_NO_CUT_FIRST)) {
            sev.mark(2, st);
            have_tigs = device_cut_first(st, d_from, d_mirror, E, V, E0, first_brk, d_pw, d_row0, d_adj0, &zip, b_te, b_tl, &n_kept, &n_tigs, &cf_stats);
            sev.mark(3, st);
        }
        if (!have_tigs) device_euler_decompose(st, d_from, d_mirror, E, V, b_cyc, b_clen, b_cbase, &n_cycles, &kernel_ms, d_row0, d_adj0, E0, &zip);
        b_cin.release(); b_cout.release(); b_pin.release(); b_pout.release();
    } else {
        Walks cycles;
        KeepAwake keep_awake((finish_tuning().flags.load() & FT_KEEP_AWAKE) != 0, device_id);
        {
            Buf b_row, b_adj, b_need, b_off, b_tot, b_nodes, b_xe, b_xt;
            uint32_t *d_row = b_row.alloc<uint32_t>(st, V + 1), *d_adj = b_adj.alloc<uint32_t>(st, E);
            sev.mark(2, st);
            if (d_row0) device_build_buckets_merged(st, d_from, d_mirror, E0, E, V, d_row0, d_adj0, d_row, d_adj, &zip);
            else device_build_buckets(st, d_from, E, V, d_row, d_adj, nullptr);
            b_cin.release(); b_cout.release(); b_pin.release(); b_pout.release();
            uint32_t *d_need = b_need.alloc<uint32_t>(st, V), *d_off = b_off.alloc<uint32_t>(st, V), *d_tot = b_tot.alloc<uint32_t>(st, 1);
            lean_ext_kernel<<<grid_for(V), EB, 0, st>>>(V, d_row, d_need, d_small);
            scan_u32<uint32_t>(st, d_need, V, d_off, d_bsum, d_tot);
            uint32_t ext_total = 0;
            HIP_CHECK(hipMemcpyAsync(&ext_total, d_tot, 4, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            if (h_small[0] & 8u) {  // a node with more than 65535 out-edges: the simple host formulation (never a de Bruijn graph)
                b_row.release(); b_adj.release(); b_need.release(); b_off.release();
                cycles = euler_cycles_generic(g);
            } else {
                struct EventHolder {
                    hipEvent_t e = nullptr;
                    EventHolder() { HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
                    ~EventHolder() { if (e) (void)hipEventDestroy(e); }
                    hipEvent_t get() const { return e; }
                } ev_lean;
                LeanNode *d_nodes = b_nodes.alloc<LeanNode>(st, V);
                uint32_t *d_xe = b_xe.alloc<uint32_t>(st, ext_total), *d_xt = b_xt.alloc<uint32_t>(st, ext_total);
                lean_build_kernel<<<grid_for(V), EB, 0, st>>>(V, d_row, d_adj, d_from, d_mirror, d_off, d_nodes, d_xe, d_xt);
                HIP_CHECK(hipGetLastError());
                sev.mark(3, st);
                // (a download of the 32-byte records on the side stream waits for this: nothing else orders the two streams)
                HIP_CHECK(hipEventRecord(ev_lean.get(), st));
                std::vector<uint32_t> ext_eid(ext_total), ext_to(ext_total);
                if (ext_total) {
                    HIP_CHECK(hipMemcpyAsync(ext_eid.data(), d_xe, (uint64_t)ext_total * 4, hipMemcpyDeviceToHost, st));
                    HIP_CHECK(hipMemcpyAsync(ext_to.data(), d_xt, (uint64_t)ext_total * 4, hipMemcpyDeviceToHost, st));
                }
                // 256-byte records (two levels of copied adjacency: 2.6 steps per DRAM miss) while they fit comfortably, 128-byte
                // ones (one level: 1.9 steps per miss) up to 100 GB of them, the 32-byte records themselves beyond (one miss per
                // step, an eighth of the memory). Same walk either way; mtg_set_finish_tuning(records) overrides the choice
                // (speed / memory only).
                // The larger formats are only chosen when the host has room for them next to the walk's entry arrays (16 bytes
                // per dart) -- free memory as the kernel and the cgroup report it.
                const int rec = finish_tuning().records.load();  // (mtg_set_finish_tuning: 1 lean, 2 mid, 3 wide)
                const int tuning_flags = finish_tuning().flags.load();
                const long delay_us = finish_tuning().record_delay_us.load();  // (tests: slow arrival, so that small graphs take the 32-byte path too)
                const uint64_t room = host_available_bytes(), walk_arrays = E * 16;
                const bool wide = rec ? rec == 3 : (V * 256 <= (48ull << 30) && V * 256 + walk_arrays <= room / 4 * 3);
                const bool mid = rec ? rec == 2 : (!wide && V * 128 <= (100ull << 30) && V * (128 + 32) + walk_arrays <= room / 4 * 3);
                if (mid) {
                    // built on the GPU (one gather level) and brought down in slices through pageable memory: these are the
                    // graphs of a hundred gigabytes, where page-locking the arena would cost more than the copy
                    HugeBuf<EulerNode2> mbuf(V, &g.arena);
                    b_row.release(); b_adj.release(); b_need.release(); b_off.release();
                    const bool overlap_mid = !(tuning_flags & FT_NO_RECORD_OVERLAP);
                    const uint64_t slice = std::max<uint64_t>(1, (1ull << 30) / sizeof(EulerNode2));  // 1 GB of records at a time
                    auto build_and_download = [&](std::atomic<uint64_t> *arrived, long delay_us) {
                        Buf b_mid;
                        EulerNode2 *d_mid = b_mid.alloc<EulerNode2>(st, std::min<uint64_t>(V, slice));
                        for (uint64_t lo = 0; lo < V; lo += slice) {
                            const uint64_t n = std::min(slice, V - lo);
                            mid_build_slice_kernel<<<grid_for(n), EB, 0, st>>>(lo, n, V, d_nodes, d_mid);
                            HIP_CHECK(hipGetLastError());
                            HIP_CHECK(hipMemcpyAsync(mbuf.p + lo, d_mid, n * sizeof(EulerNode2), hipMemcpyDeviceToHost, st));
                            if (arrived) {
                                HIP_CHECK(hipStreamSynchronize(st));
                                if (delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
                                arrived->store(lo + n, std::memory_order_release);
                            }
                        }
                        HIP_CHECK(hipStreamSynchronize(st));
                    };
                    if (overlap_mid && V >= (1u << 16) && host_available_bytes() >= V * sizeof(LeanNode) + (4ull << 30)) {  // (room for the 32-byte records beside everything else)
                        // as for the 256-byte records below: the walk starts on the 32-byte records (23 GB at 2^30, down first) while
                        // the 128-byte ones (92 GB there: 6 s of a 100-s finish) are built and brought down slice by slice on a
                        // thread of their own
                        HugeBuf<LeanNode> lbuf(V, &g.arena);
                        download_sliced(lbuf.p, d_nodes, V * sizeof(LeanNode), st, device_id);
                        std::atomic<uint64_t> arrived{0};
                        std::thread mover([&build_and_download, &arrived, delay_us, device_id]() {
                            HIP_CHECK(hipSetDevice(device_id));
                            build_and_download(&arrived, delay_us);
                        });
                        acc2 += lap.lap("32-byte records down (the 128-byte ones follow beside the walk)");
                        cycles = euler_cycles_from_mid_arriving(mbuf.p, lbuf.p, &arrived, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena);
                        mover.join();
                        b_nodes.release(); b_xe.release(); b_xt.release();
                    } else {
                        build_and_download(nullptr, 0);
                        b_nodes.release(); b_xe.release(); b_xt.release();
                        acc2 += lap.lap("walk records, two levels (GPU) + download");
                        cycles = euler_cycles_from_mid(mbuf.p, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena);
                    }
                } else if (!wide) {
                    HugeBuf<LeanNode> nodes(V, &g.arena);
                    download_sliced(nodes.p, d_nodes, V * sizeof(LeanNode), st, device_id);
                    b_row.release(); b_adj.release(); b_need.release(); b_off.release(); b_nodes.release(); b_xe.release(); b_xt.release();
                    acc2 += lap.lap("walk records (GPU) + download");
                    cycles = euler_cycles_lean(nodes.p, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena);
                } else {
                    // The wide records live in the graph's arena. A mapping that comes back for its second call belongs to a
                    // caller that iterates: it is page-locked once (hipHostRegister) and the records come down in one copy.
                    // A first (or only) call does not pin (pinning 23 GB of fresh memory costs more than it saves a one-shot
                    // caller): its records go through the pinned ring of download_sliced.
                    HugeBuf<EulerNode3> wbuf(V, &g.arena);
                    bool pinned = false;
                    const unsigned uses = g.arena.uses_of(wbuf.p, &pinned);
                    const bool pin_ok = !(tuning_flags & FT_NO_PIN);
                    if (!pinned && uses >= 2 && pin_ok) {
                        HugeArena::unpin_hook() = [](void *p) { (void)hipHostUnregister(p); };
                        if (hipHostRegister(wbuf.p, wbuf.bytes, hipHostRegisterDefault) == hipSuccess) {
                            g.arena.set_pinned(wbuf.p);
                            pinned = true;
                        } else (void)hipGetLastError();
                    }
                    {
                        // all three levels on the GPU; into a page-locked arena the records travel in one copy, into a fresh
                        // one through the pinned ring (host threads copy the slices out): 0.5 s either way at 2^27, where
                        // filling levels two and three with host threads took 1.2 s
                        Buf b_wide;
                        EulerNode3 *d_wide = b_wide.alloc<EulerNode3>(st, V);
                        wide_build_kernel<<<grid_for(V), EB, 0, st>>>(V, d_nodes, d_wide);
                        HIP_CHECK(hipGetLastError());
                        sev.mark(3, st);
                        const bool overlap_ok = !(tuning_flags & FT_NO_RECORD_OVERLAP);
                        if (overlap_ok && V >= (1u << 16) && host_available_bytes() >= V * sizeof(LeanNode) + (4ull << 30)) {  // (room for the 32-byte records beside everything else)
                            // The walk starts while the records still cross PCIe (23 GB = 0.4-0.5 s at 2^27, a twentieth of the step): they
                            // arrive in node order, slice by slice, `arrived` says how far they have come, and a step that needs a record
                            // beyond that mark takes the node's 32-byte record instead (euler_fast.cpp) -- those came down first, on the
                            // side stream, while wide_build_kernel ran. Into a page-locked arena (a graph's second call and later) the
                            // slices are plain copies and a watcher thread follows their events; into a fresh one they go through the
                            // pinned ring on a thread of their own, whose copying threads report every slice they have moved out.
                            HugeBuf<LeanNode> lbuf(V, &g.arena);
                            hipStream_t side = finish_side_stream(device_id);
                            HIP_CHECK(hipStreamWaitEvent(side, ev_lean.get(), 0));  // (lean_build_kernel, which writes d_nodes, runs on `st`)
                            download_sliced(lbuf.p, d_nodes, V * sizeof(LeanNode), side, device_id);
                            HIP_CHECK(hipStreamSynchronize(st));  // (the spill arrays' copies and wide_build_kernel: done by now, the 32-byte records took longer)
                            std::atomic<uint64_t> arrived{0};
                            std::vector<hipEvent_t> evs;
                            std::vector<uint64_t> upto;
                            std::thread mover;
                            if (pinned) {
                                constexpr int N_SLICES = 64;
                                const uint64_t per = (V + N_SLICES - 1) / N_SLICES;
                                for (uint64_t lo = 0; lo < V; lo += per) {
                                    const uint64_t n = std::min<uint64_t>(per, V - lo);
                                    HIP_CHECK(hipMemcpyAsync(wbuf.p + lo, d_wide + lo, n * sizeof(EulerNode3), hipMemcpyDeviceToHost, st));
                                    hipEvent_t e;
                                    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToSystem));
                                    HIP_CHECK(hipEventRecord(e, st));
                                    evs.push_back(e);
                                    upto.push_back(lo + n);
                                }
                                mover = std::thread([&evs, &upto, &arrived, delay_us]() {
                                    for (size_t i = 0; i < evs.size(); i++) {
                                        HIP_CHECK(hipEventSynchronize(evs[i]));
                                        if (delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
                                        arrived.store(upto[i], std::memory_order_release);
                                    }
                                });
                            } else {
                                EulerNode3 *dst = wbuf.p;
                                const uint64_t n_nodes = V;
                                mover = std::thread([dst, d_wide, n_nodes, st, device_id, &arrived, delay_us]() {
                                    HIP_CHECK(hipSetDevice(device_id));
                                    const std::function<void(size_t)> progress = [&arrived, n_nodes, delay_us](size_t bytes_done) {
                                        if (delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(delay_us));
                                        const uint64_t whole = bytes_done / sizeof(EulerNode3);  // (a record cut by a slice boundary counts with the next slice)
                                        arrived.store(whole < n_nodes ? whole : n_nodes, std::memory_order_release);
                                    };
                                    download_sliced_with(d_wide, n_nodes * sizeof(EulerNode3), st, device_id,
                                                         [dst](size_t off, const char *src, size_t n) { std::memcpy((char *)dst + off, src, n); }, &progress);
                                    arrived.store(n_nodes, std::memory_order_release);
                                });
                            }
                            acc2 += lap.lap("walk records, all levels (GPU); 32-byte records down");
                            cycles = euler_cycles_from_wide_arriving(wbuf.p, lbuf.p, &arrived, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena);
                            mover.join();
                            HIP_CHECK(hipStreamSynchronize(st));
                            for (hipEvent_t e : evs) HIP_CHECK(hipEventDestroy(e));
                            b_wide.release();
                            b_row.release(); b_adj.release(); b_need.release(); b_off.release(); b_nodes.release(); b_xe.release(); b_xt.release();
                        } else {
                        if (pinned) {
                            HIP_CHECK(hipMemcpyAsync(wbuf.p, d_wide, V * sizeof(EulerNode3), hipMemcpyDeviceToHost, st));
                            HIP_CHECK(hipStreamSynchronize(st));
                        } else download_sliced(wbuf.p, d_wide, V * sizeof(EulerNode3), st, device_id);
                        b_wide.release();
                        b_row.release(); b_adj.release(); b_need.release(); b_off.release(); b_nodes.release(); b_xe.release(); b_xt.release();
                        acc2 += lap.lap("walk records, all levels (GPU) + download");
                        cycles = euler_cycles_from_wide(wbuf.p, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena);
                        }
                    }
                }
            }
        }
        n_cycles = (uint32_t)cycles.limits.size();
        if (cycles.limits.size() >= 0xFFFFFFFFull) MTG_DIE("device_finish: too many cycles");
        uint32_t *d_cyc = b_cyc.alloc<uint32_t>(st, E / 2);
        uint32_t *d_clen = b_clen.alloc<uint32_t>(st, n_cycles), *d_cbase = b_cbase.alloc<uint32_t>(st, n_cycles);
        std::vector<uint32_t> clen(n_cycles), cbase(n_cycles);
        for (uint32_t c = 0; c < n_cycles; c++) {
            const uint64_t lo = c ? cycles.limits[c - 1] : 0;
            cbase[c] = (uint32_t)lo;
            clen[c] = (uint32_t)(cycles.limits[c] - lo);
        }
        if (cycles.edges.size() != E / 2) MTG_DIE("device_finish: internal error (closed walks cover %zu of %llu biedges)", cycles.edges.size(), (unsigned long long)(E / 2));
        HIP_CHECK(hipMemcpyAsync(d_cyc, cycles.edges.data(), (E / 2) * 4, hipMemcpyHostToDevice, st));
        if (n_cycles) {
            HIP_CHECK(hipMemcpyAsync(d_clen, clen.data(), (uint64_t)n_cycles * 4, hipMemcpyHostToDevice, st));
            HIP_CHECK(hipMemcpyAsync(d_cbase, cbase.data(), (uint64_t)n_cycles * 4, hipMemcpyHostToDevice, st));
        }
        HIP_CHECK(hipStreamSynchronize(st));
    }
    acc2 += lap.lap("Euler bicycles");
    if (times_out) { times_out[2] = acc2; times_out[4] = kernel_ms; }

    // ---- rotate + cut ----
    {
        const uint64_t n = E / 2;
        const uint32_t *d_cyc = b_cyc.as<uint32_t>(), *d_clen = b_clen.as<uint32_t>(), *d_cbase = b_cbase.as<uint32_t>();
        Buf b_rotkey, b_ck, b_ce;
        sev.mark(4, st);
        if (!have_tigs) {
        const uint64_t n_chunks = (n + CUT_CHUNK - 1) / CUT_CHUNK;
        unsigned long long *d_rotkey = b_rotkey.alloc<unsigned long long>(st, n_cycles);
        uint32_t *d_ck = b_ck.alloc<uint32_t>(st, n_chunks), *d_ce = b_ce.alloc<uint32_t>(st, n_chunks);
        HIP_CHECK(hipMemsetAsync(d_rotkey, 0, (uint64_t)std::max<uint32_t>(n_cycles, 1) * 8, st));
        CutIds ids{(uint32_t)E0, (uint32_t)first_brk};
        rotation_kernel<<<(unsigned)n_chunks, EB, 0, st>>>(d_cyc, n, d_cbase, n_cycles, ids, (uint32_t)k, d_pw, d_rotkey);
        cut_count_kernel<<<(unsigned)n_chunks, EB, 0, st>>>(d_cyc, n, d_cbase, d_clen, n_cycles, d_rotkey, ids, d_ck, d_ce);
        scan_u32<uint32_t>(st, d_ck, n_chunks, d_ck, d_bsum, d_small + 7);
        scan_u32<uint32_t>(st, d_ce, n_chunks, d_ce, d_bsum, d_small + 8);
        HIP_CHECK(hipMemcpyAsync(h_small, d_small, 64, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        n_kept = h_small[7]; n_tigs = h_small[8];
        uint32_t *d_te0 = b_te.alloc<uint32_t>(st, n_kept);
        uint32_t *d_tl0 = b_tl.alloc<uint32_t>(st, n_tigs);
        cut_emit_kernel<<<(unsigned)n_chunks, EB, 0, st>>>(d_cyc, n, d_cbase, d_clen, n_cycles, d_rotkey, ids, d_ck, d_ce, d_te0, d_tl0);
        HIP_CHECK(hipGetLastError());
        }
        uint32_t *d_te = b_te.as<uint32_t>(), *d_tl = b_tl.as<uint32_t>();
        sev.mark(5, st);
        if (prefault_thread.joinable()) prefault_thread.join();
        if (sink && sink->pretoucher && sink->pretoucher->joinable()) sink->pretoucher->join();  // no helper write after a result write
        if (sink) {
            // clib.rs:393-407 on the way out of the download ring: an original edge e is unitig e >> 1, forwards iff e is even
            // (host_graph.hpp); a dummy edge inside a tig is a matched pair, whose weight is its distance
            while (!pw_ready.load(std::memory_order_acquire)) std::this_thread::yield();  // (resident pairs: their weights came down first, on the side stream)
            const uint32_t E0u = (uint32_t)E0;
            const uint32_t *pw = h_pw.get();
            int64_t *eo = sink->edge_out;
            uint64_t *io = sink->insert_out;
            // (16 bytes written per 4 read, never read again here: streaming stores -- no read-for-ownership of the caller's lines)
            // (measured and dropped: groups of eight edges without a dummy through a loop without branches or lookups, which the
            // compiler vectorises -- 22-42 ms against 25-27 ms: the expansion waits for memory, not for instructions)
            auto expand = [=](size_t first, const uint32_t *e, size_t n) {
                for (size_t i = 0; i < n; i++) {
                    const uint32_t x = e[i];
                    int64_t ev;
                    uint64_t iv;
                    if (x < E0u) {
                        ev = (x & 1u) ? -(int64_t)(x >> 1) : (int64_t)(x >> 1);
                        iv = 0;
                    } else {
                        const uint64_t b = (x - E0u) >> 1;
                        ev = 0;
                        iv = b < n_pairs ? (pairs ? pairs[b].distance : (uint64_t)pw[b]) : k;
                    }
                    __builtin_nontemporal_store(ev, &eo[first + i]);
                    __builtin_nontemporal_store(iv, &io[first + i]);
                }
            };
            const auto t_edges = std::chrono::steady_clock::now();
            if (n_kept * 4 < hu::PLAIN_COPY_LIMIT) {  // (few tigs: one copy, one thread)
                std::vector<uint32_t> tmp(n_kept);
                if (n_kept) HIP_CHECK(hipMemcpyAsync(tmp.data(), d_te, n_kept * 4, hipMemcpyDeviceToHost, st));
                HIP_CHECK(hipStreamSynchronize(st));
                expand(0, tmp.data(), n_kept);
            } else
                download_sliced_with(d_te, n_kept * 4, st, device_id,
                                     [&expand](size_t off, const char *src, size_t n) { expand(off / 4, reinterpret_cast<const uint32_t *>(src), n / 4); },
                                     nullptr, 16);
            if (std::getenv("MTG_DEBUG")) std::fprintf(stderr, "[mtg] device_finish:   tig edges flattened into the caller's arrays (%.2f GB written)  %.3f ms\n", n_kept * 16 / 1e9,
                                                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_edges).count());
            static const bool dbg_sink = std::getenv("MTG_DEBUG") != nullptr;
            const auto t_sink = std::chrono::steady_clock::now();
            download_sliced_widen(sink->limits_out, d_tl, n_tigs, st, device_id, 16);
            if (dbg_sink) std::fprintf(stderr, "[mtg] device_finish:   tig ends into the caller's array (%.2f GB written)  %.3f ms\n", n_tigs * 8 / 1e9,
                                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sink).count());
            sink->n_tigs = n_tigs;
            sink->n_edges = n_kept;
        } else if (resident_out) {
            // the tigs stay where the cutter wrote them: the two arrays leave their Bufs and belong to the caller's object
            HIP_CHECK(hipStreamSynchronize(st));
            ResidentTigs *r = new ResidentTigs();
            r->device = device_id;
            r->d_edges = d_te; r->d_limits = d_tl;
            r->n_edges = n_kept; r->n_tigs = n_tigs;
            b_te.p = nullptr; b_tl.p = nullptr;
            *resident_out = r;
        } else {
            tigs.edges.resize(n_kept);
            tigs.limits.resize(n_tigs);
            download_sliced(tigs.edges.data(), d_te, n_kept * 4, st, device_id);
            // (the limits cross PCIe as 32-bit words and are widened by the host threads that empty the download ring)
            download_sliced_widen(tigs.limits.data(), d_tl, n_tigs, st, device_id);
        }
    }
    if (times_out) times_out[3] = lap.lap("rotate + cut + download");
    if (append_thread.joinable()) {
        append_thread.join();
        HIP_CHECK(hipEventDestroy(ev_heads));
        const double waited = lap.lap("host graph: dummy edges (rest, after the GPU stages)");
        if (times_out) times_out[1] += waited;
    }
    b_to.release();
    b_from.release(); b_pw.release(); b_clen.release(); b_cbase.release(); b_bsum.release(); b_small.release();
    HIP_CHECK(hipStreamSynchronize(st));
    if (times_out) {  // [6] insertion + Euleriser, [7] buckets + walk records, [8] rotate + cut: GPU ms; [9] darts, [10] units, [11] closed walks
        times_out[6] = sev.ms(0, 1);
        times_out[7] = sev.ms(2, 3);
        if (euler_mode == MTG_EULER_DEVICE) { times_out[4] += times_out[7]; times_out[7] = 0; }  // (cut first: the stretch passes take the decomposition's place; an attempt that fell back to the closed walks is booked with them)
        times_out[8] = sev.ms(4, 5);
        times_out[9] = (double)E;
        times_out[10] = (double)N;
        times_out[11] = (double)n_cycles;
    }
    finish_trim(device_id, E * 40 + V * 28);
    {
        static const bool dbg = std::getenv("MTG_DEBUG") != nullptr;
        if (dbg) {
            DeviceArena &a = device_arena(device_id);
            std::lock_guard<std::mutex> lock(a.m);
            std::fprintf(stderr, "[mtg] device_finish: arena after the call: %.2f GB in %zu chunk(s), %llu taken from the driver so far, %.2f GB live\n", a.chunk_bytes / 1e9,
                         a.chunks.size(), (unsigned long long)a.n_chunk_allocs, a.live_bytes / 1e9);
        }
    }
    return tigs;
}

}  // namespace mtg
