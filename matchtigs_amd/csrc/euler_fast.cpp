// euler_fast.cpp -- Euler bicycle decomposition (bigraph 5.0.1 compute_minimum_bidirected_eulerian_cycle_decomposition,
// call site /root/reference/src/implementation/greedytigs/mod.rs:722), latency-optimised host formulation.
//
// Same sequences as the literal algorithm (see host_pipeline.cpp, euler_cycles_generic, for the policy and for how
// "rotate to x and append W" becomes a splice). On the bench graph the decomposition is ONE closed walk of ~11.6 M
// biedges plus a handful of splices, i.e. a single chain of dependent random memory accesses -- the only lever is the
// number of DRAM misses on that chain:
//   v0  used[] / e_to / cursor / e_next_out / e_from ...  ~7 misses per biedge   4.2 s  (profiles/r01_*)
//   v1  one 32-byte record per node (cursor + 3 inline adjacency entries), used-bitmap      ~1 miss per biedge, 1.5 s
//   v2  (this file) 128-byte records that also carry, for each of the 3 inline out-edges, the inline adjacency of the
//       edge's HEAD node: after one miss the walk knows where it can go from the next node too  ->  ~1 miss per 2 biedges.
// Record build and sub-adjacency fill are embarrassingly parallel gathers and run on host threads; the walk itself is
// inherently sequential (each step depends on every edge used so far).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "host_graph.hpp"
#include "hugebuf.hpp"
#include "parallel.hpp"

namespace mtg {

Walks euler_cycles_generic(const HostGraph &g);

namespace {
struct alignas(128) EulerNode2 {
    uint32_t eid[3];         // adjacency positions 0..2 in iteration order (newest edge first)
    uint32_t to[3];
    uint32_t ext_begin;      // spill entries for positions 3..deg-1
    uint16_t deg;
    uint16_t pos;            // positions < pos are known to be used
    uint32_t sub_eid[3][3];  // inline adjacency of to[j] (valid if sub_deg[j] != 0xFF)
    uint32_t sub_to[3][3];
    uint8_t sub_deg[3];
    uint8_t pad[21];
};
static_assert(sizeof(EulerNode2) == 128, "EulerNode2 must be 128 bytes");
}  // namespace

Walks euler_cycles(const HostGraph &g) {
    const uint64_t E = g.edge_count();
    const uint64_t V = g.node_count();
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    for (uint64_t n = 0; n < V; n++)
        if (g.out_deg[n] > 65535) return euler_cycles_generic(g);  // not a de Bruijn graph: simple formulation

    static const bool dbg_t = std::getenv("MTG_DEBUG") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin;  // the walk is one latency-bound thread: keep it next to the memory it chases through

    // ---- records ----
    HugeBuf<EulerNode2> nodes(V);
    std::vector<uint32_t> ext_begin(V + 1, 0);
    {
        uint64_t ext_total = 0;
        for (uint64_t n = 0; n < V; n++) {
            ext_begin[n] = (uint32_t)ext_total;
            if (g.out_deg[n] > 3) ext_total += g.out_deg[n] - 3;
            if (ext_total >= NONE) MTG_DIE("adjacency spill too large");
        }
        ext_begin[V] = (uint32_t)ext_total;
    }
    std::vector<uint32_t> ext_eid(ext_begin[V]), ext_to(ext_begin[V]);
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {  // phase A: own adjacency, newest first (the petgraph order)
        for (uint64_t n = lo; n < hi; n++) {
            EulerNode2 &r = nodes[n];
            r.deg = (uint16_t)g.out_deg[n];
            r.pos = 0;
            r.ext_begin = ext_begin[n];
            r.eid[0] = r.eid[1] = r.eid[2] = NONE;
            r.to[0] = r.to[1] = r.to[2] = NONE;
            r.sub_deg[0] = r.sub_deg[1] = r.sub_deg[2] = 0xFF;
            uint32_t i = 0;
            for (uint32_t e = g.head_out[n]; e != NONE; e = g.e_next_out[e], i++) {
                if (i < 3) { r.eid[i] = e; r.to[i] = g.e_to[e]; }
                else { ext_eid[r.ext_begin + i - 3] = e; ext_to[r.ext_begin + i - 3] = g.e_to[e]; }
            }
        }
    });
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {  // phase B: inline adjacency of each inline edge's head node
        for (uint64_t n = lo; n < hi; n++) {
            EulerNode2 &r = nodes[n];
            const uint32_t d = r.deg < 3 ? r.deg : 3;
            for (uint32_t j = 0; j < d; j++) {
                const EulerNode2 &w = nodes[r.to[j]];
                if (w.deg <= 3) {
                    r.sub_deg[j] = (uint8_t)w.deg;
                    for (uint32_t q = 0; q < 3; q++) { r.sub_eid[j][q] = w.eid[q]; r.sub_to[j][q] = w.to[q]; }
                }
            }
        }
    });
    const auto t_built = std::chrono::steady_clock::now();

    std::vector<uint64_t> used((E / 2 + 63) / 64 + 1, 0);
    auto is_used = [&](uint32_t e) -> bool { return (used[(e >> 1) >> 6] >> ((e >> 1) & 63)) & 1ull; };
    auto set_used = [&](uint32_t e) { used[(e >> 1) >> 6] |= 1ull << ((e >> 1) & 63); };
    // first unused out-edge of `node` in iteration order; j_out = its adjacency position
    auto next_unused = [&](uint32_t node, uint32_t &to_out, uint32_t &j_out) -> uint32_t {
        EulerNode2 &r = nodes[node];
        while (r.pos < r.deg) {
            const uint32_t e = r.pos < 3 ? r.eid[r.pos] : ext_eid[r.ext_begin + r.pos - 3];
            if (!is_used(e)) {
                to_out = r.pos < 3 ? r.to[r.pos] : ext_to[r.ext_begin + r.pos - 3];
                j_out = r.pos;
                return e;
            }
            r.pos++;
        }
        return NONE;
    };

    // entries: one per biedge plus one per splice; the FIFO sees every entry once plus one re-push per splice
    HugeBuf<uint32_t> ent_edge(E + 1), ent_next(E + 1), ent_node(E + 1), fifo(E + E / 2 + 2);
    size_t n_ent = 0, fifo_tail = 0;
    Walks out;
    out.edges.reserve(E / 2);
    constexpr size_t PF = 12;  // FIFO prefetch distance
    uint64_t n_walks = 0, n_hinted = 0;

    for (uint64_t e0 = 0; e0 < E; e0++) {
        if (is_used((uint32_t)e0)) continue;
        n_ent = 0; fifo_tail = 0;
        size_t fifo_head = 0;
        uint32_t head = NONE;
        uint32_t start_edge = (uint32_t)e0, start_to = g.e_to[e0], start_node = g.e_from[e0];
        uint32_t splice_at = NONE;

        while (start_edge != NONE) {
            n_walks++;
            const size_t w_begin = n_ent;
            uint32_t e = start_edge, from = start_node, to = start_to;
            // hint: adjacency of `to`, copied into the record we just left (0xFF: not available)
            const uint32_t *h_eid = nullptr, *h_to = nullptr;
            uint32_t h_deg = 0xFF;
            for (;;) {
                set_used(e);
                ent_edge[n_ent] = e;
                ent_node[n_ent] = from;
                ent_next[n_ent] = (uint32_t)(n_ent + 1);
                n_ent++;
                from = to;
                if (h_deg != 0xFF) {  // no memory access on the critical path for this step
                    e = NONE;
                    for (uint32_t q = 0; q < h_deg; q++)
                        if (!is_used(h_eid[q])) { e = h_eid[q]; to = h_to[q]; break; }
                    h_deg = 0xFF;
                    n_hinted++;
                } else {
                    uint32_t j = 0;
                    e = next_unused(from, to, j);
                    if (e != NONE && j < 3) {
                        const EulerNode2 &r = nodes[from];
                        if (r.sub_deg[j] != 0xFF) {
                            h_eid = r.sub_eid[j]; h_to = r.sub_to[j]; h_deg = r.sub_deg[j];
                            // the node after next is one of h_to[*]: start those (<= 3) record fetches and the used-bitmap
                            // words now, one whole step before the chain needs them
                            for (uint32_t q = 0; q < h_deg; q++) {
                                __builtin_prefetch(&nodes[h_to[q]]);
                                __builtin_prefetch(reinterpret_cast<const char *>(&nodes[h_to[q]]) + 64);
                                __builtin_prefetch(&used[(h_eid[q] >> 1) >> 6]);
                            }
                        }
                    }
                }
                if (e == NONE) {
                    if (from != start_node)
                        MTG_DIE("Euler walk stuck at node %u != start node %u: graph is not Eulerian", from, start_node);
                    break;
                }
            }
            const size_t w_end = n_ent;
            if (splice_at == NONE) {
                head = (uint32_t)w_begin;
                ent_next[w_end - 1] = head;
                for (size_t i = w_begin; i < w_end; i++) fifo[fifo_tail++] = (uint32_t)i;
            } else {
                // insert W before x = splice_at: x's edge moves to a fresh entry y behind W, x receives W's first edge
                const uint32_t x = splice_at;
                const uint32_t y = (uint32_t)n_ent++;
                ent_edge[y] = ent_edge[x];
                ent_node[y] = ent_node[x];
                ent_next[y] = ent_next[x];  // if x was the only entry this is x itself: y -> x(W1)
                ent_edge[x] = ent_edge[w_begin];
                if (w_end - w_begin == 1) ent_next[x] = y;
                else { ent_next[x] = (uint32_t)(w_begin + 1); ent_next[w_end - 1] = y; }
                head = y;
                fifo[fifo_head] = y;
                fifo[fifo_tail++] = x;
                for (size_t i = w_begin + 1; i < w_end; i++) fifo[fifo_tail++] = (uint32_t)i;
            }
            // next start edge: first entry in cycle order whose from-node still has an unused out-edge
            start_edge = NONE;
            while (fifo_head < fifo_tail) {
                if (fifo_head + PF < fifo_tail) __builtin_prefetch(&nodes[ent_node[fifo[fifo_head + PF]]]);
                const uint32_t ent = fifo[fifo_head];
                const uint32_t node = ent_node[ent];
                uint32_t to2 = NONE, j = 0;
                const uint32_t cand = next_unused(node, to2, j);
                if (cand != NONE) { start_edge = cand; start_to = to2; start_node = node; splice_at = ent; break; }
                fifo_head++;
            }
        }
        uint32_t ent = head;
        const size_t o0 = out.edges.size();
        out.edges.resize(o0 + n_ent);  // upper bound (the cycle has at most n_ent entries); trimmed below
        size_t o = o0;
        do {
            out.edges[o++] = ent_edge[ent];
            ent = ent_next[ent];
        } while (ent != head);
        out.edges.resize(o);
        out.limits.push_back(out.edges.size());
    }
    if (dbg_t) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] euler_cycles: records %.3f s, walk+splice+emit %.3f s (%llu closed walks, %zu biedges, %.0f%% of steps hinted)\n",
                     std::chrono::duration<double>(t_built - t_begin).count(), std::chrono::duration<double>(t_end - t_built).count(),
                     (unsigned long long)n_walks, out.edges.size(), out.edges.empty() ? 0.0 : 100.0 * n_hinted / out.edges.size());
    }
    return out;
}

}  // namespace mtg
