// euler_fast.cpp -- Euler bicycle decomposition (bigraph 5.0.1 compute_minimum_bidirected_eulerian_cycle_decomposition,
// call site /root/reference/src/implementation/greedytigs/mod.rs:722), latency-optimised host formulation.
//
// Same sequences as the literal algorithm (see host_pipeline.cpp, euler_cycles_generic, for the policy and for how
// "rotate to x and append W" becomes a splice). On the bench graph the decomposition is ONE closed walk of ~11.6 M
// biedges plus a handful of splices, i.e. a single chain of dependent random memory accesses -- the only lever is the
// number of DRAM misses on that chain:
//   v0  used[] / e_to / cursor / e_next_out / e_from ...  ~7 misses per biedge   4.2 s  (profiles/r01_*)
//   v1  one 32-byte record per node (cursor + 3 inline adjacency entries), used-bitmap      ~1 miss per biedge, 1.5 s
//   v2  128-byte records that also carry, for each of the 3 inline out-edges, the inline adjacency of the edge's HEAD
//       node: after one miss the walk knows where it can go from the next node too  ->  ~1 miss per 2 biedges, 1.4 s
//   v3  (this file) 256-byte records with TWO levels of copied adjacency (3 positions of every head node, 2 positions of
//       every head's head): one record access serves up to three steps, and -- what matters more -- the record the walk
//       will need three steps from now is known (up to used-bit changes in between) when the current one arrives, so its
//       fetch overlaps two whole steps instead of a fraction of one.
// Record build and sub-adjacency fill are embarrassingly parallel gathers and run on host threads; the walk itself is
// inherently sequential (each step depends on every edge used so far).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mtg_policy.h"
#include "euler_lean.hpp"
#include "euler_splice_last.hpp"
#include "host_graph.hpp"
#include "hugebuf.hpp"
#include "parallel.hpp"

namespace mtg {

Walks euler_cycles_generic(const HostGraph &g);


// Everything after the records' own adjacency (phase A) is filled: copies of the heads' adjacency (phases B, C), the walk.
template <typename Rec>
static Walks euler_walk_records(Rec *nodes, const uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                                const uint32_t *e_from, const uint32_t *e_to, const uint64_t E, HugeArena *arena_ptr,
                                std::chrono::steady_clock::time_point t_begin, bool have_sub_levels, LeanNode *lean = nullptr,
                                const std::atomic<uint64_t> *arrived = nullptr);
template <typename Rec>
static void seed_from_lean(const LeanNode *lean, Rec *nodes, uint64_t V);

Walks euler_cycles(const HostGraph &g) {
    const uint64_t E = g.edge_count();
    const uint64_t V = g.node_count();
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    g.ensure_linked();
    for (uint64_t n = 0; n < V; n++)
        if (g.out_deg[n] > 65535) return euler_cycles_generic(g);  // not a de Bruijn graph: simple formulation

    const auto t_begin = std::chrono::steady_clock::now();
    // the walk is one latency-bound thread: keep it next to the memory it chases through (the graph's arena from an earlier call, if any)
    NumaPin pin(g.arena.node);

    // ---- records ----
    HugeBuf<EulerNode3> nodes(V, &g.arena);
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
            EulerNode3 &r = nodes[n];
            r.deg = (uint16_t)g.out_deg[n];
            r.pos = 0;
            r.pad = 0;
            r.sub_info = 0;
            r.sub2_info = 0;
            r.ext_begin = ext_begin[n];
            r.eid[0] = r.eid[1] = r.eid[2] = NONE;
            r.to[0] = r.to[1] = r.to[2] = NONE;
            uint32_t i = 0;
            for (uint32_t e = g.head_out[n]; e != NONE; e = g.e_next_out[e], i++) {
                if (i < 3) { r.eid[i] = e; r.to[i] = g.e_to[e]; }
                else { ext_eid[r.ext_begin + i - 3] = e; ext_to[r.ext_begin + i - 3] = g.e_to[e]; }
            }
        }
    });
    return euler_walk_records(nodes.p, V, ext_eid.data(), ext_to.data(), g.e_from.data(), g.e_to.data(), E, &g.arena, t_begin, false);
}

// The same walk from the 32-byte records the GPU builds out of the Eulerised dart arrays (finish_device.hip): a LeanNode IS
// phase A of a record (own adjacency, newest first), so the host graph's adjacency lists are never linked or walked.
Walks euler_cycles_from_lean(const LeanNode *lean, EulerNode3 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                             const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    seed_from_lean(lean, nodes, V);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, false);
}

Walks euler_cycles_from_wide(EulerNode3 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                             const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, true);
}

Walks euler_cycles_from_wide_arriving(EulerNode3 *nodes, LeanNode *lean, const std::atomic<uint64_t> *arrived, uint64_t V, const uint32_t *ext_eid,
                                      const uint32_t *ext_to, const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, true, lean, arrived);
}

Walks euler_cycles_from_mid_arriving(EulerNode2 *nodes, LeanNode *lean, const std::atomic<uint64_t> *arrived, uint64_t V, const uint32_t *ext_eid,
                                     const uint32_t *ext_to, const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, true, lean, arrived);
}

// seeds phase A of the records (own adjacency) from the 32-byte ones
template <typename Rec>
static void seed_from_lean(const LeanNode *lean, Rec *nodes, uint64_t V) {
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++) {
            Rec &r = nodes[n];
            const LeanNode &l = lean[n];
            for (int i = 0; i < 3; i++) { r.eid[i] = l.eid[i]; r.to[i] = l.to[i]; }
            r.deg = l.deg;
            r.pos = 0;
            r.pad = 0;
            r.sub_info = 0;
            r.sub2_info = 0;
            r.ext_begin = l.ext_begin;
        }
    });
}

Walks euler_cycles_from_lean_mid(const LeanNode *lean, EulerNode2 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                                 const uint32_t *e_from, const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    seed_from_lean(lean, nodes, V);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, false);
}

Walks euler_cycles_from_mid(EulerNode2 *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                            const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);
    return euler_walk_records(nodes, V, ext_eid, ext_to, e_from, e_to, E, arena, t_begin, true);
}

// (lean / arrived: the records of nodes >= *arrived are still on their way from the GPU, in node order -- a step that needs one
// takes the node's 32-byte record instead: own adjacency only, one step per record, the same sequences. *arrived only grows and
// reaches V; a node's record is complete and visible once its index lies below it.)
template <typename Rec>
static Walks euler_walk_records(Rec *nodes, const uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to,
                                const uint32_t *e_from, const uint32_t *e_to, const uint64_t E, HugeArena *arena_ptr,
                                std::chrono::steady_clock::time_point t_begin, bool have_sub_levels, LeanNode *lean,
                                const std::atomic<uint64_t> *arrived) {
    constexpr bool L3 = Rec::LEVELS == 3;  // records with the heads' heads
    if (mtg_policy_euler_splice_last()) {  // policy P5, other setting (euler_splice_last.hpp): the plain walk over the records' own adjacency
        (void)V; (void)arena_ptr; (void)t_begin; (void)have_sub_levels; (void)arrived;
        // (records that still arrive: the 32-byte ones are complete from the start and carry the same adjacency)
        return lean ? euler_walk_splice_last(lean, ext_eid, ext_to, e_from, e_to, E) : euler_walk_splice_last(nodes, ext_eid, ext_to, e_from, e_to, E);
    }
    static const bool dbg_t = std::getenv("MTG_DEBUG") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    constexpr unsigned BUILD_THREADS = 128;  // phases B and C are random gathers: latency bound, so more threads than cores pay
    if (!have_sub_levels)
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {  // phase B: first 3 positions of each inline edge's head node
        for (uint64_t n = lo; n < hi; n++) {
            if (n + 16 < hi) {
                const Rec &a = nodes[n + 16];
                for (uint32_t j = 0; j < 3; j++)
                    if (a.to[j] != NONE) __builtin_prefetch(&nodes[a.to[j]]);
            }
            Rec &r = nodes[n];
            const uint32_t d = r.deg < 3 ? r.deg : 3;
            uint32_t info = 0;
            for (uint32_t j = 0; j < d; j++) {
                const Rec &w = nodes[r.to[j]];
                const uint32_t c = w.deg < 3 ? w.deg : 3;
                for (uint32_t q = 0; q < c; q++) { r.sub_eid[j][q] = w.eid[q]; r.sub_to[j][q] = w.to[q]; }
                info |= (c | (w.deg > 3 ? 4u : 0u)) << (3 * j);
            }
            r.sub_info = (uint16_t)info;
        }
    }, BUILD_THREADS);
    const auto t_b = std::chrono::steady_clock::now();
    // phase C: first 2 positions of each head's heads. The head w = to[j] already holds copies of ITS heads' adjacency
    // (phase B), so one gather of w's record serves all three (j, q) slots.
    if constexpr (L3)
    if (!have_sub_levels)
    parallel_ranges(V, [&](uint64_t lo, uint64_t hi) {
        for (uint64_t n = lo; n < hi; n++) {
            if (n + 12 < hi) {
                const Rec &a = nodes[n + 12];
                for (uint32_t j = 0; j < 3; j++)
                    if (a.to[j] != NONE) {
                        const char *p = reinterpret_cast<const char *>(&nodes[a.to[j]]);
                        __builtin_prefetch(p);
                        __builtin_prefetch(p + 64);
                    }
            }
            Rec &r = nodes[n];
            const uint32_t d = r.deg < 3 ? r.deg : 3;
            uint32_t info = 0;
            for (uint32_t j = 0; j < d; j++) {
                const Rec &w = nodes[r.to[j]];
                for (uint32_t q = 0; q < r.sub_cnt(j); q++) {
                    const uint32_t wc = w.sub_cnt(q);  // copied positions of x = w.to[q] (up to 3)
                    const uint32_t c = wc < 2 ? wc : 2;
                    if constexpr (L3)
                        for (uint32_t t = 0; t < c; t++) { r.sub2_eid[j][q][t] = w.sub_eid[q][t]; r.sub2_to[j][q][t] = w.sub_to[q][t]; }
                    info |= (c | ((wc > 2 || w.sub_more(q)) ? 4u : 0u)) << (3 * (3 * j + q));
                }
            }
            r.sub2_info = info;
        }
    }, BUILD_THREADS);
    const auto t_built = std::chrono::steady_clock::now();
    if (dbg_t) {  // how much of the process's anonymous memory sits on huge pages (the records are most of it)
        if (FILE *f = std::fopen("/proc/self/smaps_rollup", "r")) {
            char line[256];
            unsigned long anon = 0, huge = 0;
            while (std::fgets(line, sizeof line, f)) {
                std::sscanf(line, "Anonymous: %lu kB", &anon);
                std::sscanf(line, "AnonHugePages: %lu kB", &huge);
            }
            std::fclose(f);
            std::fprintf(stderr, "[mtg] euler_cycles: anonymous memory %.1f GB, of it on transparent huge pages %.1f GB\n", anon / 1e6, huge / 1e6);
        }
    }
    if (dbg_t)
        std::fprintf(stderr, "[mtg] euler_cycles: records: alloc + own adjacency %.3f s, heads %.3f s, heads of heads %.3f s\n",
                     std::chrono::duration<double>(t_a - t_begin).count(), std::chrono::duration<double>(t_b - t_a).count(),
                     std::chrono::duration<double>(t_built - t_b).count());

    std::vector<uint64_t> used((E / 2 + 63) / 64 + 1, 0);
    auto is_used = [&](uint32_t e) -> bool { return (used[(e >> 1) >> 6] >> ((e >> 1) & 63)) & 1ull; };
    auto set_used = [&](uint32_t e) { used[(e >> 1) >> 6] |= 1ull << ((e >> 1) & 63); };
    // first unused out-edge of `node` in iteration order; j_out = its adjacency position
    auto next_unused = [&](uint32_t node, uint32_t &to_out, uint32_t &j_out) -> uint32_t {
        Rec &r = nodes[node];
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
    HugeBuf<uint32_t> ent_edge(E + 1, arena_ptr), ent_next(E + 1, arena_ptr), ent_node(E + 1, arena_ptr), fifo(E + E / 2 + 2, arena_ptr);
    size_t n_ent = 0, fifo_tail = 0;
    Walks out;
    out.edges.reserve(E / 2);
    constexpr size_t PF = 12;  // FIFO prefetch distance
    uint64_t n_walks = 0, n_hinted = 0, n_full = 0, n_pred_hit = 0, n_lean = 0;
    uint64_t arrived_seen = arrived ? 0 : V;  // nodes below it have their record (a cached lower bound of *arrived)
    auto wait_for_all_records = [&]() {
        while (arrived_seen < V) {
            arrived_seen = arrived->load(std::memory_order_acquire);
            if (arrived_seen < V) std::this_thread::yield();
        }
    };
    uint32_t last_pf = NONE;
    double t_walk = 0, t_scan = 0, t_emit = 0, t_fill = 0, t_jump = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

    std::vector<uint64_t> maybe_unused;  // (built by the first long splice scan) bit per node: it may still have an unused out-edge
    std::vector<uint32_t> breaks;  // entries whose successor is not the next entry (a handful per splice): the cycle is a few long runs
    for (uint64_t e0 = 0; e0 < E; e0++) {
        if (is_used((uint32_t)e0)) continue;
        n_ent = 0; fifo_tail = 0;
        breaks.clear();
        size_t fifo_head = 0;
        uint32_t head = NONE;
        uint32_t start_edge = (uint32_t)e0, start_to = e_to[e0], start_node = e_from[e0];
        uint32_t splice_at = NONE;

        while (start_edge != NONE) {
            n_walks++;
            const auto tw0 = now();
            const size_t w_begin = n_ent;
            uint32_t e = start_edge, from = start_node, to = start_to;
            // hint state: `H` is the last record read; level 2 = `from` is H->to[hj], level 3 = `from` is H->sub_to[hj][hq]
            const Rec *H = nullptr;
            uint32_t level = 0, hj = 0, hq = 0;
            auto prefetch_record = [&](uint32_t node) {
                const char *p = reinterpret_cast<const char *>(&nodes[node]);
                __builtin_prefetch(p);
                __builtin_prefetch(p + 64);
                if constexpr (sizeof(Rec) > 128) {
                    __builtin_prefetch(p + 128);
                    __builtin_prefetch(p + 192);
                }
            };
            for (;;) {
                set_used(e);
                ent_edge[n_ent] = e;
                ent_node[n_ent] = from;
                ent_next[n_ent] = (uint32_t)(n_ent + 1);
                n_ent++;
                from = to;
                bool full = true;
                e = NONE;
                if (level == 2) {  // no memory access on the critical path: the copy of `from`'s adjacency is in H
                    const uint32_t cnt = H->sub_cnt(hj);
                    uint32_t q = 0;
                    for (; q < cnt; q++)
                        if (!is_used(H->sub_eid[hj][q])) { e = H->sub_eid[hj][q]; to = H->sub_to[hj][q]; break; }
                    if (e != NONE) {
                        full = false; hq = q; n_hinted++;
                        if constexpr (L3) level = 3;
                        else { level = 0; prefetch_record(to); }  // (two levels: this was the record's last hint)
                    } else if (!H->sub_more(hj)) { full = false; level = 0; }  // exhausted: the walk is stuck here
                } else if (level == 3) {
                    if constexpr (L3) {
                        const uint32_t cnt = H->sub2_cnt(hj, hq);
                        for (uint32_t t = 0; t < cnt; t++)
                            if (!is_used(H->sub2_eid[hj][hq][t])) { e = H->sub2_eid[hj][hq][t]; to = H->sub2_to[hj][hq][t]; break; }
                        if (e != NONE) { full = false; n_hinted++; prefetch_record(to); }
                        else if (!H->sub2_more(hj, hq)) full = false;
                    }
                    level = 0;
                }
                if (full && from >= arrived_seen && from >= (arrived_seen = arrived->load(std::memory_order_acquire))) {
                    // the node's 256-byte record has not arrived yet: its 32-byte record (own adjacency) decides this step
                    level = 0;
                    n_lean++;
                    LeanNode &l = lean[from];
                    e = NONE;
                    while (l.pos < l.deg) {
                        const uint32_t cand = l.pos < 3 ? l.eid[l.pos] : ext_eid[l.ext_begin + l.pos - 3];
                        if (!is_used(cand)) {
                            e = cand;
                            to = l.pos < 3 ? l.to[l.pos] : ext_to[l.ext_begin + l.pos - 3];
                            break;
                        }
                        l.pos++;
                    }
                    if (e != NONE) { __builtin_prefetch(&lean[to]); prefetch_record(to); last_pf = to; }
                } else if (full) {
                    level = 0;
                    n_full++;
                    n_pred_hit += (from == last_pf);
                    {
                        // The record has just arrived (the one DRAM miss of this step; measured floor on the bench host:
                        // 130-140 ns per dependent miss, tools/host_latency.c). Everything that follows up to the prefetch of
                        // the next record is a chain of used-bit lookups (an 11-MB bitmap at the bench size: L3 hits, 4-5 of
                        // them in sequence), which used to delay that prefetch by ~60 ns per record. So, before looking at
                        // any bit: (a) start the fetch of the record the walk needs next IF every first-listed edge on the way
                        // is still unused (right about half of the time, and a wrong guess costs nothing but a line fill);
                        // (b) start the fetches of all bitmap words the exact computation below can touch, so that its lookups
                        // overlap instead of queueing up.
                        const Rec &r0 = nodes[from];
                        const uint32_t j0 = r0.pos;
                        if (j0 < 3 && j0 < r0.deg) {
                            const uint32_t c1 = r0.sub_cnt(j0);
                            auto touch_bit = [&](uint32_t eid) { __builtin_prefetch(&used[(eid >> 1) >> 6]); };
                            // every record the walk can need three steps from here if it leaves by position j0 (at most 3 x 2:
                            // the first-listed edges are taken on a first visit, the second-listed ones on a second): wrong
                            // guesses cost line fills, a few GB/s in all, and nothing on the dependent chain
                            if (!c1) prefetch_record(r0.to[j0]);
                            for (uint32_t q = 0; q < c1; q++) {
                                const uint32_t c2 = r0.sub2_cnt(j0, q);
                                if (!c2) prefetch_record(r0.sub_to[j0][q]);
                                if constexpr (L3)
                                    for (uint32_t t = 0; t < c2; t++) prefetch_record(r0.sub2_to[j0][q][t]);
                            }
                            for (uint32_t p = j0; p < 3 && p < r0.deg; p++) touch_bit(r0.eid[p]);
                            for (uint32_t q = 0; q < c1; q++) {
                                touch_bit(r0.sub_eid[j0][q]);
                                if constexpr (L3) {
                                    const uint32_t c2 = r0.sub2_cnt(j0, q);
                                    for (uint32_t t = 0; t < c2; t++) touch_bit(r0.sub2_eid[j0][q][t]);
                                }
                            }
                        }
                    }
                    uint32_t j = 0;
                    e = next_unused(from, to, j);
                    if (e != NONE) {
                        if (j < 3) {
                            H = &nodes[from];
                            hj = j;
                            level = 2;
                            // The record needed three steps from now is the head of the first unused copied edge of the first
                            // unused copied edge of `to` -- unless used bits change in between. Start its fetch now.
                            uint32_t pf = to;
                            const uint32_t c1 = H->sub_cnt(j);
                            for (uint32_t q = 0; q < c1; q++)
                                if (!is_used(H->sub_eid[j][q])) {
                                    pf = H->sub_to[j][q];
                                    if constexpr (L3) {
                                        const uint32_t c2 = H->sub2_cnt(j, q);
                                        for (uint32_t t = 0; t < c2; t++)
                                            if (!is_used(H->sub2_eid[j][q][t])) { pf = H->sub2_to[j][q][t]; break; }
                                    }
                                    break;
                                }
                            prefetch_record(pf);
                            last_pf = pf;
                        } else { prefetch_record(to); last_pf = to; }
                    }
                }
                if (e == NONE) {
                    if (from != start_node)
                        MTG_DIE("Euler walk stuck at node %u != start node %u: graph is not Eulerian", from, start_node);
                    break;
                }
            }
            const size_t w_end = n_ent;
            wait_for_all_records();  // (the splice scans below read the records of arbitrary nodes)
            const auto tw1 = now();
            t_walk += secs(tw0, tw1);
            if (splice_at == NONE) {
                head = (uint32_t)w_begin;
                ent_next[w_end - 1] = head;
                breaks.push_back((uint32_t)(w_end - 1));
                {   // (93 M consecutive numbers at the bench size: by several threads)
                    uint32_t *dst = fifo.p + fifo_tail;
                    const size_t first = w_begin;
                    parallel_ranges(w_end - w_begin, [dst, first](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) dst[i] = (uint32_t)(first + i); }, 16);
                    fifo_tail += w_end - w_begin;
                }
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
                breaks.push_back(x); breaks.push_back(y); breaks.push_back((uint32_t)(w_end - 1));
                head = y;
                fifo[fifo_head] = y;
                fifo[fifo_tail++] = x;
                {
                    uint32_t *dst = fifo.p + fifo_tail;
                    const size_t first = w_begin + 1;
                    parallel_ranges(w_end - first, [dst, first](uint64_t lo, uint64_t hi) { for (uint64_t i = lo; i < hi; i++) dst[i] = (uint32_t)(first + i); }, 16);
                    fifo_tail += w_end - first;
                }
            }
            // next start edge: first entry in cycle order whose from-node still has an unused out-edge. A long backlog is
            // first narrowed down by host threads (read-only: each finds the first such entry of its chunk).
            const auto tq0 = now();
            t_fill += secs(tw1, tq0);
            start_edge = NONE;
            auto maybe = [&](uint32_t node) -> bool { return (maybe_unused[node >> 6] >> (node & 63)) & 1ull; };
            // The next start edge, in turns: a sequential stretch checks entries exactly (when splice points are frequent the next one
            // is close, and spawning threads per splice would cost more than the scan); when it has run through its budget without a
            // find and a long rest of the FIFO lies ahead, threads jump over the part of it in which no node can have an unused
            // out-edge.
            for (;;) {
                const bool filtered = !maybe_unused.empty();
                size_t budget = filtered ? ((size_t)1 << 17) : 4096;  // (filtered: a stream over the entries and one bit each; else a record read each)
                bool found = false;
                while (fifo_head < fifo_tail && budget--) {
                    const uint32_t ent = fifo[fifo_head];
                    const uint32_t node = ent_node[ent];
                    if (filtered) {
                        if (!maybe(node)) { fifo_head++; continue; }
                    } else if (fifo_head + PF < fifo_tail) __builtin_prefetch(&nodes[ent_node[fifo[fifo_head + PF]]]);
                    uint32_t to2 = NONE, j = 0;
                    const uint32_t cand = next_unused(node, to2, j);
                    if (cand != NONE) { start_edge = cand; start_to = to2; start_node = node; splice_at = ent; found = true; break; }
                    if (filtered) maybe_unused[node >> 6] &= ~(1ull << (node & 63));  // exhausted for good: edges are only ever used up
                    fifo_head++;
                }
                if (found || fifo_head >= fifo_tail) break;
                if (fifo_tail - fifo_head < (1u << 18)) continue;  // (a short rest: sequentially to its end)
                // Which nodes can still have an unused out-edge at all: the from-nodes of the biedges that are unused NOW (a few per
                // thousand once the first closed sub-walk is through). Edges are only ever used up, so the set stays a superset for
                // the scans that follow: the threads below test one bit per entry (an 11-MB bitmap) instead of reading the entry's node
                // record (23 GB, random), and the sequential stretch after them checks exactly.
                {   // (rebuilt before every jump: a bit left over from an earlier scan whose node has been exhausted since costs a
                    // sequential stretch and another jump -- a hundred of them were most of the scan time at the bench size)
                    maybe_unused.assign((V + 63) / 64, 0);
                    const uint64_t n_b = E / 2, n_words = (n_b + 63) / 64;
                    parallel_ranges(n_words, [&](uint64_t lo, uint64_t hi) {
                        for (uint64_t w = lo; w < hi; w++) {
                            uint64_t free_bits = ~used[w];
                            if (w == n_words - 1 && (n_b & 63)) free_bits &= (1ull << (n_b & 63)) - 1ull;
                            while (free_bits) {
                                const uint64_t b = w * 64 + (uint64_t)__builtin_ctzll(free_bits);
                                free_bits &= free_bits - 1;
                                for (uint64_t e = 2 * b; e < 2 * b + 2; e++) {
                                    const uint32_t node = e_from[e];
                                    __atomic_fetch_or(&maybe_unused[node >> 6], 1ull << (node & 63), __ATOMIC_RELAXED);
                                }
                            }
                        }
                    }, 16);
                }
                const auto tj0 = now();
                std::atomic<size_t> first_hit{fifo_tail};
                const size_t base = fifo_head;
                // tasks of 64 K entries handed out in FIFO order: all threads work on the earliest unfinished stretch, so the
                // scan takes (position of the first hit) / (throughput of all threads) wherever that hit lies
                constexpr uint64_t TASK = 1u << 16;
                const uint64_t n_scan = fifo_tail - base;
                parallel_tasks((n_scan + TASK - 1) / TASK, [&](uint64_t task) {
                    const uint64_t lo = base + task * TASK, hi = std::min<uint64_t>(base + n_scan, lo + TASK);
                    if (first_hit.load(std::memory_order_relaxed) < lo) return;
                    for (uint64_t i = lo; i < hi; i++) {
                        if (maybe(ent_node[fifo[i]])) {
                            size_t cur = first_hit.load(std::memory_order_relaxed);
                            while (i < cur && !first_hit.compare_exchange_weak(cur, (size_t)i)) {}
                            return;
                        }
                    }
                }, 32);
                fifo_head = first_hit.load();  // everything before it is exhausted; the next sequential stretch takes it from here
                t_jump += secs(tj0, now());
            }
            t_scan += secs(tw1, now());
        }
        const auto te0 = now();
        // The cycle in list order. Entries follow each other in the arrays except at the `breaks`, so it is copied run by run (the
        // long runs by several threads) instead of followed pointer by pointer: 93 M entries in a dozen runs at the bench size.
        std::sort(breaks.begin(), breaks.end());
        breaks.erase(std::unique(breaks.begin(), breaks.end()), breaks.end());
        uint32_t ent = head;
        const size_t o0 = out.edges.size();
        out.edges.resize(o0 + n_ent);  // upper bound (the cycle has at most n_ent entries); trimmed below
        size_t o = o0;
        do {
            const auto b = std::lower_bound(breaks.begin(), breaks.end(), ent);  // end of the run that starts at ent (the last entry is a break)
            if (b == breaks.end()) MTG_DIE("euler_cycles: internal error (entry list has no end)");
            const size_t run = (size_t)*b - ent + 1;
            if (o + run > o0 + n_ent) MTG_DIE("euler_cycles: internal error (entry list longer than its entries)");
            uint32_t *dst = out.edges.data() + o;
            const uint32_t *src = ent_edge.p + ent;
            parallel_ranges(run, [&](uint64_t lo, uint64_t hi) { std::memcpy(dst + lo, src + lo, (hi - lo) * sizeof(uint32_t)); }, 16);
            o += run;
            ent = ent_next[*b];
        } while (ent != head);
        out.edges.resize(o);
        out.limits.push_back(out.edges.size());
        t_emit += secs(te0, now());
    }
    if (dbg_t) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mtg] euler_cycles: walk %.3f s, scan for splice points %.3f s, emit %.3f s; %llu record reads, %.0f%% of them at the predicted node\n",
                     t_walk, t_scan, t_emit, (unsigned long long)n_full, n_full ? 100.0 * n_pred_hit / n_full : 0.0);
        std::fprintf(stderr, "[mtg] euler_cycles: of the scan time: splices + FIFO fill %.3f s, parallel jumps %.3f s, sequential stretches %.3f s\n", t_fill, t_jump,
                     t_scan - t_fill - t_jump);
        if (arrived) std::fprintf(stderr, "[mtg] euler_cycles: %llu steps took the 32-byte record of a node whose larger record was still on its way\n", (unsigned long long)n_lean);
        std::fprintf(stderr, "[mtg] euler_cycles: records %.3f s, walk+splice+emit %.3f s (%llu closed walks, %zu biedges, %.0f%% of steps hinted)\n",
                     std::chrono::duration<double>(t_built - t_begin).count(), std::chrono::duration<double>(t_end - t_built).count(),
                     (unsigned long long)n_walks, out.edges.size(), out.edges.empty() ? 0.0 : 100.0 * n_hinted / out.edges.size());
    }
    return out;
}

}  // namespace mtg
