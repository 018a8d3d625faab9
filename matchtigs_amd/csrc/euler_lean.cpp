// euler_lean.cpp -- Euler bicycle decomposition in the reference's order (bigraph 5.0.1
// compute_minimum_bidirected_eulerian_cycle_decomposition, call site /root/reference/src/implementation/greedytigs/mod.rs:722)
// over 32-byte node records that the GPU builds from the Eulerised dart arrays (finish_device.hip).
//
// Same policy and same sequences as euler_cycles_generic / euler_cycles (host_pipeline.cpp, euler_fast.cpp): an edge and its
// mirror (e ^ 1) are consumed together; cycles start at the lowest unused edge id; the walk takes the first unused out-edge in
// adjacency order (newest first); when stuck at its start node the cycle is scanned from its first edge for a node with an unused
// out-edge, rotated there and continued ("insert the next closed walk before that entry": circular entry list + FIFO).
// Memory: 32 B per node + 12 B per biedge instead of the 256-byte records of euler_fast.cpp, so the exact order is available
// at the human-like size (2^30 nominal edges: 23 GB of records) -- one dependent DRAM access per step, no lookahead copies.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>

#include "../../include/mtg_policy.h"
#include "euler_lean.hpp"
#include "euler_splice_last.hpp"
#include "hugebuf.hpp"
#include "parallel.hpp"

namespace mtg {

Walks euler_cycles_lean(LeanNode *nodes, uint64_t V, const uint32_t *ext_eid, const uint32_t *ext_to, const uint32_t *e_from,
                        const uint32_t *e_to, uint64_t E, HugeArena *arena) {
    if (E % 2) MTG_DIE("edge count must be even (edge / mirror pairs)");
    (void)V;
    if (mtg_policy_euler_splice_last()) return euler_walk_splice_last(nodes, ext_eid, ext_to, e_from, e_to, E);  // policy P5, other setting
    static const bool dbg_t = std::getenv("MTG_DEBUG") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    NumaPin pin(arena ? arena->node : -1);

    HugeBuf<uint64_t> used_buf((E / 2 + 63) / 64 + 1, arena);
    uint64_t *used = used_buf.p;
    std::memset(used, 0, ((E / 2 + 63) / 64 + 1) * 8);
    auto is_used = [&](uint32_t e) -> bool { return (used[(e >> 1) >> 6] >> ((e >> 1) & 63)) & 1ull; };
    auto set_used = [&](uint32_t e) { used[(e >> 1) >> 6] |= 1ull << ((e >> 1) & 63); };
    // MTG_WALK_SIM=1 (with MTG_DEBUG=1): how many record reads the walk of euler_fast.cpp would make on THIS walk with records of other
    // shapes -- shape (a0, a1, ...) = own positions that carry copies, positions copied per head, per head's head, ...; a hinted step
    // succeeds when the position the walk takes lies among the copied ones. Reproduces the measured counts of the shipped shapes
    // exactly ((3,3,2): 35 961 336 at 2^27, (3,3): 49 247 138) and is how (2,2,2,2) and (3,3,3) were ruled out.
    static const bool sim_on = std::getenv("MTG_WALK_SIM") != nullptr;
    constexpr int N_SIM = 8;
    static const int SIM[N_SIM][5] = {{3, 3, 2, 0, 0}, {3, 3, 0, 0, 0}, {3, 2, 2, 2, 0}, {2, 2, 2, 2, 0}, {3, 3, 2, 2, 0}, {2, 2, 2, 2, 2}, {3, 3, 3, 0, 0}, {3, 2, 2, 0, 0}};
    static const int SIM_K[N_SIM] = {3, 2, 4, 4, 4, 5, 3, 3};
    int sim_L[N_SIM] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long sim_reads[N_SIM] = {0, 0, 0, 0, 0, 0, 0, 0};
    // first unused out-edge of `node` in iteration order (moves the node's cursor past used positions)
    auto next_unused = [&](uint32_t node, uint32_t &to_out) -> uint32_t {
        LeanNode &r = nodes[node];
        while (r.pos < r.deg) {
            const uint32_t e = r.pos < 3 ? r.eid[r.pos] : ext_eid[r.ext_begin + r.pos - 3];
            if (!is_used(e)) {
                to_out = r.pos < 3 ? r.to[r.pos] : ext_to[r.ext_begin + r.pos - 3];
                return e;
            }
            r.pos++;
        }
        return NONE;
    };
    auto has_unused = [&](uint32_t node) -> bool {  // like next_unused, without moving the cursor (read-only: host threads use it)
        const LeanNode &r = nodes[node];
        for (uint32_t p = r.pos; p < r.deg; p++)
            if (!is_used(p < 3 ? r.eid[p] : ext_eid[r.ext_begin + p - 3])) return true;
        return false;
    };

    // entries: one per biedge plus one per splice; the FIFO sees every entry once plus one re-push per splice
    // (worst case: every closed sub-walk is one biedge, so entries <= E / 2 + splices <= E; FIFO pushes = biedges. Untouched
    // pages of the mappings cost nothing.)
    const size_t ent_cap = E + 4, fifo_cap = E / 2 + 4;
    HugeBuf<uint32_t> ent_edge(ent_cap, arena), ent_next(ent_cap, arena), ent_node(ent_cap, arena), fifo(fifo_cap, arena);
    size_t n_ent = 0, fifo_tail = 0;
    Walks out;
    out.edges.reserve(E / 2);
    constexpr size_t PF = 12;  // FIFO prefetch distance
    uint64_t n_walks = 0;
    double t_walk = 0, t_scan = 0, t_emit = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

    for (uint64_t e0 = 0; e0 < E; e0++) {
        if (is_used((uint32_t)e0)) continue;
        n_ent = 0;
        fifo_tail = 0;
        size_t fifo_head = 0;
        uint32_t head = NONE;
        uint32_t start_edge = (uint32_t)e0, start_to = e_to[e0], start_node = e_from[e0];
        uint32_t splice_at = NONE;

        while (start_edge != NONE) {
            for (int sc = 0; sc < N_SIM; sc++) sim_L[sc] = 0;
            n_walks++;
            const auto tw0 = now();
            const size_t w_begin = n_ent;
            uint32_t e = start_edge, from = start_node, to = start_to;
            for (;;) {
                if (n_ent + 2 >= ent_cap) MTG_DIE("euler_cycles_lean: internal error (entry arrays exhausted)");
                set_used(e);
                ent_edge[n_ent] = e;
                ent_node[n_ent] = from;
                ent_next[n_ent] = (uint32_t)(n_ent + 1);
                n_ent++;
                from = to;
                e = next_unused(from, to);
                if (sim_on && e != NONE) {  // (experiment: record reads of other record shapes on this very walk)
                    const uint32_t j = nodes[from].pos;
                    for (int sc = 0; sc < N_SIM; sc++) {
                        int &L = sim_L[sc];
                        const int *A = SIM[sc];
                        if (L != 0 && j < (uint32_t)A[L]) L = (L + 1 < SIM_K[sc]) ? L + 1 : 0;
                        else { sim_reads[sc]++; L = j < (uint32_t)A[0] ? 1 : 0; }
                    }
                }
                if (e == NONE) {
                    if (from != start_node)
                        MTG_DIE("Euler walk stuck at node %u != start node %u: graph is not Eulerian", from, start_node);
                    break;
                }
                __builtin_prefetch(&nodes[to]);
            }
            const size_t w_end = n_ent;
            const auto tw1 = now();
            t_walk += secs(tw0, tw1);
            if (fifo_tail + (w_end - w_begin) + 2 >= fifo_cap) MTG_DIE("euler_cycles_lean: internal error (FIFO exhausted)");
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
            // next start edge: first entry in cycle order whose from-node still has an unused out-edge. A long backlog is first
            // narrowed down by host threads (read-only: each finds the first such entry of its chunk).
            start_edge = NONE;
            for (size_t probe = 0; probe < 4096 && fifo_head < fifo_tail; probe++) {
                if (has_unused(ent_node[fifo[fifo_head]])) break;
                fifo_head++;
            }
            if (fifo_tail - fifo_head >= (1u << 18) && !has_unused(ent_node[fifo[fifo_head]])) {
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
                        if (i + PF < hi) __builtin_prefetch(&nodes[ent_node[fifo[i + PF]]]);
                        if (has_unused(ent_node[fifo[i]])) {
                            size_t cur = first_hit.load(std::memory_order_relaxed);
                            while (i < cur && !first_hit.compare_exchange_weak(cur, (size_t)i)) {}
                            return;
                        }
                    }
                }, 64);
                fifo_head = first_hit.load();  // everything before it is exhausted; the sequential loop takes it from here
            }
            while (fifo_head < fifo_tail) {
                if (fifo_head + PF < fifo_tail) __builtin_prefetch(&nodes[ent_node[fifo[fifo_head + PF]]]);
                const uint32_t ent = fifo[fifo_head];
                const uint32_t node = ent_node[ent];
                uint32_t to2 = NONE;
                const uint32_t cand = next_unused(node, to2);
                if (cand != NONE) { start_edge = cand; start_to = to2; start_node = node; splice_at = ent; break; }
                fifo_head++;
            }
            t_scan += secs(tw1, now());
        }
        const auto te0 = now();
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
        t_emit += secs(te0, now());
    }
    if (dbg_t) {
        const auto t_end = std::chrono::steady_clock::now();
        if (sim_on) {
            for (int sc = 0; sc < N_SIM; sc++)
                std::fprintf(stderr, "[mtg] walk sim: shape (%d,%d,%d,%d,%d) -> %llu record reads\n", SIM[sc][0], SIM[sc][1], SIM[sc][2], SIM[sc][3], SIM[sc][4], sim_reads[sc]);
        }
        std::fprintf(stderr, "[mtg] euler_cycles_lean: walk %.3f s, scan for splice points %.3f s, emit %.3f s, total %.3f s (%llu closed walks, %zu biedges, %.1f ns per biedge)\n",
                     t_walk, t_scan, t_emit, std::chrono::duration<double>(t_end - t_begin).count(), (unsigned long long)n_walks, out.edges.size(),
                     out.edges.empty() ? 0.0 : 1e9 * t_walk / (double)out.edges.size());
    }
    return out;
}

}  // namespace mtg
